// CSR matrix handles (upload, launch plan) and the plain alpha/beta SpMV entry point.
#include "bpcg2.h"
#include "csr_stream.h"
#include "precond.h"

#include <algorithm>
#include <climits>
#include <vector>

namespace nss {

#ifndef NSS_PLAN_FILL_CHIP
#define NSS_PLAN_FILL_CHIP 1
#endif
constexpr int kMinRowBlocks = 2048;   // 256 CUs x 8 resident workgroups

#ifndef NSS_DIRECT_MIN_ROWS
#define NSS_DIRECT_MIN_ROWS (1 << 21)
#endif

// Large matrices whose rows hold at most kDirectWidth entries (B^T of the staggered grid: two per row) get a padded
// fixed-width copy and the row-per-lane kernel (csr_direct_kernel).  Below NSS_DIRECT_MIN_ROWS rows the iteration
// is launch-bound and the one launch a pair of matrices shares (csr_stream_dual_kernel) is worth more.
static int64_t g_direct_min_rows = NSS_DIRECT_MIN_ROWS;   // nss_csr_direct_rows_threshold() (tests, measurements)

static int g_pair_mode = -1;
bool pair_staging_enabled() { return g_pair_mode != 0; }

bool direct_rows_candidate(int32_t m, const int32_t* rowptr) {
#if NSS_DIRECT_ROWS
  if (m < g_direct_min_rows) return false;
  for (int32_t r = 0; r < m; ++r)
    if (rowptr[r + 1] - rowptr[r] > kDirectWidth) return false;
  return true;
#else
  (void)m;
  (void)rowptr;
  return false;
#endif
}

void plan_row_blocks(int32_t m, int64_t nnz, const int32_t* rowptr, int32_t* rg_out, int32_t* chunk_out,
                     std::vector<int32_t>& blk, const int32_t* cuts, int ncuts, int products, int max_rows,
                     const uint8_t* row_pos) {
  const double mean = m > 0 ? double(nnz) / double(m) : 0.0;
  const int chunk = kChunk;
  *chunk_out = chunk;
  products = std::max(kBlock, std::min(products, chunk));
  // Lanes per row: the largest power of two for which one reduce pass (kBlock / rg rows) still
  // covers a full chunk of products, i.e. rg ~ mean / 8.  Long rows then fill the LDS chunk
  // (all 8 loads per lane in flight) and every row's bounds and epilogue operands are
  // prefetched (csr_stream.h); rows shorter than 8 take several passes with one lane per row.
  int rg = 1;
  while (rg < kWave && double(2 * rg) * (chunk / kBlock) <= mean) rg *= 2;
  *rg_out = rg;
  const int rows_per_pass = kBlock / rg;
  int passes = 1;
  if (rg == 1) {
    passes = mean > 0.0 ? int(double(products) / (mean * rows_per_pass)) : kMaxRowsPerBlock / rows_per_pass;
    passes = std::max(1, passes);
#if NSS_PLAN_FILL_CHIP
    // small matrices (a slab of a partitioned system, the small configs): rather more, shorter row
    // blocks than fewer than ~8 workgroups per CU
    while (passes > 1 && int64_t(m) / (int64_t(passes) * rows_per_pass) < kMinRowBlocks) --passes;
#endif
  }
  int row_cap = std::min(kMaxRowsPerBlock, passes * rows_per_pass);
  if (max_rows > 0) row_cap = std::min(row_cap, max_rows);
  if (direct_rows_candidate(m, rowptr)) row_cap = std::min(row_cap, kDirectRows);   // see csr_direct_kernel
  blk.clear();
  blk.push_back(0);
  int32_t r = 0;
  int ci = 0;
  while (r < m) {
    const int32_t start = r;
    int64_t acc = 0;
    while (ci < ncuts && cuts[ci] <= start) ++ci;          // next cut strictly above the block start
    const int32_t limit = ci < ncuts ? std::min<int32_t>(m, cuts[ci]) : m;
    while (r < limit && r - start < row_cap) {
      const int64_t len = int64_t(rowptr[r + 1]) - rowptr[r];
      if (acc + len > products) break;
      acc += len;
      ++r;
    }
    if (r == start) ++r;  // a single row longer than the chunk gets a block of its own
    if (row_pos) {        // rows that belong together stay together: end where a group starts (or take the whole group)
      int32_t e = r;
      while (e > start && e < limit && row_pos[e] != 0) --e;
      if (e == start) {
        e = r;
        while (e < limit && row_pos[e] != 0) ++e;
      }
      r = e;
    }
    blk.push_back(r);
  }
}

// ---- 16-bit column offsets -----------------------------------------------------------------------
#ifndef NSS_COL16
#define NSS_COL16 1
#endif

// one workgroup per row block: the distinct 4096-column windows its entries fall into (lane 0, a
// linear table of at most kWindows; consecutive entries mostly repeat the window).  More than
// kWindows raises *wide; row blocks that are one over-long row are skipped (the kernel reads their
// 4-byte indices).
__global__ __launch_bounds__(kBlock) void col_windows_kernel(int32_t nblk, const int32_t* __restrict__ rowblk,
                                                              const int32_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ col, int32_t chunk,
                                                              int32_t* __restrict__ base, int32_t* __restrict__ wide) {
  const int b = blockIdx.x * kBlock + threadIdx.x;      // one lane per row block
  if (b >= nblk) return;
  const int p0 = rowptr[rowblk[b]], p1 = rowptr[rowblk[b + 1]];
  int32_t tab[kWindows];
  int cnt = 0;
  if (p1 - p0 <= chunk) {
    int32_t last = -1;
    for (int p = p0; p < p1; ++p) {
      const int32_t w = col[p] >> kWindowBits;
      if (w == last) continue;
      last = w;
      bool found = false;
      for (int k = 0; k < cnt; ++k) found = found || tab[k] == w;
      if (found) continue;
      if (cnt == kWindows) {
        atomicOr(wide, 1);
        break;
      }
      tab[cnt++] = w;
    }
  }
  for (int k = 0; k < kWindows; ++k) base[b * kWindows + k] = k < cnt ? tab[k] << kWindowBits : 0;
}

__global__ __launch_bounds__(kBlock) void col_pack16_kernel(int32_t nblk, const int32_t* __restrict__ rowblk,
                                                             const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ col, int32_t chunk,
                                                             const int32_t* __restrict__ base,
                                                             uint16_t* __restrict__ col16) {
  __shared__ int32_t window[kWindows];
  const int b = blockIdx.x;
  if (b >= nblk) return;
  if (threadIdx.x < kWindows) window[threadIdx.x] = base[b * kWindows + threadIdx.x];
  __syncthreads();
  const int p0 = rowptr[rowblk[b]], p1 = rowptr[rowblk[b + 1]];
  if (p1 - p0 > chunk) return;
  for (int p = p0 + threadIdx.x; p < p1; p += kBlock) {
    const int32_t c = col[p];
    const int32_t w = (c >> kWindowBits) << kWindowBits;
    int k = 0;
    while (k < kWindows - 1 && window[k] != w) ++k;
    col16[p] = uint16_t((k << kWindowBits) | (c & ((1 << kWindowBits) - 1)));
  }
}

// ---- staged operand: segments of consecutive columns per row block ------------------------------------
#ifndef NSS_STAGE_X
#define NSS_STAGE_X 1
#endif
constexpr int kSegGap = 8;   // two columns at most this far apart belong to the same run (unused columns in between are copied too)

// One workgroup per row block: sort the block's columns (bitonic, in LDS), cut the sorted list into runs where
// two neighbours are more than kSegGap apart, write the descriptor (csr_stream.h: kSegWords) and, per entry,
// the position of its column in the concatenation of the runs.  A block with more than kSegMax runs or more
// than `chunk` staged columns counts into *nbad (the matrix is then not staged); a block that is one over-long
// row keeps total = 0 (the kernel reduces it from the 4-byte columns).
__global__ __launch_bounds__(kBlock) void seg_build_kernel(int32_t nblk, const int32_t* __restrict__ rowblk,
                                                            const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ col, int32_t chunk,
                                                            int32_t ncols, int32_t* __restrict__ desc,
                                                            uint16_t* __restrict__ pos16, int32_t* __restrict__ nbad,
                                                            int32_t* __restrict__ maxtotal) {
  __shared__ int32_t cols[kChunk];
  __shared__ int32_t hidx[kSegMax];
  __shared__ int32_t sstart[kSegMax], spre[kSegMax];
  __shared__ int32_t hcount, ok;
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  if (b >= nblk) return;
  const int r0 = rowblk[b], r1 = rowblk[b + 1];
  const int p0 = rowptr[r0];
  const int cnt = rowptr[r1] - p0;
  int32_t* d = desc + size_t(b) * kSegWords;
  if (tid < kSegWords) {
    int32_t w = 0;
    if (tid == 0) w = r0;
    if (tid == 1) w = r1;
    if (tid == 2) w = p0;
    if (tid == 3) w = cnt;
    if (tid >= kSegPre && tid < kSegPre + kSegMax - 1) w = INT_MAX;
    d[tid] = w;
  }
  if (tid == 0) {
    hcount = 0;
    ok = 0;
  }
  if (cnt == 0 || cnt > chunk) return;
  int npad = 2;
  while (npad < cnt) npad <<= 1;
  for (int i = tid; i < npad; i += kBlock) cols[i] = i < cnt ? col[p0 + i] : INT_MAX;
  __syncthreads();
  for (int k = 2; k <= npad; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npad; i += kBlock) {
        const int l = i ^ j;
        if (l > i) {
          const int32_t x = cols[i], y = cols[l];
          if ((x > y) == ((i & k) == 0)) {
            cols[i] = y;
            cols[l] = x;
          }
        }
      }
      __syncthreads();
    }
  for (int i = tid; i < cnt; i += kBlock)
    if (i == 0 || cols[i] - cols[i - 1] > kSegGap) {
      const int slot = atomicAdd(&hcount, 1);
      if (slot < kSegMax) hidx[slot] = i;
    }
  __syncthreads();
  const int nseg = hcount;
  if (nseg > kSegMax) {
    if (tid == 0) atomicAdd(nbad, 1);
    return;
  }
  if (tid == 0) {
    for (int i = 1; i < nseg; ++i) {                     // the heads in ascending order (<= kSegMax of them)
      const int32_t h = hidx[i];
      int j = i - 1;
      while (j >= 0 && hidx[j] > h) {
        hidx[j + 1] = hidx[j];
        --j;
      }
      hidx[j + 1] = h;
    }
    int32_t acc = 0;
    bool fits = true;
    for (int s2 = 0; s2 < nseg; ++s2) {
      int32_t first = cols[hidx[s2]];
      int32_t last = cols[(s2 + 1 < nseg ? hidx[s2 + 1] : cnt) - 1];
      // the kernel copies PAIRS of doubles (16-byte LDS-DMA pieces): even run lengths, hence even run starts in
      // the copy; the extra column is a neighbour inside [0, ncols) (runs are more than kSegGap apart)
      if ((last - first + 1) % 2 != 0) {
        if (last + 1 < ncols) ++last;
        else if (first > 0) --first;
        else fits = false;
      }
      sstart[s2] = first;
      spre[s2] = acc;
      acc += last - first + 1;
    }
    if (fits && acc <= chunk) {
      ok = 1;
      atomicMax(maxtotal, acc);
      d[4] = nseg;
      d[5] = acc;
      for (int s2 = 1; s2 < nseg; ++s2) d[kSegPre + s2 - 1] = spre[s2];
      for (int s2 = 0; s2 < nseg; ++s2) d[kSegOff + s2] = sstart[s2] - spre[s2];
    } else {
      atomicAdd(nbad, 1);
    }
  }
  __syncthreads();
  if (!ok) return;
  for (int i = tid; i < cnt; i += kBlock) {
    const int32_t c = col[p0 + i];
    int s2 = 0;
    for (int t = 1; t < nseg; ++t)
      if (c >= sstart[t]) s2 = t;
    pos16[p0 + i] = uint16_t(c - sstart[s2] + spre[s2]);
  }
}

// ---- grouped column stream ----------------------------------------------------------------------
// bad[0] != 0 unless every row start is a multiple of gb and every entry that is not the first of its
// aligned group of gb continues the column run of its predecessor
__global__ __launch_bounds__(kBlock) void col_group_check_kernel(int32_t m, int64_t nnz, const int32_t* __restrict__ rowptr,
                                                                  const int32_t* __restrict__ col, int32_t gb,
                                                                  int32_t* __restrict__ bad) {
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  bool b = false;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i <= m; i += stride) b = b || (rowptr[i] % gb != 0);
  for (int64_t p = int64_t(blockIdx.x) * kBlock + threadIdx.x; p < nnz; p += stride)
    if (p % gb != 0) b = b || (col[p] != col[p - 1] + 1);
  if (b) atomicOr(bad, 1);
}

__global__ __launch_bounds__(kBlock) void col_group_pack_kernel(int64_t ngroups, int32_t gb, const uint16_t* __restrict__ c16,
                                                                 uint16_t* __restrict__ out) {
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  for (int64_t g = int64_t(blockIdx.x) * kBlock + threadIdx.x; g < ngroups; g += stride) out[g] = c16[g * gb];
}

#ifndef NSS_COL_GROUPS
#define NSS_COL_GROUPS 1
#endif

// one 16-bit index per group of gb entries: `*stream` is replaced by its packed form
static void pack_groups(uint16_t** stream, int64_t nnz, int gb, hipStream_t st) {
  if (!*stream) return;
  const int64_t ngroups = nnz / gb;
  uint16_t* packed = nullptr;
  try {
    NSS_HIP(hipMalloc(&packed, sizeof(uint16_t) * (size_t(ngroups) + 8)));
    NSS_HIP(hipMemsetAsync(packed, 0, sizeof(uint16_t) * (size_t(ngroups) + 8), st));
    hipLaunchKernelGGL(col_group_pack_kernel, dim3(stream_grid(ngroups, kBlock * 4)), dim3(kBlock), 0, st, ngroups, gb,
                       *stream, packed);
    NSS_CHECK_LAUNCH();
    NSS_HIP(hipStreamSynchronize(st));
  } catch (...) {
    (void)hipFree(packed);
    throw;
  }
  (void)hipFree(*stream);
  *stream = packed;
}

// Largest gb in 16 .. 2 for which the matrix is made of aligned runs of gb consecutive columns: keep one
// 16-bit index per run in both 16-bit streams (see nss_csr_s::gb; consecutive columns have consecutive window
// offsets and consecutive staged positions).  Every row block starts at a row start, hence at a multiple of gb.
static void group_columns(nss_csr_s& A, hipStream_t st) {
#if NSS_COL_GROUPS
  if ((!A.col16 && !A.pos16) || A.nnz < 16) return;
  // host-side pre-filter on the first rows (a few KB): candidates that already fail there -- every
  // candidate, for an operator without column runs -- never cost a pass over the matrix
  const int32_t probe_rows = std::min<int32_t>(A.m, 256);
  std::vector<int32_t> h_ptr(size_t(probe_rows) + 1);
  NSS_HIP(hipMemcpy(h_ptr.data(), A.rowptr, sizeof(int32_t) * h_ptr.size(), hipMemcpyDeviceToHost));
  std::vector<int32_t> h_col(size_t(std::max<int32_t>(1, h_ptr.back())));
  if (h_ptr.back() > 0)
    NSS_HIP(hipMemcpy(h_col.data(), A.col, sizeof(int32_t) * size_t(h_ptr.back()), hipMemcpyDeviceToHost));
  auto plausible = [&](int gb) {
    for (int32_t r = 0; r <= probe_rows; ++r)
      if (h_ptr[size_t(r)] % gb != 0) return false;
    for (int32_t p = 1; p < h_ptr.back(); ++p)
      if (p % gb != 0 && h_col[size_t(p)] != h_col[size_t(p) - 1] + 1) return false;
    return true;
  };
  int32_t* bad = nullptr;
  try {
    NSS_HIP(hipMalloc(&bad, sizeof(int32_t)));
    for (int gb = 16; gb >= 2; --gb) {
      if (A.nnz % gb != 0 || !plausible(gb)) continue;
      NSS_HIP(hipMemsetAsync(bad, 0, sizeof(int32_t), st));
      hipLaunchKernelGGL(col_group_check_kernel, dim3(stream_grid(A.nnz, kBlock * 4)), dim3(kBlock), 0, st, A.m, A.nnz,
                         A.rowptr, A.col, gb, bad);
      NSS_CHECK_LAUNCH();
      int32_t h_bad = 1;
      NSS_HIP(hipMemcpyAsync(&h_bad, bad, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      NSS_HIP(hipStreamSynchronize(st));
      if (h_bad != 0) continue;
      pack_groups(&A.col16, A.nnz, gb, st);
      pack_groups(&A.pos16, A.nnz, gb, st);
      A.gb = gb;
      break;
    }
  } catch (...) {
    (void)hipFree(bad);
    throw;
  }
  (void)hipFree(bad);
#else
  (void)A;
  (void)st;
#endif
}

// Shortest mean row for which the matrix is staged.  Two entries per row (B^T) pay where launches dominate
// (1e5 DoF: MINRES 23.7-25.2 -> 21.8 us per iteration, BPCG v1 40 -> 37.5; 1e6 DoF: no change); the large
// two-entry-per-row matrices take the row-per-lane kernel instead (direct_rows).
#ifndef NSS_STAGE_MIN_MEAN
#define NSS_STAGE_MIN_MEAN 2
#endif

// Staged operand form: A.pos16 / A.blkseg when the matrix takes it.
static void stage_columns(nss_csr_s& A, hipStream_t st) {
#if NSS_STAGE_X
  if (A.nnz < int64_t(NSS_STAGE_MIN_MEAN) * A.m) return;
  int32_t* desc = nullptr;
  int32_t* nbad = nullptr;            // [0] row blocks that do not fit, [1] largest number of staged columns of a block
  uint16_t* p16 = nullptr;
  A.pair_ok = false;
  try {
    NSS_HIP(hipMalloc(&desc, sizeof(int32_t) * size_t(A.nblk) * kSegWords));
    NSS_HIP(hipMalloc(&nbad, 2 * sizeof(int32_t)));
    NSS_HIP(hipMalloc(&p16, sizeof(uint16_t) * (size_t(A.nnz) + 8)));
    NSS_HIP(hipMemsetAsync(nbad, 0, 2 * sizeof(int32_t), st));
    NSS_HIP(hipMemsetAsync(p16, 0, sizeof(uint16_t) * (size_t(A.nnz) + 8), st));
    hipLaunchKernelGGL(seg_build_kernel, dim3(A.nblk), dim3(kBlock), 0, st, A.nblk, A.rowblk, A.rowptr, A.col, A.chunk,
                       A.n, desc, p16, nbad, nbad + 1);
    NSS_CHECK_LAUNCH();
    int32_t h_bad[2] = {0, 0};
    NSS_HIP(hipMemcpyAsync(h_bad, nbad, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    NSS_HIP(hipStreamSynchronize(st));
    if (h_bad[0] == 0) {              // every row block fits (the staged kernels have no per-block fallback)
      A.pos16 = p16;
      A.blkseg = desc;
      A.pair_ok = h_bad[1] <= A.chunk / 2;
      p16 = nullptr;
      desc = nullptr;
    }
  } catch (...) {
    (void)hipFree(desc);
    (void)hipFree(nbad);
    (void)hipFree(p16);
    throw;
  }
  (void)hipFree(desc);
  (void)hipFree(nbad);
  (void)hipFree(p16);
#else
  (void)A;
  (void)st;
#endif
}

// ---- reuse-aware dispatch order (nss_csr_s::blkdisp) ----------------------------------------------------
// -1: automatic; 0: never; > 0: tile whenever a period is found, with this many planes per tile.
// DEFAULT OFF -- measured at 5e7 DoF (profiles/r03_ab_dispatch_cfg5.txt): the tiled walk does what it was built for --
// the dominant launch fetches 5.54 GB instead of 6.20 GB through the L2s (1.05 x instead of 1.18 x its algorithmic
// reads), the plain A SpMV 3.25 instead of 3.78 GB -- and is not faster: 8 planes x 1 block per visit -8 %, with
// 192-block visits (a CU's resident workgroups then stay in one plane) 0 ... -1 % against the natural order.  The
// re-reads it removes were evidently served by the 256-MB memory-side cache (the natural order's reuse distance is
// ~200 MB of streamed data across the 8 XCDs), not by HBM; the launch is held elsewhere.
#ifndef NSS_DISPATCH_DEFAULT
#define NSS_DISPATCH_DEFAULT 0
#endif
static int g_dispatch_planes = NSS_DISPATCH_DEFAULT;
#ifndef NSS_DISPATCH_MIN_PERIOD
#define NSS_DISPATCH_MIN_PERIOD 256   // row blocks between a block and the one that re-reads its far runs: below, the XCD's L2 still holds them
#endif
#ifndef NSS_DISPATCH_PLANES
#define NSS_DISPATCH_PLANES 8
#endif
static int g_dispatch_min_period = NSS_DISPATCH_MIN_PERIOD;
#ifndef NSS_DISPATCH_RUN
#define NSS_DISPATCH_RUN 1            // consecutive row blocks of one plane before the walk moves to the next plane
#endif
static int g_dispatch_run = NSS_DISPATCH_RUN;

namespace {
struct Runs {
  int n = 0;
  int32_t lo[kSegMax], hi[kSegMax];
};

Runs runs_of(const int32_t* d) {
  Runs r;
  const int nseg = d[4], total = d[5];
  for (int s = 0; s < nseg && s < kSegMax; ++s) {
    const int32_t pre = s == 0 ? 0 : d[kSegPre + s - 1];
    const int32_t nxt = s + 1 < nseg ? d[kSegPre + s] : total;
    r.lo[r.n] = d[kSegOff + s] + pre;
    r.hi[r.n] = r.lo[r.n] + (nxt - pre);
    ++r.n;
  }
  return r;
}

bool share_columns(const Runs& a, const Runs& b) {
  for (int i = 0; i < a.n; ++i)
    for (int j = 0; j < b.n; ++j)
      if (a.lo[i] < b.hi[j] && b.lo[j] < a.hi[i]) return true;
  return false;
}
}  // namespace

// The period P of a matrix: how many positions further down the natural order the row blocks sit that stage the far
// runs of a block again.  For a sample of blocks b: the blocks b + p that share columns with b form a cluster right
// behind b (its neighbours in the same grid plane), a gap, and a cluster around p = P (the next plane); P = the
// centre of that second cluster, the median over the sample.  0 when there is no second cluster within reach.
static double dispatch_period(const std::vector<int32_t>& desc, int nblk) {
  const int reach = std::min(nblk - 1, 8192);
  if (nblk < 64 || reach < 16) return 0.0;
  std::vector<double> found;
  const int samples = 24;
  for (int q = 0; q < samples; ++q) {
    const int b = int((int64_t(nblk - 1 - reach) * (2 * q + 1)) / (2 * samples));
    const Runs rb = runs_of(&desc[size_t(b) * kSegWords]);
    if (rb.n == 0) continue;
    int p = 1;
    while (p <= reach && share_columns(rb, runs_of(&desc[size_t(b + p) * kSegWords]))) ++p;    // near cluster
    const int gap0 = p;
    while (p <= reach && !share_columns(rb, runs_of(&desc[size_t(b + p) * kSegWords]))) ++p;   // gap
    if (p > reach) continue;
    const int first = p;
    while (p <= reach && share_columns(rb, runs_of(&desc[size_t(b + p) * kSegWords]))) ++p;    // second cluster
    if (first - gap0 < 8) continue;                       // no real gap: not a plane structure
    found.push_back(0.5 * double(first + p - 1));
  }
  if (found.size() < size_t(samples) / 2) return 0.0;
  std::sort(found.begin(), found.end());
  const double med = found[found.size() / 2];
  // a consistent structure: most samples agree with the median
  size_t agree = 0;
  for (double v : found) agree += std::abs(v - med) <= 0.02 * med + 2.0 ? 1 : 0;
  return agree * 4 >= found.size() * 3 ? med : 0.0;
}

// dispatch order of the row blocks [lo, hi) of one XCD: tiles of `planes` periods, inside a tile the blocks at equal
// offset of every period back to back
static void tile_order(int lo, int hi, double period, int planes, int run, std::vector<int32_t>& out) {
  int base = lo;
  run = std::max(1, run);
  while (base < hi) {
    int start[64 + 1];
    for (int k = 0; k <= planes; ++k) start[k] = std::min(hi, base + int(std::llround(double(k) * period)));
    int longest = 0;
    for (int k = 0; k < planes; ++k) longest = std::max(longest, start[k + 1] - start[k]);
    for (int j0 = 0; j0 < longest; j0 += run)
      for (int k = 0; k < planes; ++k)
        for (int j = j0; j < j0 + run && start[k] + j < start[k + 1]; ++j) out.push_back(start[k] + j);
    base = start[planes];
  }
}

static void build_dispatch(nss_csr_s& A, hipStream_t st) {
  if (!A.blkseg || g_dispatch_planes == 0 || A.nblk < 8 * kXcds) return;
  std::vector<int32_t> desc(size_t(A.nblk) * kSegWords);
  NSS_HIP(hipMemcpyAsync(desc.data(), A.blkseg, sizeof(int32_t) * desc.size(), hipMemcpyDeviceToHost, st));
  NSS_HIP(hipStreamSynchronize(st));
  const double period = dispatch_period(desc, A.nblk);
  const int planes = std::min(64, g_dispatch_planes > 0 ? g_dispatch_planes : NSS_DISPATCH_PLANES);
  if (period <= 0.0 || (g_dispatch_planes < 0 && period < double(g_dispatch_min_period))) return;
  const int per_xcd = (A.nblk + kXcds - 1) / kXcds;
  if (period * 2.0 > double(per_xcd)) return;              // fewer than two planes per XCD: nothing to tile
  std::vector<int32_t> table(size_t(kXcds) * per_xcd * kSegWords, 0);
  std::vector<int32_t> order;
  for (int x = 0; x < kXcds; ++x) {
    const int lo = std::min(A.nblk, x * per_xcd), hi = std::min(A.nblk, (x + 1) * per_xcd);
    order.clear();
    tile_order(lo, hi, period, planes, g_dispatch_run, order);
    if (int(order.size()) != hi - lo) throw Error("dispatch order does not cover the XCD's row blocks");
    for (int q = 0; q < per_xcd; ++q) {
      int32_t* slot = &table[(size_t(x) * per_xcd + q) * kSegWords];
      if (q < int(order.size())) {
        std::copy_n(&desc[size_t(order[q]) * kSegWords], kSegWords, slot);
        slot[kSegBlock] = order[q];
      } else {
        slot[kSegBlock] = -1;                              // padding slot
      }
    }
  }
  int32_t* dev = nullptr;
  NSS_HIP(hipMalloc(&dev, sizeof(int32_t) * table.size()));
  try {
    NSS_HIP(hipMemcpy(dev, table.data(), sizeof(int32_t) * table.size(), hipMemcpyHostToDevice));
  } catch (...) {
    (void)hipFree(dev);
    throw;
  }
  A.blkdisp = dev;
  A.disp_period = period;
  A.disp_planes = planes;
}

// 16-bit window form: A.col16 / A.blkbase when every row block fits kWindows windows.
static void window_columns(nss_csr_s& A, hipStream_t st) {
  int32_t* base = nullptr;
  int32_t* wide = nullptr;
  uint16_t* c16 = nullptr;
  try {
    NSS_HIP(hipMalloc(&base, sizeof(int32_t) * size_t(A.nblk) * kWindows));
    NSS_HIP(hipMalloc(&wide, sizeof(int32_t)));
    NSS_HIP(hipMemsetAsync(wide, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(col_windows_kernel, dim3((A.nblk + kBlock - 1) / kBlock), dim3(kBlock), 0, st, A.nblk, A.rowblk,
                       A.rowptr, A.col, A.chunk, base, wide);
    NSS_CHECK_LAUNCH();
    int32_t h_wide = 0;
    NSS_HIP(hipMemcpyAsync(&h_wide, wide, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    NSS_HIP(hipStreamSynchronize(st));
    if (h_wide == 0) {
      NSS_HIP(hipMalloc(&c16, sizeof(uint16_t) * (size_t(A.nnz) + 8)));
      NSS_HIP(hipMemsetAsync(c16, 0, sizeof(uint16_t) * (size_t(A.nnz) + 8), st));
      hipLaunchKernelGGL(col_pack16_kernel, dim3(A.nblk), dim3(kBlock), 0, st, A.nblk, A.rowblk, A.rowptr, A.col,
                         A.chunk, base, c16);
      NSS_CHECK_LAUNCH();
      NSS_HIP(hipStreamSynchronize(st));
      A.col16 = c16;
      A.blkbase = base;
      c16 = nullptr;
      base = nullptr;
    }
  } catch (...) {
    (void)hipFree(base);
    (void)hipFree(wide);
    (void)hipFree(c16);
    throw;
  }
  (void)hipFree(base);
  (void)hipFree(wide);
  (void)hipFree(c16);
}

// fixed-width copy: kDirectWidth (column, value) pairs per row, column -1 where the row is shorter
__global__ __launch_bounds__(kBlock) void ell_build_kernel(int32_t m, const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ col,
                                                            const double* __restrict__ val, int32_t* __restrict__ ecol,
                                                            double* __restrict__ eval, int32_t* __restrict__ wide) {
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  for (int64_t r = int64_t(blockIdx.x) * kBlock + threadIdx.x; r < m; r += stride) {
    const int s = rowptr[r], e = rowptr[r + 1];
    if (e - s > kDirectWidth) atomicOr(wide, 1);
    for (int j = 0; j < kDirectWidth; ++j) {
      const bool has = s + j < e;
      ecol[r * kDirectWidth + j] = has ? col[s + j] : -1;
      eval[r * kDirectWidth + j] = has ? val[s + j] : 0.0;
    }
  }
}

static void direct_rows(nss_csr_s& A, hipStream_t st) {
#if NSS_DIRECT_ROWS
  if (A.m < g_direct_min_rows || A.nnz > int64_t(kDirectWidth) * A.m) return;
  int32_t* ecol = nullptr;
  double* eval = nullptr;
  int32_t* wide = nullptr;
  try {
    NSS_HIP(hipMalloc(&ecol, sizeof(int32_t) * size_t(A.m) * kDirectWidth));
    NSS_HIP(hipMalloc(&eval, sizeof(double) * size_t(A.m) * kDirectWidth));
    NSS_HIP(hipMalloc(&wide, sizeof(int32_t)));
    NSS_HIP(hipMemsetAsync(wide, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(ell_build_kernel, dim3(stream_grid(A.m, kBlock * 4)), dim3(kBlock), 0, st, A.m, A.rowptr, A.col,
                       A.val, ecol, eval, wide);
    NSS_CHECK_LAUNCH();
    int32_t h_wide = 1;
    NSS_HIP(hipMemcpyAsync(&h_wide, wide, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    NSS_HIP(hipStreamSynchronize(st));
    if (h_wide == 0) {
      A.ell_col = ecol;
      A.ell_val = eval;
      ecol = nullptr;
      eval = nullptr;
    }
  } catch (...) {
    (void)hipFree(ecol);
    (void)hipFree(eval);
    (void)hipFree(wide);
    throw;
  }
  (void)hipFree(ecol);
  (void)hipFree(eval);
  (void)hipFree(wide);
#else
  (void)A;
  (void)st;
#endif
}

// Both 16-bit forms the matrix admits (the kernels pick per launch: staged where the operand is one stored
// vector, the window form otherwise), then one index per column run where the matrix is made of runs.
void compress_columns(nss_csr_s& A, hipStream_t st) {
#if NSS_COL16
  if (A.nnz == 0 || A.nblk == 0) return;
  stage_columns(A, st);
  window_columns(A, st);
  group_columns(A, st);
  direct_rows(A, st);
  if (!A.ell_col) build_dispatch(A, st);
#else
  (void)A;
  (void)st;
#endif
}

__global__ __launch_bounds__(kBlock) void csr_diag_kernel(int32_t m, const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ col,
                                                           const double* __restrict__ val, double* __restrict__ d) {
  const int r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= m) return;
  double v = 0.0;
  for (int p = rowptr[r]; p < rowptr[r + 1]; ++p)
    if (col[p] == r) v = val[p];
  d[r] = v;
}

}  // namespace nss

using namespace nss;

extern "C" {

int nss_csr_create(int32_t nrows, int32_t ncols, int64_t nnz, const int32_t* h_rowptr, const int32_t* h_col,
                   const double* h_val, nss_csr_t* out) {
  return nss_csr_create_cuts(nrows, ncols, nnz, h_rowptr, h_col, h_val, 0, nullptr, out);
}

int nss_csr_create_cuts(int32_t nrows, int32_t ncols, int64_t nnz, const int32_t* h_rowptr, const int32_t* h_col,
                        const double* h_val, int32_t ncuts, const int32_t* h_cuts, nss_csr_t* out) {
  return guarded([&] {
    NSS_REQUIRE(ncuts >= 0 && (ncuts == 0 || h_cuts != nullptr), "csr_create: bad cuts");
    for (int i = 0; i < ncuts; ++i)
      NSS_REQUIRE(h_cuts[i] >= 0 && h_cuts[i] <= nrows && (i == 0 || h_cuts[i] >= h_cuts[i - 1]),
                  "csr_create: cuts must be ascending row positions");
    NSS_REQUIRE(out != nullptr, "csr_create: out is NULL");
    NSS_REQUIRE(nrows >= 0 && ncols >= 0 && nnz >= 0, "csr_create: negative size");
    NSS_REQUIRE(nnz < (int64_t(1) << 31), "csr_create: nnz must fit int32 offsets");
    NSS_REQUIRE(h_rowptr != nullptr, "csr_create: rowptr is NULL");
    NSS_REQUIRE(h_rowptr[0] == 0 && int64_t(h_rowptr[nrows]) == nnz, "csr_create: rowptr does not span nnz");
    for (int32_t r = 0; r < nrows; ++r)
      NSS_REQUIRE(h_rowptr[r + 1] >= h_rowptr[r], "csr_create: rowptr not monotone");
    for (int64_t p = 0; p < nnz; ++p)
      NSS_REQUIRE(h_col[p] >= 0 && h_col[p] < ncols, "csr_create: column index out of range");
    nss_csr_s* A = new nss_csr_s;
    try {
      A->m = nrows;
      A->n = ncols;
      A->nnz = nnz;
      std::vector<int32_t> blk;
      plan_row_blocks(nrows, nnz, h_rowptr, &A->rg, &A->chunk, blk, h_cuts, ncuts);
      A->cuts.assign(h_cuts, h_cuts + ncuts);
      A->nblk = int32_t(blk.size()) - 1;
      NSS_HIP(hipMalloc(&A->rowptr, sizeof(int32_t) * (size_t(nrows) + 1)));
      NSS_HIP(hipMalloc(&A->col, sizeof(int32_t) * (nnz + 4)));   // +4: paired loads may touch one entry past the end
      NSS_HIP(hipMalloc(&A->val, sizeof(double) * (nnz + 4)));
      NSS_HIP(hipMalloc(&A->rowblk, sizeof(int32_t) * blk.size()));
      NSS_HIP(hipMemcpy(A->rowptr, h_rowptr, sizeof(int32_t) * (size_t(nrows) + 1), hipMemcpyHostToDevice));
      NSS_HIP(hipMemset(A->col, 0, sizeof(int32_t) * (nnz + 4)));
      NSS_HIP(hipMemset(A->val, 0, sizeof(double) * (nnz + 4)));
      if (nnz > 0) {
        NSS_HIP(hipMemcpy(A->col, h_col, sizeof(int32_t) * nnz, hipMemcpyHostToDevice));
        NSS_HIP(hipMemcpy(A->val, h_val, sizeof(double) * nnz, hipMemcpyHostToDevice));
      }
      NSS_HIP(hipMemcpy(A->rowblk, blk.data(), sizeof(int32_t) * blk.size(), hipMemcpyHostToDevice));
      compress_columns(*A, nullptr);
    } catch (...) {
      nss_csr_destroy(A);
      throw;
    }
    *out = A;
  });
}

int nss_csr_destroy(nss_csr_t a) {
  return guarded([&] {
    if (!a) return;
    (void)hipFree(a->rowptr);
    (void)hipFree(a->col);
    (void)hipFree(a->val);
    (void)hipFree(a->rowblk);
    (void)hipFree(a->col16);
    (void)hipFree(a->blkbase);
    (void)hipFree(a->blkseg);
    (void)hipFree(a->blkdisp);
    (void)hipFree(a->pos16);
    (void)hipFree(a->ell_col);
    (void)hipFree(a->ell_val);
    (void)hipFree(a->jb_first);
    (void)hipFree(a->jb_order);
    if (a->fw_col != a->ell_col) {
      (void)hipFree(a->fw_col);
      (void)hipFree(a->fw_val);
    }
    delete a;
  });
}

// new launch plan with at most `products` products per row block; every derived column stream is rebuilt
static void replan(nss_csr_s& A, int products, int max_rows = 0, const uint8_t* row_pos = nullptr,
                   std::vector<int32_t>* blk_out = nullptr) {
  NSS_HIP(hipDeviceSynchronize());                       // no kernel may still read the arrays that go away
  std::vector<int32_t> h_rowptr(size_t(A.m) + 1);
  NSS_HIP(hipMemcpy(h_rowptr.data(), A.rowptr, sizeof(int32_t) * h_rowptr.size(), hipMemcpyDeviceToHost));
  std::vector<int32_t> blk;
  int32_t rg = 1, chunk = kChunk;
  plan_row_blocks(A.m, A.nnz, h_rowptr.data(), &rg, &chunk, blk, A.cuts.data(), int(A.cuts.size()), products, max_rows,
                  row_pos);
  if (A.fw_col != A.ell_col) {                            // (an own fixed-width copy; an alias of ell_col goes with it below)
    (void)hipFree(A.fw_col);
    (void)hipFree(A.fw_val);
  }
  A.fw_col = nullptr;
  A.fw_val = nullptr;
  A.fw_state = 0;
  (void)hipFree(A.jb_first);                             // a plan around Jacobi blocks ends with the plan
  (void)hipFree(A.jb_order);
  A.jb_first = nullptr;
  A.jb_order = nullptr;
  A.jb_serial = 0;
  int32_t* rowblk = nullptr;
  NSS_HIP(hipMalloc(&rowblk, sizeof(int32_t) * blk.size()));
  NSS_HIP(hipMemcpy(rowblk, blk.data(), sizeof(int32_t) * blk.size(), hipMemcpyHostToDevice));
  for (void* p : {(void*)A.rowblk, (void*)A.col16, (void*)A.blkbase, (void*)A.blkseg, (void*)A.pos16, (void*)A.ell_col,
                  (void*)A.ell_val, (void*)A.blkdisp})
    (void)hipFree(p);
  A.blkdisp = nullptr;
  A.disp_period = 0.0;
  A.disp_planes = 0;
  A.rowblk = rowblk;
  A.col16 = nullptr;
  A.blkbase = nullptr;
  A.blkseg = nullptr;
  A.pos16 = nullptr;
  A.ell_col = nullptr;
  A.ell_val = nullptr;
  A.gb = 1;
  A.pair_ok = false;
  A.rg = rg;                                             // (unchanged: the lanes-per-row rule does not see `products`)
  A.nblk = int32_t(blk.size()) - 1;
  A.blk_products = products;
  compress_columns(A, nullptr);
  if (blk_out) blk_out->swap(blk);
}

extern "C++" {
namespace nss {
bool fixed_width_copy(nss_csr_s& A) {
  if (A.fw_state != 0) return A.fw_state > 0;
  A.fw_state = -1;
  if (A.ell_col) {
    A.fw_col = A.ell_col;
    A.fw_val = A.ell_val;
    A.fw_state = 1;
    return true;
  }
  if (A.m == 0 || A.nnz > int64_t(kDirectWidth) * A.m) return false;
  int32_t* ecol = nullptr;
  double* eval = nullptr;
  int32_t* wide = nullptr;
  int32_t h_wide = 1;
  try {
    NSS_HIP(hipMalloc(&ecol, sizeof(int32_t) * size_t(A.m) * kDirectWidth));
    NSS_HIP(hipMalloc(&eval, sizeof(double) * size_t(A.m) * kDirectWidth));
    NSS_HIP(hipMalloc(&wide, sizeof(int32_t)));
    NSS_HIP(hipMemset(wide, 0, sizeof(int32_t)));
    hipLaunchKernelGGL(ell_build_kernel, dim3(stream_grid(A.m, kBlock * 4)), dim3(kBlock), 0, nullptr, A.m, A.rowptr, A.col,
                       A.val, ecol, eval, wide);
    NSS_CHECK_LAUNCH();
    NSS_HIP(hipMemcpy(&h_wide, wide, sizeof(int32_t), hipMemcpyDeviceToHost));
  } catch (...) {
    (void)hipFree(ecol);
    (void)hipFree(eval);
    (void)hipFree(wide);
    throw;
  }
  (void)hipFree(wide);
  if (h_wide != 0) {
    (void)hipFree(ecol);
    (void)hipFree(eval);
    return false;
  }
  A.fw_col = ecol;
  A.fw_val = eval;
  A.fw_state = 1;
  return true;
}

void replan_row_blocks(nss_csr_s& A, int products) {
  if (A.blk_products != products) replan(A, products);
}
}  // namespace nss
}

int nss_csr_plan_for_pairs(nss_csr_t a, int32_t* pair_staged) {
  return guarded([&] {
    NSS_REQUIRE(a != nullptr, "csr_plan_for_pairs: NULL matrix");
    // only matrices the single-operand kernels stage, and not the two-entry-per-row ones of the row-per-lane kernel
    if (a->blkseg && !a->pair_ok && !a->ell_col && a->blk_products > kChunk / 2) {
      const int before = a->blk_products;
      replan(*a, kChunk / 2);
      if (!a->pair_ok) replan(*a, before);               // wide operators: shorter blocks do not help; back to the full plan
    }
    if (pair_staged) *pair_staged = (a->blkseg && a->pair_ok) ? 1 : 0;
  });
}

int nss_csr_plan_for_blocks(nss_csr_t a, nss_bjac_t j, int32_t* planned) {
  return guarded([&] {
    NSS_REQUIRE(a != nullptr && j != nullptr, "csr_plan_for_blocks: NULL argument");
    static_assert(kBlockRows >= kDirectRows, "a row block of the row-per-lane kernel must fit the LDS copy of its results");
    if (planned) *planned = 0;
    if (a->jb_first && a->jb_serial == j->serial) {      // already planned around this handle
      if (planned) *planned = 1;
      return;
    }
    // what the fused epilogue applies: symmetric inverse blocks over runs of consecutive dofs that tile the rows in order
    if (j->n != a->m || !j->run || !j->inv_sym || j->gs_mat || j->n_uncovered != 0 || j->nblocks == 0) return;
    if (!fuse_block_jacobi_wanted(a->m)) return;         // (large systems: the stand-alone apply is faster; no re-plan)
    std::vector<int32_t> run(size_t(j->nblocks));
    NSS_HIP(hipMemcpy(run.data(), j->run, sizeof(int32_t) * run.size(), hipMemcpyDeviceToHost));
    // the blocks in row order (callers number them as they like: the lines of one velocity component after the other)
    std::vector<int32_t> order(run.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = int32_t(i);
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return (run[size_t(x)] >> 5) < (run[size_t(y)] >> 5); });
    std::vector<uint8_t> row_pos(size_t(a->m), 0);
    int64_t next = 0;
    for (int32_t jb : order) {
      const int32_t first = run[size_t(jb)] >> 5, len = run[size_t(jb)] & 31;
      if (first != next || len < 1 || first + len > a->m) return;          // the blocks do not tile the rows: no fusion
      for (int c = 0; c < len; ++c) row_pos[size_t(first + c)] = uint8_t(c);
      next = first + len;
    }
    if (next != a->m) return;
    std::vector<int32_t> blk;
    replan(*a, a->blk_products, kBlockRows, row_pos.data(), &blk);
    // position (in row order) of the first Jacobi block of every row block (row blocks start at block starts)
    std::vector<int32_t> first_of(blk.size());
    size_t b = 0;
    for (size_t i = 0; i < blk.size(); ++i) {
      while (b < order.size() && (run[size_t(order[b])] >> 5) < blk[i]) ++b;
      // (a forced cut of the matrix inside a Jacobi block, or one over-long row: the plan stands, without the fusion)
      if (!(i + 1 == blk.size() ? b == order.size() : (b < order.size() && (run[size_t(order[b])] >> 5) == blk[i]))) return;
      if (i > 0 && blk[i] - blk[i - 1] > kBlockRows) return;
      first_of[i] = int32_t(b);
    }
    NSS_HIP(hipMalloc(&a->jb_first, sizeof(int32_t) * first_of.size()));
    NSS_HIP(hipMemcpy(a->jb_first, first_of.data(), sizeof(int32_t) * first_of.size(), hipMemcpyHostToDevice));
    NSS_HIP(hipMalloc(&a->jb_order, sizeof(int32_t) * order.size()));
    NSS_HIP(hipMemcpy(a->jb_order, order.data(), sizeof(int32_t) * order.size(), hipMemcpyHostToDevice));
    a->jb_serial = j->serial;
    if (planned) *planned = 1;
  });
}

int nss_csr_dispatch_mode(int32_t planes, int32_t min_period, int32_t run) {
  return guarded([&] {
    NSS_REQUIRE(planes >= -2 && planes <= 64, "csr_dispatch_mode: planes must be -2 (library default), -1 (automatic), 0 (natural order) or 1 .. 64");
    g_dispatch_planes = planes == -2 ? NSS_DISPATCH_DEFAULT : planes;
    g_dispatch_min_period = min_period > 0 ? min_period : NSS_DISPATCH_MIN_PERIOD;
    g_dispatch_run = run > 0 ? run : NSS_DISPATCH_RUN;
  });
}

int nss_csr_dispatch_info(nss_csr_t a, double* period, int32_t* planes) {
  return guarded([&] {
    NSS_REQUIRE(a != nullptr, "csr_dispatch_info: NULL matrix");
    if (period) *period = a->blkdisp ? a->disp_period : 0.0;
    if (planes) *planes = a->blkdisp ? a->disp_planes : 0;
  });
}

int nss_csr_pair_mode(int32_t mode) {
  return guarded([&] {
    NSS_REQUIRE(mode >= -1 && mode <= 1, "csr_pair_mode: -1 (automatic), 0 (never) or 1");
    g_pair_mode = mode;
  });
}

int nss_csr_pair_staged(nss_csr_t a, int32_t* pair_staged) {
  return guarded([&] {
    NSS_REQUIRE(a != nullptr && pair_staged != nullptr, "csr_pair_staged: NULL argument");
    *pair_staged = (a->blkseg && a->pair_ok) ? 1 : 0;
  });
}

int nss_csr_spmv_f64(nss_csr_t a, double alpha, const double* x, double beta, double* y, nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(a != nullptr, "csr_spmv: NULL matrix");
    NSS_REQUIRE(x != y, "csr_spmv: x must not alias y");
    launch_csr_stream(*a, x, EpiAxpby{alpha, beta, y}, as_stream(stream));
  });
}

int nss_csr_index_width(nss_csr_t a, int32_t* bytes) {
  return guarded([&] {
    NSS_REQUIRE(a != nullptr && bytes != nullptr, "csr_index_width: NULL argument");
    *bytes = (a->col16 || a->pos16) ? 2 : 4;
  });
}

int nss_csr_index_group(nss_csr_t a, int32_t* entries_per_index) {
  return guarded([&] {
    NSS_REQUIRE(a != nullptr && entries_per_index != nullptr, "csr_index_group: NULL argument");
    *entries_per_index = (a->col16 || a->pos16) ? a->gb : 1;
  });
}

int nss_csr_direct_rows_threshold(int64_t min_rows) {
  return guarded([&] {
    NSS_REQUIRE(min_rows >= -1, "csr_direct_rows_threshold: -1 (default) or a row count");
    g_direct_min_rows = min_rows < 0 ? int64_t(NSS_DIRECT_MIN_ROWS) : min_rows;
  });
}

int nss_csr_operand_form(nss_csr_t a, int32_t* form) {
  return guarded([&] {
    NSS_REQUIRE(a != nullptr && form != nullptr, "csr_operand_form: NULL argument");
    *form = a->ell_col ? 3 : a->idx_mode();
  });
}

int nss_csr_info(nss_csr_t a, int32_t* nrows, int32_t* ncols, int64_t* nnz, int32_t* nblocks,
                 int32_t* lanes_per_row, int64_t* algorithmic_bytes) {
  return guarded([&] {
    NSS_REQUIRE(a != nullptr, "csr_info: NULL matrix");
    if (nrows) *nrows = a->m;
    if (ncols) *ncols = a->n;
    if (nnz) *nnz = a->nnz;
    if (nblocks) *nblocks = a->nblk;
    if (lanes_per_row) *lanes_per_row = a->rg;
    // bytes one y = A x launch has to move: the value and the column stream THIS matrix is stored with
    // (2-byte staged positions + a 128-byte run table per row block, 2-byte window-relative columns + 16 window
    // bases per row block, or 4-byte columns -- what the plain SpMV of this matrix streams), the row
    // pointers, x once and y once.  (The CSR fp64/int32 textbook figure is 12 nnz + ...; pricing a
    // launch that streams 10 bytes per entry at 12 would overstate its bandwidth.)
    if (algorithmic_bytes && a->ell_col)      // fixed-width copy: 12 bytes per slot, no row pointers
      *algorithmic_bytes = int64_t(12) * nss::kDirectWidth * a->m + 8 * int64_t(a->n) + 8 * int64_t(a->m);
    else if (algorithmic_bytes)
      *algorithmic_bytes = ((a->col16 || a->pos16) ? 8 * a->nnz + 2 * (a->nnz / a->gb) +
                                                           int64_t(4) * (a->blkseg ? kSegWords : kWindows) * a->nblk
                                                     : 12 * a->nnz) +
                           4 * (int64_t(a->m) + 1) + 8 * int64_t(a->n) + 8 * int64_t(a->m);
  });
}

int nss_csr_row_blocks(nss_csr_t a, int32_t* h_out, int64_t cap) {
  return guarded([&] {
    NSS_REQUIRE(a != nullptr && h_out != nullptr, "csr_row_blocks: NULL argument");
    NSS_REQUIRE(cap >= int64_t(a->nblk) + 1, "csr_row_blocks: output too small");
    NSS_HIP(hipMemcpy(h_out, a->rowblk, sizeof(int32_t) * (size_t(a->nblk) + 1), hipMemcpyDeviceToHost));
  });
}

int nss_csr_diagonal(nss_csr_t a, double* diag_dev, nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(a != nullptr, "csr_diagonal: NULL matrix");
    if (a->m == 0) return;
    hipLaunchKernelGGL(csr_diag_kernel, dim3((a->m + kBlock - 1) / kBlock), dim3(kBlock), 0, as_stream(stream),
                       a->m, a->rowptr, a->col, a->val, diag_dev);
    NSS_CHECK_LAUNCH();
  });
}

}  // extern "C"
