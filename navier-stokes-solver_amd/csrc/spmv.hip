// CSR matrix handles (upload, launch plan) and the plain alpha/beta SpMV entry point.
#include "csr_stream.h"

#include <algorithm>
#include <climits>
#include <vector>

namespace nss {

#ifndef NSS_PLAN_FILL_CHIP
#define NSS_PLAN_FILL_CHIP 1
#endif
constexpr int kMinRowBlocks = 2048;   // 256 CUs x 8 resident workgroups

void plan_row_blocks(int32_t m, int64_t nnz, const int32_t* rowptr, int32_t* rg_out, int32_t* chunk_out,
                     std::vector<int32_t>& blk, const int32_t* cuts, int ncuts) {
  const double mean = m > 0 ? double(nnz) / double(m) : 0.0;
  const int chunk = mean >= double(kLongRowMean) ? kChunkLong : kChunk;
  *chunk_out = chunk;
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
    passes = mean > 0.0 ? int(double(chunk) / (mean * rows_per_pass)) : kMaxRowsPerBlock / rows_per_pass;
    passes = std::max(1, passes);
#if NSS_PLAN_FILL_CHIP
    // small matrices (a slab of a partitioned system, the small configs): rather more, shorter row
    // blocks than fewer than ~8 workgroups per CU
    while (passes > 1 && int64_t(m) / (int64_t(passes) * rows_per_pass) < kMinRowBlocks) --passes;
#endif
  }
  const int row_cap = std::min(kMaxRowsPerBlock, passes * rows_per_pass);
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
      if (acc + len > chunk) break;
      acc += len;
      ++r;
    }
    if (r == start) ++r;  // a single row longer than the chunk gets a block of its own
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

// Largest gb in 16 .. 2 for which the matrix is made of aligned runs of gb consecutive columns: keep one
// 16-bit index per run (see nss_csr_s::gb).  Every row block starts at a row start, hence at a multiple of gb.
static void group_columns(nss_csr_s& A, hipStream_t st) {
#if NSS_COL_GROUPS
  if (!A.col16 || A.nnz < 16) return;
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
  uint16_t* packed = nullptr;
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
      const int64_t ngroups = A.nnz / gb;
      NSS_HIP(hipMalloc(&packed, sizeof(uint16_t) * (size_t(ngroups) + 8)));
      NSS_HIP(hipMemsetAsync(packed, 0, sizeof(uint16_t) * (size_t(ngroups) + 8), st));
      hipLaunchKernelGGL(col_group_pack_kernel, dim3(stream_grid(ngroups, kBlock * 4)), dim3(kBlock), 0, st, ngroups, gb,
                         A.col16, packed);
      NSS_CHECK_LAUNCH();
      NSS_HIP(hipStreamSynchronize(st));
      (void)hipFree(A.col16);
      A.col16 = packed;
      packed = nullptr;
      A.gb = gb;
      break;
    }
  } catch (...) {
    (void)hipFree(bad);
    (void)hipFree(packed);
    throw;
  }
  (void)hipFree(bad);
#else
  (void)A;
  (void)st;
#endif
}

void compress_columns(nss_csr_s& A, hipStream_t st) {
#if NSS_COL16
  if (A.nnz == 0 || A.nblk == 0) return;
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
      group_columns(A, st);
    }
  } catch (...) {
    (void)hipFree(base);
    (void)hipFree(wide);
    (void)hipFree(c16);
    throw;
  }
  (void)hipFree(base);
  (void)hipFree(wide);
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
    delete a;
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
    *bytes = a->col16 ? 2 : 4;
  });
}

int nss_csr_index_group(nss_csr_t a, int32_t* entries_per_index) {
  return guarded([&] {
    NSS_REQUIRE(a != nullptr && entries_per_index != nullptr, "csr_index_group: NULL argument");
    *entries_per_index = a->col16 ? a->gb : 1;
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
    // (2-byte window-relative columns + 16 window bases per row block, or 4-byte columns), the row
    // pointers, x once and y once.  (The CSR fp64/int32 textbook figure is 12 nnz + ...; pricing a
    // launch that streams 10 bytes per entry at 12 would overstate its bandwidth.)
    if (algorithmic_bytes)
      *algorithmic_bytes = (a->col16 ? 8 * a->nnz + 2 * (a->nnz / a->gb) + int64_t(4) * kWindows * a->nblk : 12 * a->nnz) +
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
