// CSR-stream SpMV kernel template for gfx950 (fp64 values, int32 indices).
//
// The Stokes blocks have short rows (7 / 6 / 2 non-zeros for A / B / B^T on the MAC
// grid, ~25-84 for HDG-like operators), so a row-per-wave kernel would idle most of
// its 64 lanes.  Instead a 256-thread workgroup owns a contiguous *row block* whose
// non-zeros fit one LDS chunk:
//   phase 1  every lane streams val[] / col[] with unit stride (fully coalesced HBM
//            reads, 8 independent loads in flight per lane), gathers x[col] (served by
//            L2 / Infinity Cache for banded operators) and stages the products in LDS;
//   phase 2  each row is reduced from LDS by RG lanes (RG = 1 ... 64, a power of two chosen from
//            the mean row length) with a __shfl_xor butterfly, in a fixed order -- no
//            atomics, bit-reproducible;
//   epilogue a functor consumes (row, A x) -- plain alpha/beta update or the fused
//            vector updates + dot partials of the Krylov loops -- so y is written once
//            and the extra vectors are read while the row is still in registers.
// Row blocks are precomputed on the host when the matrix is uploaded; one workgroup per row
// block.  The
// blockIdx -> row-block map is XCD-aware: workgroups b and b+8 share an XCD (round-robin
// dispatch), so XCD i walks its own contiguous eighth of the rows and its private 4 MiB
// L2 keeps one window of x instead of all eight L2s caching the same window.
#pragma once

#include <type_traits>
#include <vector>

#include "nss_common.h"

namespace nss {

// Non-temporal loads for the col/val streams: they are read exactly once, and keeping them
// out of the 4-MiB XCD L2s leaves room for the gathered x window.  Measured on the 1e7-DoF
// case (interleaved A/B, one box): plain A SpMV 0.173 -> 0.163 ms, BPCG iteration +4 %.
#ifndef NSS_STREAM_NT
#define NSS_STREAM_NT 1
#endif

#ifndef NSS_STREAM_PREFETCH
#define NSS_STREAM_PREFETCH 1   // request row bounds + epilogue operands before the matrix stream
#endif

#ifndef NSS_STREAM_VEC2
#define NSS_STREAM_VEC2 0   // 1: 16-byte (val) / 8-byte (col) loads, two consecutive entries per lane
#endif

#ifndef NSS_CHUNK
#define NSS_CHUNK 2048
#endif
constexpr int kChunk = NSS_CHUNK;       // products staged per workgroup: 16 KiB of LDS
// Matrices with long rows (mean >= kLongRowMean non-zeros) stage twice as many: a row block then
// holds enough rows to keep the phase-2 lanes busy and 16 loads per lane are in flight (K2 +5 % at
// 82 non-zeros per row, +1.5 % at 34); short-row matrices lose occupancy to the 32 KiB and stay at kChunk.
constexpr int kChunkLong = 2 * kChunk;
#ifndef NSS_LONG_ROW_MEAN
#define NSS_LONG_ROW_MEAN 32
#endif
constexpr int kLongRowMean = NSS_LONG_ROW_MEAN;
constexpr int kMaxRowsPerBlock = 2048;  // bound for blocks of empty / very short rows
constexpr int kXcds = 8;
constexpr int kWindows = 16;          // column windows per row block of the 16-bit index stream
constexpr int kWindowBits = 12;      // 4096 columns per window

struct CsrView {
  const int32_t* __restrict__ rowblk;
  const int32_t* __restrict__ rowptr;
  const int32_t* __restrict__ col;
  const uint16_t* __restrict__ col16;   // (window << 12 | offset) per entry, or NULL (see nss_csr_s)
  const int32_t* __restrict__ blkbase;  // kWindows window bases per row block
  const double* __restrict__ val;
  int32_t blk0;     // first row block of this launch (sub-range launches: interior / boundary)
  int32_t nblk;     // row blocks in this launch
  int32_t per_xcd;  // ceil(nblk / 8)
};

}  // namespace nss

struct nss_csr_s {
  int32_t m = 0, n = 0;
  int64_t nnz = 0;
  int32_t* rowptr = nullptr;
  int32_t* col = nullptr;
  double* val = nullptr;
  int32_t* rowblk = nullptr;
  int32_t nblk = 0;
  int32_t rg = 1;
  int32_t chunk = nss::kChunk;   // products per row block: kChunk or kChunkLong (plan_row_blocks)
  // Compressed column stream: when the columns of every row block fall into at most 16 aligned
  // windows of 4096 columns (grid operators: a row block touches its own grid plane and the two
  // neighbouring ones -- a few narrow clusters far apart) the kernel streams 2 bytes per entry,
  // 4 bits of window number + 12 bits of offset, decoded through the block's 16 window bases
  // (held in LDS), instead of the 4-byte index: 2 bytes per non-zero less HBM traffic, same
  // products in the same order.  `col` is kept for the set-up kernels and rows longer than a chunk.
  uint16_t* col16 = nullptr;
  int32_t* blkbase = nullptr;
  // launch view of the row blocks [b0, b1)
  nss::CsrView view(int b0, int b1) const {
    return nss::CsrView{rowblk, rowptr, col, col16, blkbase, val, b0, b1 - b0,
                        (b1 - b0 + nss::kXcds - 1) / nss::kXcds};
  }
  // one workgroup per row block, padded to a multiple of the XCD count
  static int grid(int count) { return ((count + nss::kXcds - 1) / nss::kXcds) * nss::kXcds; }
};

namespace nss {

// Build col16 / blkbase of a matrix whose device arrays and launch plan are complete (spmv.hip).
void compress_columns(nss_csr_s& A, hipStream_t st);

// Launch plan of a CSR matrix: lanes per row (*rg_out) and the row-block boundaries (spmv.hip).
void plan_row_blocks(int32_t m, int64_t nnz, const int32_t* rowptr, int32_t* rg_out, int32_t* chunk_out,
                     std::vector<int32_t>& blk, const int32_t* cuts = nullptr, int ncuts = 0);

// Epi interface:
//   __device__ void row(int r, double ax);          // called once per row by one lane
// or, to have the row's read-only operands requested before the matrix stream,
//   struct Pre;  __device__ Pre fetch(int r) const;
//   __device__ void row(int r, double ax, const Pre&);
//   __device__ void finish(int b, double* lds);     // called by all threads at the end; b = row
//                                                   // block of this workgroup (-1: padding), the
//                                                   // slot of its dot partial
//   __device__ bool skip() const;                   // e.g. solver already converged
template <class E, class = void>
struct EpiPre {
  struct type {};
  static __device__ type fetch(const E&, int) { return type{}; }
  static __device__ void row(E& e, int r, double ax, const type&) { e.row(r, ax); }
};
template <class E>
struct EpiPre<E, std::void_t<typename E::Pre>> {
  using type = typename E::Pre;
  static __device__ type fetch(const E& e, int r) { return e.fetch(r); }
  static __device__ void row(E& e, int r, double ax, const type& p) { e.row(r, ax, p); }
};

template <int RG, class Epi, bool C16 = false, int CH = kChunk>
__global__ __launch_bounds__(kBlock) void csr_stream_kernel(CsrView a, const double* __restrict__ x, Epi epi) {
  __shared__ double prod[CH];
  __shared__ double red[kBlock / kWave];
  __shared__ int32_t window[kWindows];
  if (epi.skip()) return;
  const int tid = threadIdx.x;
  // XCD-aware map: workgroups with equal (blockIdx & 7) share an XCD; XCD i owns the i-th
  // contiguous eighth of the row blocks.  One row block per workgroup: a striding
  // (persistent) loop around this body measured 9 % slower on the A SpMV (extra barrier, and
  // the hardware dispatcher balances the tail better).
  const int lb = (blockIdx.x & (kXcds - 1)) * a.per_xcd + (blockIdx.x >> 3);
  const int b = lb < a.nblk ? a.blk0 + lb : -1;
  if (b >= 0) {
    const int r0 = a.rowblk[b];
    const int r1 = a.rowblk[b + 1];
    const int p0 = a.rowptr[r0];
    const int cnt = a.rowptr[r1] - p0;
#if NSS_STREAM_PREFETCH
    // Row bounds and epilogue operands of this lane's first phase-2 row are requested now, so
    // their HBM latency overlaps the matrix stream instead of following the barrier.
    const int rf = r0 + tid / RG;
    const bool has_first = rf < r1;
    const int rf_s = has_first ? a.rowptr[rf] : 0;
    const int rf_e = has_first ? a.rowptr[rf + 1] : 0;
    typename EpiPre<Epi>::type pre0 = has_first ? EpiPre<Epi>::fetch(epi, rf) : typename EpiPre<Epi>::type{};
#endif
    if (cnt <= CH) {
      // ---- phase 1: coalesced stream of (col, val), gather x, stage products ---------
#if NSS_STREAM_VEC2
      // two consecutive entries per lane from an even-aligned base: 16-byte val / 8-byte col loads
      constexpr int kPer = CH / (2 * kBlock);
      const int pa = p0 & ~1;
      const int lead = p0 - pa;                  // 0 or 1 entries in front of the row block
      const int span = cnt + lead;
      typedef int int2v __attribute__((ext_vector_type(2)));
      typedef double double2v __attribute__((ext_vector_type(2)));
      int2v c[kPer + 1];
      double2v v[kPer + 1];
#pragma unroll
      for (int k = 0; k <= kPer; ++k) {
        const int e = 2 * (tid + k * kBlock);     // entry offset from pa
        const bool live = e < span;
        c[k] = live ? __builtin_nontemporal_load(reinterpret_cast<const int2v*>(a.col + pa + e)) : int2v{0, 0};
        v[k] = live ? __builtin_nontemporal_load(reinterpret_cast<const double2v*>(a.val + pa + e)) : double2v{0.0, 0.0};
      }
#pragma unroll
      for (int k = 0; k <= kPer; ++k) {
        const int e = 2 * (tid + k * kBlock);
        const int i0 = e - lead, i1 = e + 1 - lead;          // LDS slots of the two entries
        if (i0 >= 0 && i0 < cnt) prod[i0] = v[k].x * x[c[k].x];
        if (i1 < cnt && e < span) prod[i1] = v[k].y * x[c[k].y];
      }
#else
      constexpr int kPer = CH / kBlock;
      int32_t c[kPer];
      double v[kPer];
      uint16_t c16[kPer];
      if (C16) {
        if (tid < kWindows) window[tid] = a.blkbase[b * kWindows + tid];
      }
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        const int i = tid + k * kBlock;
        const bool live = i < cnt;
#if NSS_STREAM_NT
        if (C16) c16[k] = live ? __builtin_nontemporal_load(&a.col16[p0 + i]) : uint16_t(0);
        else c[k] = live ? __builtin_nontemporal_load(&a.col[p0 + i]) : 0;
        v[k] = live ? __builtin_nontemporal_load(&a.val[p0 + i]) : 0.0;
#else
        if (C16) c16[k] = live ? a.col16[p0 + i] : uint16_t(0);
        else c[k] = live ? a.col[p0 + i] : 0;
        v[k] = live ? a.val[p0 + i] : 0.0;
#endif
      }
      if (C16) {
        __syncthreads();                                   // window bases in LDS
#pragma unroll
        for (int k = 0; k < kPer; ++k) c[k] = window[c16[k] >> kWindowBits] + int32_t(c16[k] & ((1 << kWindowBits) - 1));
      }
      double xv[kPer];
#pragma unroll
      for (int k = 0; k < kPer; ++k) xv[k] = (tid + k * kBlock < cnt) ? x[c[k]] : 0.0;
#pragma unroll
      for (int k = 0; k < kPer; ++k)
        if (tid + k * kBlock < cnt) prod[tid + k * kBlock] = v[k] * xv[k];
#endif
      __syncthreads();
      // ---- phase 2: per-row reduction from LDS -----------------------------------------
      constexpr int kRowsPerPass = kBlock / RG;
      const int sub = tid % RG;
#if NSS_STREAM_PREFETCH
      for (int r = rf; r < r1; r += kRowsPerPass) {
        const bool first = r == rf;
        const int s = (first ? rf_s : a.rowptr[r]) - p0;
        const int e = (first ? rf_e : a.rowptr[r + 1]) - p0;
        double sum = 0.0;
        for (int j = s + sub; j < e; j += RG) sum += prod[j];
        if (RG > 1) {
#pragma unroll
          for (int off = RG / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off, kWave);
        }
        if (sub == 0) EpiPre<Epi>::row(epi, r, sum, first ? pre0 : EpiPre<Epi>::fetch(epi, r));
      }
#else
      for (int r = r0 + tid / RG; r < r1; r += kRowsPerPass) {
        const int s = a.rowptr[r] - p0;
        const int e = a.rowptr[r + 1] - p0;
        double sum = 0.0;
        for (int j = s + sub; j < e; j += RG) sum += prod[j];
        if (RG > 1) {
#pragma unroll
          for (int off = RG / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off, kWave);
        }
        if (sub == 0) EpiPre<Epi>::row(epi, r, sum, EpiPre<Epi>::fetch(epi, r));
      }
#endif
    } else {
      // ---- one row longer than the LDS chunk: the whole workgroup reduces it ----------
      double acc = 0.0;
      for (int i = tid; i < cnt; i += kBlock) acc = fma(a.val[p0 + i], x[a.col[p0 + i]], acc);
      const double sum = block_sum(acc, red);
      if (tid == 0) EpiPre<Epi>::row(epi, r0, sum, EpiPre<Epi>::fetch(epi, r0));
    }
  }
  epi.finish(b, red);  // one dot partial per row block
}

// rows of the row blocks [b0, b1) (default: all)
template <class Epi>
inline void launch_csr_stream(const nss_csr_s& A, const double* x, const Epi& epi, hipStream_t st, int b0 = 0,
                              int b1 = -1) {
  if (b1 < 0) b1 = A.nblk;
  if (A.m == 0 || b1 <= b0) return;
  const CsrView v = A.view(b0, b1);
  const dim3 grid(nss_csr_s::grid(b1 - b0)), block(kBlock);
#define NSS_LAUNCH_RG(N)                                                                                      \
  case N:                                                                                                      \
    if (A.chunk == kChunkLong) {                                                                               \
      if (A.col16) hipLaunchKernelGGL((csr_stream_kernel<N, Epi, true, kChunkLong>), grid, block, 0, st, v, x, epi);   \
      else hipLaunchKernelGGL((csr_stream_kernel<N, Epi, false, kChunkLong>), grid, block, 0, st, v, x, epi);          \
    } else {                                                                                                   \
      if (A.col16) hipLaunchKernelGGL((csr_stream_kernel<N, Epi, true>), grid, block, 0, st, v, x, epi);       \
      else hipLaunchKernelGGL((csr_stream_kernel<N, Epi, false>), grid, block, 0, st, v, x, epi);              \
    }                                                                                                          \
    break;
  switch (A.rg) {
    NSS_LAUNCH_RG(1) NSS_LAUNCH_RG(2) NSS_LAUNCH_RG(4) NSS_LAUNCH_RG(8) NSS_LAUNCH_RG(16) NSS_LAUNCH_RG(32)
    NSS_LAUNCH_RG(64)
    default: throw Error("csr_stream: bad lanes-per-row in the launch plan");
  }
#undef NSS_LAUNCH_RG
  NSS_CHECK_LAUNCH();
}

// y = alpha * A x + beta * y
struct EpiAxpby {
  double alpha, beta;
  double* __restrict__ y;
  const int32_t* __restrict__ done = nullptr;   // solver stop flag (device), or NULL
  struct Pre { double y = 0.0; };
  __device__ bool skip() const { return done != nullptr && *done != 0; }
  __device__ Pre fetch(int r) const { return Pre{beta != 0.0 ? y[r] : 0.0}; }
  __device__ void row(int r, double ax, const Pre& p) const {
    double t = alpha * ax;
    if (beta != 0.0) t = fma(beta, p.y, t);
    y[r] = t;
  }
  __device__ void finish(int, double*) const {}
};

}  // namespace nss
