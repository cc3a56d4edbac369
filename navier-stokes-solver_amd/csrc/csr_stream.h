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

#ifndef NSS_PREFETCH_ROWS
#define NSS_PREFETCH_ROWS 1     // phase-2 rows per lane whose bounds and epilogue operands are requested up front (RG == 1)
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
  uint32_t gb;      // entries per column group of the 16-bit stream (1: one index per entry)
  uint32_t gstep, sstep;   // kBlock / gb, kBlock % gb: a lane's next entry is kBlock further down the stream
  uint64_t gmagic;  // ceil(2^64 / gb) (gb > 1): p / gb == __umul64hi(p, gmagic) for p < 2^32
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
  // Grouped column stream (block-structured operators: the facet blocks of the HDG-like spaces, ~84
  // non-zeros per row in runs of 12 consecutive columns): when every row length is a multiple of `gb`
  // and every aligned group of `gb` consecutive entries has consecutive columns, col16 holds ONE 16-bit
  // index per group -- 8 + 2/gb bytes per non-zero instead of 10 -- and entry p has column
  // decode(col16[p / gb]) + p % gb.  The value order stays CSR (rows contiguous), so the kernel, its
  // reduction order and its results are unchanged.  gb == 1: one index per entry.
  int32_t gb = 1;
  // launch view of the row blocks [b0, b1)
  nss::CsrView view(int b0, int b1) const {
    const uint64_t magic = gb > 1 ? ~uint64_t(0) / uint64_t(gb) + 1 : 0;    // ceil(2^64 / gb)
    return nss::CsrView{rowblk, rowptr, col, col16, blkbase, val, uint32_t(gb), uint32_t(nss::kBlock / gb),
                        uint32_t(nss::kBlock % gb), magic, b0, b1 - b0,
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
// optional:
//   __device__ bool prologue(double* lds);          // called by ALL threads of the workgroup before the
//                                                   // stream (lds: 32 doubles): e.g. sum the dot partials of
//                                                   // the previous kernel and derive alpha / beta into the
//                                                   // functor's registers; false -> the workgroup returns
//   struct X;  __device__ X xop(const double* x) const;   // operand functor: X{}(c) = value of the SpMV
//                                                   // operand at column c (default: x[c]); lets a kernel
//                                                   // multiply with a vector that is only defined by a
//                                                   // recurrence, e.g. beta * s[c] + w[c], without a pass
//                                                   // that materialises it first
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

struct XPlain {
  const double* __restrict__ x;
  __device__ double operator()(int c) const { return x[c]; }
};
template <class E, class = void>
struct EpiX {
  using type = XPlain;
  static __device__ type get(const E&, const double* x) { return XPlain{x}; }
};
template <class E>
struct EpiX<E, std::void_t<typename E::X>> {
  using type = typename E::X;
  static __device__ type get(const E& e, const double* x) { return e.xop(x); }
};
template <class E, class = void>
struct EpiPrologue {
  static __device__ bool run(E&, double*) { return true; }
};
template <class E>
struct EpiPrologue<E, std::void_t<decltype(std::declval<E&>().prologue(static_cast<double*>(nullptr)))>> {
  static __device__ bool run(E& e, double* lds) { return e.prologue(lds); }
};

constexpr int kRedDoubles = 32;   // per-workgroup reduction scratch (block_sum: 4, fixed_sum_1024: 32)

// One workgroup = one row block: `wg` is the workgroup's index inside this launch's grid part.
// The epilogue's optional prologue (sum of the previous kernel's dot partials -> alpha / beta) runs AFTER the
// row block's col / val loads have been issued and before the operand gather, which is the first thing that
// may need its result: the partials arrive while the matrix stream is in flight instead of in front of it
// (small, launch-bound systems: ~2 us per kernel).  A prologue that returns false ends the workgroup.
template <int RG, class Epi, bool C16, int CH, bool GRP>
__device__ __forceinline__ void csr_stream_body(const CsrView& a, const double* __restrict__ x, Epi& epi, int wg,
                                                double* prod, double* red, int32_t* window) {
  const int tid = threadIdx.x;
  // XCD-aware map: workgroups with equal (index & 7) share an XCD; XCD i owns the i-th
  // contiguous eighth of the row blocks.  One row block per workgroup: a striding
  // (persistent) loop around this body measured 9 % slower on the A SpMV (extra barrier, and
  // the hardware dispatcher balances the tail better).
  const int lb = (wg & (kXcds - 1)) * a.per_xcd + (wg >> 3);
  const int b = lb < a.nblk ? a.blk0 + lb : -1;
  if (b >= 0) {
    const int r0 = a.rowblk[b];
    const int r1 = a.rowblk[b + 1];
    const int p0 = a.rowptr[r0];
    const int cnt = a.rowptr[r1] - p0;
#if NSS_STREAM_PREFETCH
    // Row bounds and epilogue operands of this lane's first phase-2 rows (kPF of them: short-row matrices
    // give a lane several rows) are requested now, so their HBM latency overlaps the matrix stream instead
    // of following the barrier.
    constexpr int kPF = RG == 1 ? NSS_PREFETCH_ROWS : 1;
    const int rf = r0 + tid / RG;
    int rf_s[kPF], rf_e[kPF];
    typename EpiPre<Epi>::type pre[kPF];
#pragma unroll
    for (int j = 0; j < kPF; ++j) {
      const int rj = rf + j * (kBlock / RG);
      const bool has = rj < r1;
      rf_s[j] = has ? a.rowptr[rj] : 0;
      rf_e[j] = has ? a.rowptr[rj + 1] : 0;
      pre[j] = has ? EpiPre<Epi>::fetch(epi, rj) : typename EpiPre<Epi>::type{};
    }
#endif
    if (cnt <= CH) {
      // ---- phase 1: coalesced stream of (col, val), gather x, stage products ---------
      constexpr int kPer = CH / kBlock;
      int32_t c[kPer];
      double v[kPer];
      uint16_t c16[kPer];
      if (C16) {
        if (tid < kWindows) window[tid] = a.blkbase[b * kWindows + tid];
      }
      uint32_t grp = 0, gsub = 0;                          // group of this lane's entry, position inside it
      if (C16 && GRP) {
        const uint32_t p = uint32_t(p0 + tid);
        grp = a.gb > 1 ? uint32_t(__umul64hi(uint64_t(p), a.gmagic)) : p;   // one division per lane, then incremental
        gsub = p - grp * a.gb;
      }
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        const int i = tid + k * kBlock;
        const bool live = i < cnt;
        if (C16 && GRP) {                                  // one index per group of a.gb entries
          c16[k] = live ? a.col16[grp] : uint16_t(0);      // shared by a.gb neighbouring lanes
          c[k] = int32_t(gsub);
          gsub += a.sstep;
          grp += a.gstep + (gsub >= a.gb ? 1u : 0u);
          gsub -= gsub >= a.gb ? a.gb : 0u;
        }
#if NSS_STREAM_NT
        if (C16 && !GRP) c16[k] = live ? __builtin_nontemporal_load(&a.col16[p0 + i]) : uint16_t(0);
        if (!C16) c[k] = live ? __builtin_nontemporal_load(&a.col[p0 + i]) : 0;
        v[k] = live ? __builtin_nontemporal_load(&a.val[p0 + i]) : 0.0;
#else
        if (C16 && !GRP) c16[k] = live ? a.col16[p0 + i] : uint16_t(0);
        if (!C16) c[k] = live ? a.col[p0 + i] : 0;
        v[k] = live ? a.val[p0 + i] : 0.0;
#endif
      }
      if (!EpiPrologue<Epi>::run(epi, red)) return;        // uniform over the workgroup
      const auto xop = EpiX<Epi>::get(epi, x);
      if (C16) {
        __syncthreads();                                   // window bases in LDS
#pragma unroll
        for (int k = 0; k < kPer; ++k)
          c[k] = window[c16[k] >> kWindowBits] + int32_t(c16[k] & ((1 << kWindowBits) - 1)) + (GRP ? c[k] : 0);
      }
      double xv[kPer];
#pragma unroll
      for (int k = 0; k < kPer; ++k) xv[k] = (tid + k * kBlock < cnt) ? xop(c[k]) : 0.0;
#pragma unroll
      for (int k = 0; k < kPer; ++k)
        if (tid + k * kBlock < cnt) prod[tid + k * kBlock] = v[k] * xv[k];
      __syncthreads();
      // ---- phase 2: per-row reduction from LDS -----------------------------------------
      constexpr int kRowsPerPass = kBlock / RG;
      const int sub = tid % RG;
#if NSS_STREAM_PREFETCH
#pragma unroll
      for (int q = 0; q < kPF; ++q) {                      // the prefetched rows
        const int r = rf + q * kRowsPerPass;
        if (r < r1) {
          const int s = rf_s[q] - p0, e = rf_e[q] - p0;
          double sum = 0.0;
          for (int j = s + sub; j < e; j += RG) sum += prod[j];
          if (RG > 1) {
#pragma unroll
            for (int off = RG / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off, kWave);
          }
          if (sub == 0) EpiPre<Epi>::row(epi, r, sum, pre[q]);
        }
      }
      for (int r = rf + kPF * kRowsPerPass; r < r1; r += kRowsPerPass) {
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
      if (!EpiPrologue<Epi>::run(epi, red)) return;
      const auto xop = EpiX<Epi>::get(epi, x);
      double acc = 0.0;
      for (int i = tid; i < cnt; i += kBlock) acc = fma(a.val[p0 + i], xop(a.col[p0 + i]), acc);
      const double sum = block_sum(acc, red);
      if (tid == 0) EpiPre<Epi>::row(epi, r0, sum, EpiPre<Epi>::fetch(epi, r0));
    }
  }
  epi.finish(b, red);  // one dot partial per row block
}

template <int RG, class Epi, bool C16 = false, int CH = kChunk, bool GRP = false>
__global__ __launch_bounds__(kBlock) void csr_stream_kernel(CsrView a, const double* __restrict__ x, Epi epi) {
  __shared__ double prod[CH];
  __shared__ double red[kRedDoubles];
  __shared__ int32_t window[kWindows];
  if (epi.skip()) return;
  csr_stream_body<RG, Epi, C16, CH, GRP>(a, x, epi, int(blockIdx.x), prod, red, window);
}

// Two matrices with the same launch-plan parameters in ONE launch: workgroups [0, grid_a) stream the
// row blocks of `a` with `ea`, the rest those of `b` with `eb` -- two SpMVs that do not depend on each
// other (the fused BPCG iteration: t2 = A t1 and t3 = B (t1 - s0)) share a kernel boundary.
template <int RG, class EpiA, class EpiB, bool C16, int CH, bool GRP = false>
__global__ __launch_bounds__(kBlock) void csr_stream_dual_kernel(CsrView a, CsrView b, int grid_a,
                                                                  const double* __restrict__ xa,
                                                                  const double* __restrict__ xb, EpiA ea, EpiB eb) {
  __shared__ double prod[CH];
  __shared__ double red[kRedDoubles];
  __shared__ int32_t window[kWindows];
  if (int(blockIdx.x) < grid_a) {
    if (ea.skip()) return;
    csr_stream_body<RG, EpiA, C16, CH, GRP>(a, xa, ea, int(blockIdx.x), prod, red, window);
  } else {
    if (eb.skip()) return;
    csr_stream_body<RG, EpiB, C16, CH, GRP>(b, xb, eb, int(blockIdx.x) - grid_a, prod, red, window);
  }
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
      if (A.col16 && A.gb > 1) hipLaunchKernelGGL((csr_stream_kernel<N, Epi, true, kChunkLong, true>), grid, block, 0, st, v, x, epi);   \
      else if (A.col16) hipLaunchKernelGGL((csr_stream_kernel<N, Epi, true, kChunkLong>), grid, block, 0, st, v, x, epi);   \
      else hipLaunchKernelGGL((csr_stream_kernel<N, Epi, false, kChunkLong>), grid, block, 0, st, v, x, epi);          \
    } else {                                                                                                   \
      if (A.col16 && A.gb > 1) hipLaunchKernelGGL((csr_stream_kernel<N, Epi, true, kChunk, true>), grid, block, 0, st, v, x, epi);       \
      else if (A.col16) hipLaunchKernelGGL((csr_stream_kernel<N, Epi, true>), grid, block, 0, st, v, x, epi);  \
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

// A and B in one launch when their launch plans agree (lanes per row, chunk, index width); returns
// false (nothing launched) otherwise -- the caller then issues two launches.
template <class EpiA, class EpiB>
inline bool launch_csr_stream_dual(const nss_csr_s& A, const double* xa, const EpiA& ea, const nss_csr_s& B,
                                   const double* xb, const EpiB& eb, hipStream_t st) {
  if (A.m == 0 || B.m == 0 || A.nblk == 0 || B.nblk == 0) return false;
  if (A.rg != B.rg || A.chunk != B.chunk || (A.col16 != nullptr) != (B.col16 != nullptr)) return false;
  const bool grp = A.gb > 1 || B.gb > 1;      // the grouped decode also reads a one-per-entry stream (gb == 1)
  const CsrView va = A.view(0, A.nblk), vb = B.view(0, B.nblk);
  const int ga = nss_csr_s::grid(A.nblk), gb = nss_csr_s::grid(B.nblk);
  const dim3 grid(ga + gb), block(kBlock);
#define NSS_LAUNCH_DUAL(N)                                                                                             \
  case N:                                                                                                              \
    if (A.chunk == kChunkLong) {                                                                                       \
      if (A.col16 && grp) hipLaunchKernelGGL((csr_stream_dual_kernel<N, EpiA, EpiB, true, kChunkLong, true>), grid, block, 0, st, va, vb, ga, xa, xb, ea, eb);  \
      else if (A.col16) hipLaunchKernelGGL((csr_stream_dual_kernel<N, EpiA, EpiB, true, kChunkLong>), grid, block, 0, st, va, vb, ga, xa, xb, ea, eb);  \
      else hipLaunchKernelGGL((csr_stream_dual_kernel<N, EpiA, EpiB, false, kChunkLong>), grid, block, 0, st, va, vb, ga, xa, xb, ea, eb);         \
    } else {                                                                                                           \
      if (A.col16 && grp) hipLaunchKernelGGL((csr_stream_dual_kernel<N, EpiA, EpiB, true, kChunk, true>), grid, block, 0, st, va, vb, ga, xa, xb, ea, eb);      \
      else if (A.col16) hipLaunchKernelGGL((csr_stream_dual_kernel<N, EpiA, EpiB, true, kChunk>), grid, block, 0, st, va, vb, ga, xa, xb, ea, eb);      \
      else hipLaunchKernelGGL((csr_stream_dual_kernel<N, EpiA, EpiB, false, kChunk>), grid, block, 0, st, va, vb, ga, xa, xb, ea, eb);             \
    }                                                                                                                  \
    break;
  switch (A.rg) {
    NSS_LAUNCH_DUAL(1) NSS_LAUNCH_DUAL(2) NSS_LAUNCH_DUAL(4) NSS_LAUNCH_DUAL(8) NSS_LAUNCH_DUAL(16) NSS_LAUNCH_DUAL(32)
    NSS_LAUNCH_DUAL(64)
    default: throw Error("csr_stream: bad lanes-per-row in the launch plan");
  }
#undef NSS_LAUNCH_DUAL
  NSS_CHECK_LAUNCH();
  return true;
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
