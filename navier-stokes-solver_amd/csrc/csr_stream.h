// CSR-stream SpMV kernel template for gfx950 (fp64 values, int32 indices).
//
// The Stokes blocks have short rows (7 / 6 / 2 non-zeros for A / B / B^T on the MAC
// grid, ~25-84 for HDG-like operators), so a row-per-wave kernel would idle most of
// its 64 lanes.  Instead a 256-thread workgroup owns a contiguous *row block* whose
// non-zeros fit one LDS chunk:
//   phase 1  every lane streams val[] / col[] with unit stride (fully coalesced HBM
//            reads, 8 independent loads in flight per lane), takes x[col] -- from an LDS copy of
//            the row block's operand segments (staged form, below) or by a gather served by
//            L2 / Infinity Cache -- and stages the products in LDS;
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

#ifndef NSS_CHUNK
#define NSS_CHUNK 2048
#endif
constexpr int kChunk = NSS_CHUNK;       // products staged per workgroup: 16 KiB of LDS
// (Round 1 staged 4096 products for matrices with >= 32 non-zeros per row: with the gathered operand that was
// worth +5 % at 82 non-zeros per row.  With the staged operand the short chunk wins -- 16 KiB of LDS and ~56
// VGPRs keep 8 workgroups per CU: plain SpMV 0.346 -> 0.309 ms = 6.7 TB/s, C23 -7 % at 82 non-zeros per row,
// profiles/r02_ab_chunk.txt -- and there is one chunk size again.)
constexpr int kMaxRowsPerBlock = 2048;  // bound for blocks of empty / very short rows
constexpr int kXcds = 8;
constexpr int kWindows = 16;          // column windows per row block of the 16-bit index stream
constexpr int kWindowBits = 12;      // 4096 columns per window
#ifndef NSS_DIRECT_ROWS
#define NSS_DIRECT_ROWS 1
#endif
constexpr int kDirectWidth = 2;        // entries per row of the fixed-width copy (csr_direct_kernel)
#ifndef NSS_DIRECT_ROWS_PER_LANE
#define NSS_DIRECT_ROWS_PER_LANE 2
#endif
constexpr int kDirectRows = NSS_DIRECT_ROWS_PER_LANE * 256;   // rows per row block of such a matrix
// Staged operand (nss_csr_s::blkseg): per row block at most kSegMax runs of consecutive columns, together at
// most `chunk` columns (the LDS copy shares the product buffer), copied to LDS by LDS-DMA ahead of the matrix
// stream.
constexpr int kSegMax = 13;
constexpr int kSegWords = 32;        // descriptor: r0, r1, p0, cnt, nseg, total, pre[1..12], off[0..12], spare
constexpr int kSegPre = 6, kSegOff = 18;
constexpr int kSegBlock = 31;        // dispatch-ordered copy of the table: the row block this slot stands for

bool pair_staging_enabled();     // nss_csr_pair_mode (tests, measurements)

struct CsrView {
  const int32_t* __restrict__ rowblk;
  const int32_t* __restrict__ rowptr;
  const int32_t* __restrict__ col;
  const uint16_t* __restrict__ col16;   // the 16-bit stream of `mode` (window-relative columns or staged positions), or NULL
  const int32_t* __restrict__ blkbase;  // kWindows window bases per row block (mode 1)
  const int32_t* __restrict__ blkseg;   // kSegWords per row block (mode 2)
  const int32_t* __restrict__ blkdisp;  // the same descriptors in DISPATCH order (8 x per_xcd slots, word kSegBlock = the
                                        // row block, -1 in padding slots), or NULL: slot lb is row block blk0 + lb
  const double* __restrict__ val;
  uint32_t gb;      // entries per column group of the 16-bit stream (1: one index per entry)
  uint32_t gstep, sstep;   // kBlock / gb, kBlock % gb: a lane's next entry is kBlock further down the stream
  uint64_t gmagic;  // ceil(2^64 / gb) (gb > 1): p / gb == __umul64hi(p, gmagic) for p < 2^32
  int32_t blk0;     // first row block of this launch (sub-range launches: interior / boundary)
  int32_t nblk;     // row blocks in this launch
  int32_t per_xcd;  // ceil(nblk / 8)
  int32_t mode;     // how the operand is reached: 0 = 4-byte columns, gather; 1 = 16-bit window-relative columns,
                    // gather; 2 = staged (LDS copy of the row block's operand segments, 16-bit positions);
                    // 3 = pair-staged (the segments of TWO vectors, each in one half of the LDS buffer)
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
  int32_t chunk = nss::kChunk;   // products the kernels' LDS buffer holds (the template parameter CH)
  int32_t blk_products = nss::kChunk;   // products per row block the launch plan aims at (<= chunk; nss_csr_plan_for_pairs)
  std::vector<int32_t> cuts;     // row positions no row block spans (kept for re-planning)
  // Compressed column stream: when the columns of every row block fall into at most 16 aligned
  // windows of 4096 columns (grid operators: a row block touches its own grid plane and the two
  // neighbouring ones -- a few narrow clusters far apart) the kernel streams 2 bytes per entry,
  // 4 bits of window number + 12 bits of offset, decoded through the block's 16 window bases
  // (held in LDS), instead of the 4-byte index: 2 bytes per non-zero less HBM traffic, same
  // products in the same order.  `col` is kept for the set-up kernels and rows longer than a chunk.
  uint16_t* col16 = nullptr;
  int32_t* blkbase = nullptr;
  // Staged operand (preferred where the kernel's operand is one stored vector).  Measured on gfx950
  // (tools/spmv_probe.py, profiles/r02_gather_probe.txt): what keeps the stream kernel at ~5 TB/s is not the
  // bytes of the x gather but the gather itself, a second round of per-lane loads that can only be issued when
  // the column stream has arrived -- with the operand read from LDS instead, the same kernel streams 6.2 TB/s at
  // 7 and at 82 non-zeros per row.  So: when the columns a row block touches form at most kSegMax runs of
  // consecutive columns (gaps of up to 7 unused columns are bridged) with at most kStageCap columns in total --
  // grid operators: one run per stencil leg; block-structured operators: the block columns of a few block rows
  // -- `blkseg` holds the runs and pos16 holds, per entry, the POSITION of its column in the concatenation of
  // the runs.  The kernel copies the runs from x into LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave
  // instruction, no VGPR destination), issued BEFORE the matrix stream (the copy does not depend on it), and
  // reads the operand from LDS: no window decode, no dependent gather.  Same products, same order, same results.
  // A matrix is staged only when every one of its row blocks fits.  The window form is kept beside it for
  // kernels whose operand is an expression of two vectors (they gather).
  uint16_t* pos16 = nullptr;
  int32_t* blkseg = nullptr;
  // Reuse-aware dispatch order (grid operators beyond ~1e7 rows).  XCD i walks its contiguous eighth of the row
  // blocks; the operand runs a block stages besides its own neighbourhood -- the grid planes above and below --
  // are staged again by the blocks P positions further down the walk (P = row blocks per plane).  Up to ~200
  // positions the XCD's 4-MiB L2 still holds them; beyond (5e7 DoF: P ~ 550) every such run comes from HBM a
  // second and a third time (PMC: the dominant launch drew 1.17 x its algorithmic bytes).  `blkdisp` is a copy of
  // the descriptor table in an order that visits the blocks b, b + P, b + 2P, ... of T consecutive planes back to
  // back, then b + 1, b + 1 + P, ...: the re-reads are adjacent in time.  Only the workgroup -> row block map
  // changes: per-row sums and the per-block dot-partial slots, hence all bits, stay.  Built at upload from the run
  // starts (spmv.hip: build_dispatch); full-range launches of the staged forms use it.
  int32_t* blkdisp = nullptr;
  double disp_period = 0.0;      // P (row blocks), 0: natural order
  int32_t disp_planes = 0;       // T
  // Pair-staged operand: kernels whose operand is an expression of TWO stored vectors (t1 - s0 in the rows of B,
  // beta s1 + w1 in the rows of B^T) copy the segments of both -- the first to the lower half of the LDS buffer, the
  // second to the upper half -- and combine what they read back.  Possible when the segments of every row block hold
  // at most chunk / 2 columns (`pair_ok`); nss_csr_plan_for_pairs re-plans a matrix with shorter row blocks to get there.
  bool pair_ok = false;
  // Grouped column stream (block-structured operators: the facet blocks of the HDG-like spaces, ~84
  // non-zeros per row in runs of 12 consecutive columns): when every row length is a multiple of `gb`
  // and every aligned group of `gb` consecutive entries has consecutive columns, col16 / pos16 hold ONE 16-bit
  // index per group -- 8 + 2/gb bytes per non-zero instead of 10 -- and entry p has column (position)
  // decode(index[p / gb]) + p % gb.  The value order stays CSR (rows contiguous), so the kernel, its
  // reduction order and its results are unchanged.  gb == 1: one index per entry.
  int32_t gb = 1;
  // Fixed-width copy of a large matrix with at most kDirectWidth entries per row (B^T of the staggered grid):
  // kDirectWidth (column, value) slots per row, column -1 in unused slots.  Such a matrix is multiplied by
  // csr_direct_kernel: one lane per row, no LDS staging, every load of a lane's rows requested up front -- the
  // stream kernel spends most of a 2-entry-per-row launch in its per-row phase (1024 rows per workgroup, four
  // passes, the epilogue's loads of each pass behind the previous one's stores).
  int32_t* ell_col = nullptr;
  double* ell_val = nullptr;
  // Row blocks planned around the blocks of ONE block-Jacobi handle (nss_csr_plan_for_blocks: every row block holds
  // whole Jacobi blocks and at most kBlockRows rows): a kernel over the rows of this matrix can then apply that
  // preconditioner to its own result in the epilogue -- the row block's results pass through LDS, lanes take one
  // Jacobi block each (the fused BPCG loop: t1 = k J t0 inside C1, no launch of its own and no re-read of t0).
  // jb_order = the Jacobi blocks in row order (callers number them as they like); jb_first[b] = position in jb_order of
  // the first Jacobi block of row block b (nblk + 1 entries); jb_serial = nss_bjac_s::serial of that handle.
  int32_t* jb_first = nullptr;
  int32_t* jb_order = nullptr;
  // Fixed-width copy for EPILOGUE use (nss::fixed_width_copy, built on first request): the kDirectWidth (column,
  // value) slots per row of a matrix with at most that many entries per row, whatever its size -- a kernel over the
  // rows of ANOTHER matrix with the same row count can then add this matrix's row product from registers (fused
  // MINRES: B^T z1 inside the rows of A).  Aliases ell_col / ell_val when those exist.  fw_state: 0 not tried, 1, -1.
  int32_t* fw_col = nullptr;
  double* fw_val = nullptr;
  int fw_state = 0;
  uint64_t jb_serial = 0;
  // `stageable`: the kernel's operand functor is one stored vector that can be copied to LDS (XOp::kStageable);
  // `pairable`: it is an expression of two (XOp::kStageablePair)
  int idx_mode(bool stageable = true, bool pairable = false) const {
    if (blkseg && stageable) return 2;
    if (blkseg && pairable && pair_ok && nss::pair_staging_enabled()) return 3;
    return col16 ? 1 : 0;
  }
  // launch view of the row blocks [b0, b1)
  nss::CsrView view(int b0, int b1, int mode) const {
    const uint64_t magic = gb > 1 ? ~uint64_t(0) / uint64_t(gb) + 1 : 0;    // ceil(2^64 / gb)
    const bool full = b0 == 0 && b1 == nblk;
    return nss::CsrView{rowblk, rowptr, col, mode >= 2 ? pos16 : (mode == 1 ? col16 : nullptr), blkbase, blkseg,
                        (mode >= 2 && full) ? blkdisp : nullptr, val,
                        uint32_t(gb), uint32_t(nss::kBlock / gb), uint32_t(nss::kBlock % gb), magic, b0, b1 - b0,
                        (b1 - b0 + nss::kXcds - 1) / nss::kXcds, mode};
  }
  // one workgroup per row block, padded to a multiple of the XCD count
  static int grid(int count) { return ((count + nss::kXcds - 1) / nss::kXcds) * nss::kXcds; }
};

namespace nss {

// Build col16 / blkbase of a matrix whose device arrays and launch plan are complete (spmv.hip).
void compress_columns(nss_csr_s& A, hipStream_t st);

bool direct_rows_candidate(int32_t m, const int32_t* rowptr);

// the fixed-width copy of A for epilogue use (see nss_csr_s::fw_col); false when a row has more than kDirectWidth entries
bool fixed_width_copy(nss_csr_s& A);

// New launch plan with at most `products` products per row block (set-up only: synchronises the device and rebuilds
// every derived column stream); per-row sums keep their bits.
void replan_row_blocks(nss_csr_s& A, int products);

// Launch plan of a CSR matrix: lanes per row (*rg_out) and the row-block boundaries (spmv.hip).
// `products`: products per row block to aim at (<= kChunk; the lanes-per-row choice does not depend on it, so the
// per-row sums of a re-planned matrix keep their bits)
// `max_rows` > 0: at most that many rows per row block; `row_pos` (one byte per row, host): a row block may only
// start at a row with row_pos == 0 (rows that belong together -- the dofs of one Gauss-Seidel block -- stay in one
// workgroup)
void plan_row_blocks(int32_t m, int64_t nnz, const int32_t* rowptr, int32_t* rg_out, int32_t* chunk_out,
                     std::vector<int32_t>& blk, const int32_t* cuts = nullptr, int ncuts = 0, int products = kChunk,
                     int max_rows = 0, const uint8_t* row_pos = nullptr);

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
//                                                   // that materialises it first.  X::kStageable says
//                                                   // whether the operand is one stored vector (ptr()) with a
//                                                   // scalar map value(raw) on top: only those can use the
//                                                   // staged form of a matrix; xop() is called before the
//                                                   // prologue for ptr() and after it for value() / ().
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
  // an operand that is ONE stored vector (times a scalar at most) can be staged: ptr() is copied to LDS as it is
  // and value() applied to what is read back
  static constexpr bool kStageable = true;
  __device__ const double* ptr() const { return x; }
  __device__ double value(double r) const { return r; }
#if defined(NSS_PROBE_GATHER) && NSS_PROBE_GATHER == 1      // timing probes only (tools/spmv_probe.py): wrong results
  __device__ double operator()(int c) const { return 1.0 + 1e-12 * double(c); }   // no operand gather at all
#elif defined(NSS_PROBE_GATHER) && NSS_PROBE_GATHER == 2
  __device__ double operator()(int c) const { return x[c & 1023]; }               // every gather an L1 hit
#else
  __device__ double operator()(int c) const { return x[c]; }
#endif
};
// An operand that is an expression of TWO stored vectors can be pair-staged: ptr() / ptr2() are copied to the two
// halves of the LDS buffer and pair(a, b) is applied to what is read back (X::kStageablePair, X::ptr2, X::pair).
template <class X, class = void>
struct XPairable : std::false_type {};
template <class X>
struct XPairable<X, std::enable_if_t<X::kStageablePair>> : std::true_type {};

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

constexpr int kBlockRows = 512;   // rows per row block of a matrix planned around Jacobi blocks (their results: 4 KiB of LDS)
constexpr int kRedDoubles = 32;   // per-workgroup reduction scratch (block_sum: 4, fixed_sum_1024: 32)

// Phase 1 of one row block for ONE form of the column stream (IDX: 0 = 4-byte columns, 1 = 16-bit
// window-relative columns, 2 = staged positions; GRP: one 16-bit index per group of a.gb entries): request the
// column and value streams, run the epilogue's prologue while they are in flight, take the operand (gather, or
// the LDS copy the caller has requested) and leave the products in `prod`.  Returns false when the prologue ends
// the workgroup.  The kernel selects the form at run time with uniform branches AROUND this function, and every
// form is straight-line code from the first load to the last product: with the branches inside (one loop per
// form, a common tail) hipcc waited for vmcnt(0) at the joins -- the column stream had to arrive before the
// value stream was even requested (seen in the ISA; plain SpMV of B^T +9 %).
template <class Epi, int KPF>
struct RowPrefetch {
  int s[KPF], e[KPF];
  typename EpiPre<Epi>::type pre[KPF];
};

template <class Epi, int CH, int IDX, bool GRP, int KPF>
__device__ __forceinline__ bool csr_phase1(const CsrView& a, const double* __restrict__ x, Epi& epi, int b, int p0,
                                           int cnt, double* prod, double* red, int32_t* window,
                                           RowPrefetch<Epi, KPF>& pf, int rf, int rstep, int r1) {
  static_assert(IDX != 0 || !GRP, "grouped column stream needs a 16-bit form");
  using XOp = typename EpiX<Epi>::type;
  constexpr int kPer = CH / kBlock;
  const int tid = threadIdx.x;
  int32_t c[kPer];                                       // column (IDX 0), else position inside the group
  double v[kPer];
  uint16_t c16[kPer];
  if (IDX == 1) {
    if (tid < kWindows) window[tid] = a.blkbase[b * kWindows + tid];
  }
  uint32_t grp = 0, gsub = 0;                            // group of this lane's entry, position inside it
  if (GRP) {
    const uint32_t p = uint32_t(p0 + tid);
    grp = a.gb > 1 ? uint32_t(__umul64hi(uint64_t(p), a.gmagic)) : p;   // one division per lane, then incremental
    gsub = p - grp * a.gb;
  }
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int i = tid + k * kBlock;
    const bool live = i < cnt;
    if (GRP) {                                           // one index per group of a.gb entries
      c16[k] = live ? a.col16[grp] : uint16_t(0);        // shared by a.gb neighbouring lanes
      c[k] = int32_t(gsub);
      gsub += a.sstep;
      grp += a.gstep + (gsub >= a.gb ? 1u : 0u);
      gsub -= gsub >= a.gb ? a.gb : 0u;
    }
#if NSS_STREAM_NT
    if (IDX != 0 && !GRP) c16[k] = live ? __builtin_nontemporal_load(&a.col16[p0 + i]) : uint16_t(0);
    if (IDX == 0) c[k] = live ? __builtin_nontemporal_load(&a.col[p0 + i]) : 0;
    v[k] = live ? __builtin_nontemporal_load(&a.val[p0 + i]) : 0.0;
#else
    if (IDX != 0 && !GRP) c16[k] = live ? a.col16[p0 + i] : uint16_t(0);
    if (IDX == 0) c[k] = live ? a.col[p0 + i] : 0;
    v[k] = live ? a.val[p0 + i] : 0.0;
#endif
  }
  // behind the matrix stream and in the same straight-line code (requested in front of it, the register
  // allocator re-used their destination registers for stream addresses and the compiler waited for them in the
  // middle of the stream -- seen in the ISA)
#if NSS_STREAM_PREFETCH
#pragma unroll
  for (int j = 0; j < KPF; ++j) {
    const int rj = rf + j * rstep;
    const bool has = rj < r1;
    pf.s[j] = has ? a.rowptr[rj] : 0;
    pf.e[j] = has ? a.rowptr[rj + 1] : 0;
    pf.pre[j] = has ? EpiPre<Epi>::fetch(epi, rj) : typename EpiPre<Epi>::type{};
  }
#endif
  if (!EpiPrologue<Epi>::run(epi, red)) return false;    // uniform over the workgroup
  const XOp xop = EpiX<Epi>::get(epi, x);
  double xv[kPer];
  if constexpr (IDX == 3) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (see IDX == 2)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int p = int(c16[k]) + (GRP ? c[k] : 0);
      xv[k] = (tid + k * kBlock < cnt) ? xop.pair(prod[p], prod[CH / 2 + p]) : 0.0;
    }
    __syncthreads();
  } else if constexpr (IDX == 2) {
    // LDS-DMA is tracked by vmcnt and gfx950 does not drain it at s_barrier: every wave waits for its own copies
    // before the barrier that publishes them to the other waves.  (hipcc already placed this wait here -- the
    // matrix stream is consumed right behind the barrier -- the statement pins it against code motion.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                     // the LDS copy has landed in every wave's part
#pragma unroll
    for (int k = 0; k < kPer; ++k)
      xv[k] = (tid + k * kBlock < cnt) ? xop.value(prod[int(c16[k]) + (GRP ? c[k] : 0)]) : 0.0;
    __syncthreads();                                     // ... and has been read: the buffer takes the products
  } else {
    if (IDX == 1) {
      __syncthreads();                                   // window bases in LDS
#pragma unroll
      for (int k = 0; k < kPer; ++k)
        c[k] = window[c16[k] >> kWindowBits] + int32_t(c16[k] & ((1 << kWindowBits) - 1)) + (GRP ? c[k] : 0);
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      // columns are non-negative: without the hint the sign extension of a loaded 4-byte column is hoisted to
      // right behind its load, and every column load is followed by a wait (seen in the ISA)
      __builtin_assume(c[k] >= 0);
      xv[k] = (tid + k * kBlock < cnt) ? xop(c[k]) : 0.0;
    }
  }
#pragma unroll
  for (int k = 0; k < kPer; ++k)
    if (tid + k * kBlock < cnt) prod[tid + k * kBlock] = v[k] * xv[k];
  return true;
}

typedef __attribute__((address_space(3))) void* LdsDst;
typedef const __attribute__((address_space(1))) void* GlobalSrc;

// One workgroup = one row block: `wg` is the workgroup's index inside this launch's grid part.
// IDX / GRP: the form of the column stream (csr_phase1).  They are template parameters of the kernels: a run-time
// selection inside one kernel was tried (a fifth of the instantiations) and lost 6-10 % on the short-row
// operators -- with alternative paths in one kernel hipcc's wait-count insertion turns conservative at the joins
// and waits for vmcnt(0) in the middle of the matrix stream (seen in the ISA).  The one run-time branch left is
// the staged kernel's fallback for row blocks that do not fit the LDS copy.
template <int RG, class Epi, int IDX, int CH, bool GRP>
__device__ __forceinline__ void csr_stream_body(const CsrView& a, const double* __restrict__ x, Epi& epi, int wg,
                                                double* prod, double* red, int32_t* window) {
  using XOp = typename EpiX<Epi>::type;
  static_assert(IDX != 2 || XOp::kStageable, "staged form with an operand that cannot be copied to LDS");
  static_assert(IDX != 3 || XPairable<XOp>::value, "pair-staged form with an operand that is not a pair of vectors");
  const int tid = threadIdx.x;
  // XCD-aware map: workgroups with equal (index & 7) share an XCD; XCD i owns the i-th
  // contiguous eighth of the row blocks.  One row block per workgroup: a striding
  // (persistent) loop around this body measured 9 % slower on the A SpMV (extra barrier, and
  // the hardware dispatcher balances the tail better).
  const int lb = (wg & (kXcds - 1)) * a.per_xcd + (wg >> 3);
  int b = lb < a.nblk ? a.blk0 + lb : -1;
  typedef const int32_t __attribute__((address_space(4))) * ConstI32;
  if constexpr (IDX >= 2) {
    // reuse-aware dispatch order: slot lb of the dispatch-ordered table names its row block (uniform scalar load)
    if (a.blkdisp != nullptr) b = ((ConstI32)(a.blkdisp + size_t(lb) * kSegWords))[kSegBlock];
  }
  if (b >= 0) {
    int r0 = 0, r1 = 0, p0 = 0, cnt = 0;
    if constexpr (IDX >= 2) {
      // One descriptor per row block instead of the rowblk -> rowptr chain.  It must arrive through SCALAR loads
      // (the run table then sits in SGPRs); the constant address space makes the loads invariant, and a uniform
      // invariant load is an s_load -- as vector loads (what the compiler emitted in the two-matrix kernel) every
      // use below would wait for vmcnt(0).
      const ConstI32 d = a.blkdisp != nullptr ? (ConstI32)(a.blkdisp + size_t(lb) * kSegWords)
                                              : (ConstI32)(a.blkseg + size_t(b) * kSegWords);
      r0 = d[0];
      r1 = d[1];
      p0 = d[2];
      cnt = d[3];
      const int total = d[5];                              // 0 for a row block that is one over-long row
      // The operand segments of this row block -> LDS (the product buffer, until every lane has taken its
      // values), by LDS-DMA: a wave instruction moves 64 x 16 bytes = 128 consecutive staged positions to
      // wave-uniform base + 16 * lane; each lane supplies the source address of its pair (segments start at
      // even positions and have even lengths: spmv.hip).  Issued BEFORE the matrix stream, on which it does
      // not depend; ptr() only: the prologue has not run yet.
      int32_t seg_pre[kSegMax - 1], seg_off[kSegMax];
#pragma unroll
      for (int s = 0; s < kSegMax - 1; ++s) seg_pre[s] = d[kSegPre + s];
#pragma unroll
      for (int s = 0; s < kSegMax; ++s) seg_off[s] = d[kSegOff + s];
      const int wave = tid / kWave, lane = tid % kWave;
      if constexpr (IDX == 2) {
        const double* __restrict__ src = EpiX<Epi>::get(epi, x).ptr();
#pragma unroll
        for (int j = 0; j < CH / (2 * kBlock); ++j) {
          const int base = j * 2 * kBlock + wave * 2 * kWave;
          const int pos = base + 2 * lane;
          int o = seg_off[0];
#pragma unroll
          for (int s = 1; s < kSegMax; ++s) o = pos >= seg_pre[s - 1] ? seg_off[s] : o;
          if (pos < total) __builtin_amdgcn_global_load_lds((GlobalSrc)(src + pos + o), (LdsDst)(prod + base), 16, 0, 0);
        }
      } else {
        // pair: total <= CH / 2; the first vector's segments -> prod[0, CH/2), the second's -> prod[CH/2, CH)
        const XOp x0 = EpiX<Epi>::get(epi, x);
        const double* __restrict__ src1 = x0.ptr();
        const double* __restrict__ src2 = x0.ptr2();
#pragma unroll
        for (int j = 0; j < CH / (4 * kBlock); ++j) {
          const int base = j * 2 * kBlock + wave * 2 * kWave;
          const int pos = base + 2 * lane;
          int o = seg_off[0];
#pragma unroll
          for (int s = 1; s < kSegMax; ++s) o = pos >= seg_pre[s - 1] ? seg_off[s] : o;
          if (pos < total) {
            __builtin_amdgcn_global_load_lds((GlobalSrc)(src1 + pos + o), (LdsDst)(prod + base), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((GlobalSrc)(src2 + pos + o), (LdsDst)(prod + CH / 2 + base), 16, 0, 0);
          }
        }
      }
    } else {
      r0 = a.rowblk[b];
      r1 = a.rowblk[b + 1];
      p0 = a.rowptr[r0];
      cnt = a.rowptr[r1] - p0;
    }
    // Row bounds and epilogue operands of this lane's first phase-2 rows (kPF of them: short-row matrices
    // give a lane several rows): requested inside phase 1, right behind the matrix stream, so that their HBM
    // latency overlaps it instead of following the barrier.
    constexpr int kPF = RG == 1 ? NSS_PREFETCH_ROWS : 1;
    const int rf = r0 + tid / RG;
    RowPrefetch<Epi, kPF> pf;
    if (cnt <= CH) {
      constexpr int kPer = CH / kBlock;
      (void)kPer;
      // ---- phase 1: coalesced stream of (col, val), operand, stage products ------------
      const bool go = csr_phase1<Epi, CH, IDX, GRP, kPF>(a, x, epi, b, p0, cnt, prod, red, window, pf, rf, kBlock / RG, r1);
      if (!go) return;                                     // the prologue ended the workgroup (uniform)
      __syncthreads();
      // ---- phase 2: per-row reduction from LDS -----------------------------------------
      constexpr int kRowsPerPass = kBlock / RG;
      const int sub = tid % RG;
#if NSS_STREAM_PREFETCH
#pragma unroll
      for (int q = 0; q < kPF; ++q) {                      // the prefetched rows
        const int r = rf + q * kRowsPerPass;
        if (r < r1) {
          const int s = pf.s[q] - p0, e = pf.e[q] - p0;
          double sum = 0.0;
          for (int j = s + sub; j < e; j += RG) sum += prod[j];
          if (RG > 1) {
#pragma unroll
            for (int off = RG / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off, kWave);
          }
          if (sub == 0) EpiPre<Epi>::row(epi, r, sum, pf.pre[q]);
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
      const XOp xop = EpiX<Epi>::get(epi, x);
      double acc = 0.0;
      for (int i = tid; i < cnt; i += kBlock) acc = fma(a.val[p0 + i], xop(a.col[p0 + i]), acc);
      const double sum = block_sum(acc, red);
      if (tid == 0) EpiPre<Epi>::row(epi, r0, sum, EpiPre<Epi>::fetch(epi, r0));
    }
  }
  epi.finish(b, red);  // one dot partial per row block
}

template <int RG, class Epi, int IDX = 0, int CH = kChunk, bool GRP = false>
__global__ __launch_bounds__(kBlock) void csr_stream_kernel(CsrView a, const double* __restrict__ x, Epi epi) {
  __shared__ double prod[CH];
  __shared__ double red[kRedDoubles];
  __shared__ int32_t window[kWindows];
  if (epi.skip()) return;
  csr_stream_body<RG, Epi, IDX, CH, GRP>(a, x, epi, int(blockIdx.x), prod, red, window);
}

// Two matrices with the same launch-plan parameters in ONE launch: two SpMVs that do not depend on each other
// (the fused BPCG iteration: t2 = A t1 and t3 = B (t1 - s0)) share a kernel boundary.  The two halves may use
// different column streams (IDXA / IDXB).  The workgroups of the SECOND matrix come first in the grid
// (`grid_b` of them, then those of the first): in every use it is the smaller one with the less regular
// operand access (B: gathers over three grid planes), and at the end of the grid its few, slow workgroups
// would be the tail of the launch while the chip drains -- in front they overlap the long uniform stream of A.
#ifndef NSS_DUAL_SECOND_FIRST
#define NSS_DUAL_SECOND_FIRST 1
#endif
template <int RG, class EpiA, class EpiB, int IDXA, int IDXB, int CH, bool GRP>
__global__ __launch_bounds__(kBlock) void csr_stream_dual_kernel(CsrView a, CsrView b, int grid_a, int grid_b,
                                                                  const double* __restrict__ xa,
                                                                  const double* __restrict__ xb, EpiA ea, EpiB eb) {
  __shared__ double prod[CH];
  __shared__ double red[kRedDoubles];
  __shared__ int32_t window[kWindows];
#if NSS_DUAL_SECOND_FIRST
  const bool first = int(blockIdx.x) >= grid_b;
  const int wg = first ? int(blockIdx.x) - grid_b : int(blockIdx.x);
#else
  const bool first = int(blockIdx.x) < grid_a;
  const int wg = first ? int(blockIdx.x) : int(blockIdx.x) - grid_a;
#endif
  if (first) {
    if (ea.skip()) return;
    csr_stream_body<RG, EpiA, IDXA, CH, (GRP && IDXA != 0)>(a, xa, ea, wg, prod, red, window);
  } else {
    if (eb.skip()) return;
    csr_stream_body<RG, EpiB, IDXB, CH, (GRP && IDXB != 0)>(b, xb, eb, wg, prod, red, window);
  }
}

// Row-per-lane kernel for the fixed-width copy (nss_csr_s::ell_col): workgroup = one row block of at most
// kDirectRows rows, lane t takes rows r0 + t and r0 + t + 256 (unit stride across the lanes; giving a lane two
// CONSECUTIVE rows instead makes every per-row access of the epilogue a stride-2 access: C1 +35 %).  All loads of both rows -- slots, epilogue
// operands -- are requested before the prologue; the operand is gathered (two entries per row: nothing to
// stage).  The row sum is formed exactly as the stream kernel forms it (0 + p0 + p1, products rounded on their
// own: mul_unfused), so both kernels give identical bits.
template <class Epi>
__global__ __launch_bounds__(kBlock) void csr_direct_kernel(CsrView a, const int32_t* __restrict__ ecol,
                                                            const double* __restrict__ eval,
                                                            const double* __restrict__ x, Epi epi) {
  static_assert(kDirectWidth == 2, "csr_direct_kernel is written for two slots per row");
  using XOp = typename EpiX<Epi>::type;
  __shared__ double red[kRedDoubles];
  if (epi.skip()) return;
  const int tid = threadIdx.x, wg = int(blockIdx.x);
  const int lb = (wg & (kXcds - 1)) * a.per_xcd + (wg >> 3);
  const int b = lb < a.nblk ? a.blk0 + lb : -1;
  if (b >= 0) {
    const int r0 = a.rowblk[b], r1 = a.rowblk[b + 1];
    for (int base = r0; base < r1; base += kDirectRows) {     // (one trip: the plan caps such blocks at kDirectRows rows)
      constexpr int kRows = kDirectRows / kBlock;
      typedef int32_t int2v __attribute__((ext_vector_type(2)));
      int2v c[kRows];
      dbl2v v[kRows];
      typename EpiPre<Epi>::type pre[kRows];
#pragma unroll
      for (int q = 0; q < kRows; ++q) {
        const int r = base + tid + q * kBlock;
        const bool live = r < r1;
        c[q] = live ? __builtin_nontemporal_load(reinterpret_cast<const int2v*>(ecol) + r) : int2v{-1, -1};
        v[q] = live ? __builtin_nontemporal_load(reinterpret_cast<const dbl2v*>(eval) + r) : dbl2v{0.0, 0.0};
        pre[q] = live ? EpiPre<Epi>::fetch(epi, r) : typename EpiPre<Epi>::type{};
      }
      if (base == r0 && !EpiPrologue<Epi>::run(epi, red)) return;   // uniform over the workgroup
      const XOp xop = EpiX<Epi>::get(epi, x);
      double x0[kRows], x1[kRows];
#pragma unroll
      for (int q = 0; q < kRows; ++q) {
        x0[q] = c[q].x >= 0 ? xop(c[q].x) : 0.0;
        x1[q] = c[q].y >= 0 ? xop(c[q].y) : 0.0;
      }
#pragma unroll
      for (int q = 0; q < kRows; ++q) {
        const int r = base + tid + q * kBlock;
        if (r < r1) {
          double sum = 0.0;
          if (c[q].x >= 0) sum += mul_unfused(v[q].x, x0[q]);
          if (c[q].y >= 0) sum += mul_unfused(v[q].y, x1[q]);
          EpiPre<Epi>::row(epi, r, sum, pre[q]);
        }
      }
    }
  }
  epi.finish(b, red);
}

// one instantiation per lanes-per-row value of the launch plan
#define NSS_FOR_PLAN(A, ONE)                                                                      \
  switch ((A).rg) {                                                                                \
    case 1: ONE(1, kChunk) break;                                                                  \
    case 2: ONE(2, kChunk) break;                                                                  \
    case 4: ONE(4, kChunk) break;                                                                  \
    case 8: ONE(8, kChunk) break;                                                                  \
    case 16: ONE(16, kChunk) break;                                                                \
    case 32: ONE(32, kChunk) break;                                                                \
    case 64: ONE(64, kChunk) break;                                                                \
    default: throw Error("csr_stream: bad lanes-per-row in the launch plan");                      \
  }

// the row-per-lane kernel alone (A must hold the fixed-width copy): for epilogue variants that only exist for it
template <class Epi>
inline void launch_csr_direct(const nss_csr_s& A, const double* x, const Epi& epi, hipStream_t st, int b0 = 0,
                              int b1 = -1, size_t dyn_lds = 0) {
  if (b1 < 0) b1 = A.nblk;
  if (A.m == 0 || b1 <= b0) return;
  if (!A.ell_col) throw Error("csr_direct: the matrix has no fixed-width copy");
  hipLaunchKernelGGL((csr_direct_kernel<Epi>), dim3(nss_csr_s::grid(b1 - b0)), dim3(kBlock), dyn_lds, st, A.view(b0, b1, 0),
                     A.ell_col, A.ell_val, x, epi);
  NSS_CHECK_LAUNCH();
}

// rows of the row blocks [b0, b1) (default: all)
template <class Epi>
inline void launch_csr_stream(const nss_csr_s& A, const double* x, const Epi& epi, hipStream_t st, int b0 = 0,
                              int b1 = -1, size_t dyn_lds = 0) {
  if (b1 < 0) b1 = A.nblk;
  if (A.m == 0 || b1 <= b0) return;
  if (A.ell_col) {
    launch_csr_direct(A, x, epi, st, b0, b1, dyn_lds);
    return;
  }
  constexpr bool kCanStage = EpiX<Epi>::type::kStageable;
  constexpr bool kCanPair = XPairable<typename EpiX<Epi>::type>::value;
  const int mode = A.idx_mode(kCanStage, kCanPair);
  const bool grp = mode != 0 && A.gb > 1;
  const CsrView v = A.view(b0, b1, mode);
  const dim3 grid(nss_csr_s::grid(b1 - b0)), block(kBlock);
#define NSS_LAUNCH_ONE(N, CHK)                                                                                  \
  if (mode == 2) {                                                                                               \
    if constexpr (kCanStage) {                                                                                   \
      if (grp) hipLaunchKernelGGL((csr_stream_kernel<N, Epi, 2, CHK, true>), grid, block, dyn_lds, st, v, x, epi);     \
      else hipLaunchKernelGGL((csr_stream_kernel<N, Epi, 2, CHK, false>), grid, block, dyn_lds, st, v, x, epi);        \
    }                                                                                                            \
  } else if (mode == 3) {                                                                                        \
    if constexpr (kCanPair) {                                                                                    \
      if (grp) hipLaunchKernelGGL((csr_stream_kernel<N, Epi, 3, CHK, true>), grid, block, dyn_lds, st, v, x, epi);     \
      else hipLaunchKernelGGL((csr_stream_kernel<N, Epi, 3, CHK, false>), grid, block, dyn_lds, st, v, x, epi);        \
    }                                                                                                            \
  } else if (mode == 1) {                                                                                        \
    if (grp) hipLaunchKernelGGL((csr_stream_kernel<N, Epi, 1, CHK, true>), grid, block, dyn_lds, st, v, x, epi);       \
    else hipLaunchKernelGGL((csr_stream_kernel<N, Epi, 1, CHK, false>), grid, block, dyn_lds, st, v, x, epi);          \
  } else {                                                                                                       \
    hipLaunchKernelGGL((csr_stream_kernel<N, Epi, 0, CHK, false>), grid, block, dyn_lds, st, v, x, epi);               \
  }
  NSS_FOR_PLAN(A, NSS_LAUNCH_ONE)
#undef NSS_LAUNCH_ONE
  NSS_CHECK_LAUNCH();
}

// A and B in one launch when their launch plans agree (lanes per row, chunk) and the pair of column streams is
// one of the instantiated ones; returns false (nothing launched) otherwise -- the caller then issues two launches.
template <class EpiA, class EpiB>
inline bool launch_csr_stream_dual(const nss_csr_s& A, const double* xa, const EpiA& ea, const nss_csr_s& B,
                                   const double* xb, const EpiB& eb, hipStream_t st, size_t dyn_lds = 0) {
#ifdef NSS_NO_DUAL        // measurements: the two halves as launches of their own
  return false;
#endif
  if (A.m == 0 || B.m == 0 || A.nblk == 0 || B.nblk == 0) return false;
  if (A.ell_col || B.ell_col) return false;               // row-per-lane kernel: a launch of its own
  if (A.rg != B.rg || A.chunk != B.chunk) return false;
  constexpr bool kStageA = EpiX<EpiA>::type::kStageable, kStageB = EpiX<EpiB>::type::kStageable;
  constexpr bool kPairB = XPairable<typename EpiX<EpiB>::type>::value;
  const int ma = A.idx_mode(kStageA);
  int mb = B.idx_mode(kStageB, kPairB);
  if (mb == 3 && ma != 2) mb = B.col16 ? 1 : 0;           // the pair-staged half is only instantiated beside a staged one
  if ((ma == 0) != (mb == 0)) return false;               // a 4-byte stream only pairs with a 4-byte stream
  const bool grp = ma != 0 && (A.gb > 1 || B.gb > 1);     // the grouped decode also reads a one-per-entry stream (gb == 1)
  const CsrView va = A.view(0, A.nblk, ma), vb = B.view(0, B.nblk, mb);
  const int ga = nss_csr_s::grid(A.nblk), gb = nss_csr_s::grid(B.nblk);
  const dim3 grid(ga + gb), block(kBlock);
#define NSS_DUAL_GO(N, CHK, IA, IB, G) \
  hipLaunchKernelGGL((csr_stream_dual_kernel<N, EpiA, EpiB, IA, IB, CHK, G>), grid, block, dyn_lds, st, va, vb, ga, gb, xa, xb, ea, eb)
#define NSS_DUAL_PAIR(N, CHK, IA, IB) \
  if (grp) NSS_DUAL_GO(N, CHK, IA, IB, true); else NSS_DUAL_GO(N, CHK, IA, IB, false);
#define NSS_LAUNCH_DUAL_ONE(N, CHK)                                                        \
  if (ma == 0) {                                                                            \
    NSS_DUAL_GO(N, CHK, 0, 0, false);                                                       \
  } else if (ma == 2 && mb == 2) {                                                          \
    if constexpr (kStageA && kStageB) { NSS_DUAL_PAIR(N, CHK, 2, 2) }                       \
  } else if (ma == 2 && mb == 3) {                                                          \
    if constexpr (kStageA && kPairB) { NSS_DUAL_PAIR(N, CHK, 2, 3) }                        \
  } else if (ma == 2) {                                                                     \
    if constexpr (kStageA) { NSS_DUAL_PAIR(N, CHK, 2, 1) }                                  \
  } else if (mb == 2) {                                                                     \
    if constexpr (kStageB) { NSS_DUAL_PAIR(N, CHK, 1, 2) }                                  \
  } else {                                                                                  \
    NSS_DUAL_PAIR(N, CHK, 1, 1)                                                             \
  }
  NSS_FOR_PLAN(A, NSS_LAUNCH_DUAL_ONE)
#undef NSS_LAUNCH_DUAL_ONE
#undef NSS_DUAL_PAIR
#undef NSS_DUAL_GO
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
