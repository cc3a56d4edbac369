// Device-side set-up of the smoothed-aggregation hierarchy (the
// reference obtains its 'h1amg' hierarchy from NGSolve, templates/NavierStokesSIMPLE_iterative.py:
// 320-357).  Everything that scales with the matrix runs on the GPU:
//
//   nss_amg_aggregate     distance-2 maximal independent set over the strength graph of A (Luby
//                         rounds with fixed priorities; integer state only, so the aggregates are
//                         identical to the CPU restatement in oracle/krylov_ref.py), roots numbered
//                         by a prefix sum, neighbours joined in four sweeps;
//   nss_amg_prolongator   P = (I - w D^-1 A) T for the piecewise-constant T of the aggregates;
//   nss_csr_spgemm        C = X Y by expand / sort / compress: every product x_ik * y_kj is
//                         written out with the unique key (i, j, k), a radix sort (rocPRIM) orders them, and
//                         one lane adds each group in its original k order -- no atomics, so the
//                         result is bit-reproducible and equal to a row-wise Gustavson product
//                         evaluated without FMA contraction (what scipy computes; like scipy, sums that
//                         are exactly zero are not stored);
//   nss_csr_transpose     stable sort of the entries by column.
//
// Arithmetic that must match the CPU restatement bit for bit is written with __dmul_rn / __dadd_rn /
// __dsub_rn, and the file is compiled with -ffp-contract=off (Makefile): HIP's default
// -ffp-contract=fast fuses across the _rn intrinsics (they are plain operators) and ignores the
// FP_CONTRACT pragma.
#include "amg.h"

#include <cmath>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace nss {

namespace {

// Scratch pool: the expand / sort / compress passes need multi-GB temporaries, and hipMalloc /
// hipFree of such blocks cost far more than the kernels that use them (cfg5: 4.7 of 5.9 s of the
// hierarchy build were allocation calls).  Temporaries therefore come from a grow-only list of
// cached blocks that lives until nss_scratch_trim(); results that are handed to a CSR handle are
// plain hipMalloc allocations of exactly their size.
struct ScratchPool {
  struct Block {
    void* p;
    size_t bytes;
    bool used;
  };
  std::vector<Block> blocks;
  void* get(size_t bytes) {
    Block* best = nullptr;
    for (Block& b : blocks)
      if (!b.used && b.bytes >= bytes && (!best || b.bytes < best->bytes)) best = &b;
    if (best && best->bytes <= 4 * bytes + (size_t(1) << 20)) {
      best->used = true;
      return best->p;
    }
    void* p = nullptr;
    NSS_HIP(hipMalloc(&p, bytes));
    blocks.push_back({p, bytes, true});
    return p;
  }
  void put(void* p) {
    for (Block& b : blocks)
      if (b.p == p) b.used = false;
  }
  void trim() {
    std::vector<Block> keep;
    for (Block& b : blocks) {
      if (b.used) keep.push_back(b);
      else (void)hipFree(b.p);
    }
    blocks.swap(keep);
  }
};
inline ScratchPool& pool() {
  static ScratchPool p;
  return p;
}

template <class T>
struct Dev {  // owning device array; `result` arrays can be handed to a CSR handle with take()
  T* p = nullptr;
  size_t n = 0;
  bool result = false;
  Dev() = default;
  explicit Dev(size_t count, bool is_result = false) : result(is_result) { alloc(count); }
  Dev(const Dev&) = delete;
  Dev& operator=(const Dev&) = delete;
  Dev(Dev&& o) noexcept : p(o.p), n(o.n), result(o.result) {
    o.p = nullptr;
    o.n = 0;
  }
  ~Dev() { release(); }
  void alloc(size_t count) {
    release();
    n = count;
    if (!count) return;
    if (result) NSS_HIP(hipMalloc(&p, sizeof(T) * count));
    else p = static_cast<T*>(pool().get(sizeof(T) * count));
  }
  void release() {
    if (p) {
      if (result) (void)hipFree(p);
      else pool().put(p);
    }
    p = nullptr;
    n = 0;
  }
  T* take() {   // only for result arrays
    T* q = p;
    p = nullptr;
    n = 0;
    return q;
  }
};

inline int grid_for(int64_t work) { return int((work + kBlock - 1) / kBlock); }

inline int bits_for(uint64_t count) {  // bits needed to represent 0 .. count-1
  int b = 1;
  while (b < 64 && (uint64_t(1) << b) < count) ++b;
  return b;
}

// out[i] = sum(in[0..i)) for i in [0, n]; `in` holds n readable entries.
template <class In, class Out>
void exclusive_sum(const In* in, Out* out, size_t n, hipStream_t st) {
  if (n == 0) return;
  size_t bytes = 0;
  NSS_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, Out(0), n, rocprim::plus<Out>(), st));
  Dev<char> tmp(bytes);
  NSS_HIP(rocprim::exclusive_scan(tmp.p, bytes, in, out, Out(0), n, rocprim::plus<Out>(), st));
  NSS_HIP(hipStreamSynchronize(st));
}

template <class T>
T fetch(const T* dev, hipStream_t st) {
  T h;
  NSS_HIP(hipMemcpyAsync(&h, dev, sizeof(T), hipMemcpyDeviceToHost, st));
  NSS_HIP(hipStreamSynchronize(st));
  return h;
}

// ---- strength graph (implicit) ------------------------------------------------------------
struct Graph {
  int32_t m;
  const int32_t* __restrict__ rowptr;
  const int32_t* __restrict__ col;
  const double* __restrict__ val;
  const double* __restrict__ dabs;  // |a_ii|
  double theta;
};

// j = col[p] is a strong neighbour of i: off-diagonal and |a_ij| >= theta sqrt(|a_ii| |a_jj|)
__device__ __forceinline__ bool strong(const Graph& g, int i, int p, int j) {
  if (j == i) return false;
  if (!(g.theta > 0.0)) return true;
  return fabs(g.val[p]) >= __dmul_rn(g.theta, sqrt(__dmul_rn(g.dabs[i], g.dabs[j])));
}

__global__ __launch_bounds__(kBlock) void diag_kernel(int32_t m, const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ col,
                                                       const double* __restrict__ val, double* __restrict__ d,
                                                       double* __restrict__ dabs) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= m) return;
  double v = 0.0;
  for (int p = rowptr[i]; p < rowptr[i + 1]; ++p)
    if (col[p] == i) v = val[p];
  if (d) d[i] = v;
  if (dabs) dabs[i] = fabs(v);
}

// y[i] = max(0, max over strong neighbours j of max(x[j], x2[j]))
__global__ __launch_bounds__(kBlock) void gmax_kernel(Graph g, const int64_t* __restrict__ x,
                                                       const int64_t* __restrict__ x2, int64_t* __restrict__ y) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= g.m) return;
  int64_t best = 0;
  for (int p = g.rowptr[i]; p < g.rowptr[i + 1]; ++p) {
    const int j = g.col[p];
    if (!strong(g, i, p, j)) continue;
    int64_t v = x[j];
    if (x2) v = max(v, x2[j]);
    best = max(best, v);
  }
  y[i] = best;
}

// a candidate wins when no other candidate within distance 2 has a larger priority
__global__ __launch_bounds__(kBlock) void mis_winner_kernel(Graph g, const int64_t* __restrict__ pri,
                                                             const int64_t* __restrict__ one,
                                                             int64_t* __restrict__ win) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= g.m) return;
  int64_t two = 0;
  for (int p = g.rowptr[i]; p < g.rowptr[i + 1]; ++p) {
    const int j = g.col[p];
    if (strong(g, i, p, j)) two = max(two, max(pri[j], one[j]));
  }
  const int64_t mine = pri[i], near = one[i];
  win[i] = (mine > 0 && mine >= max(near, two) && mine > near) ? 1 : 0;
}

// winners become roots; winners and everything within distance 2 of one stop being candidates
__global__ __launch_bounds__(kBlock) void mis_update_kernel(Graph g, int64_t* __restrict__ pri,
                                                             const int64_t* __restrict__ win,
                                                             const int64_t* __restrict__ near1,
                                                             int64_t* __restrict__ root,
                                                             unsigned long long* __restrict__ remaining) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= g.m) return;
  int64_t near2 = 0;
  for (int p = g.rowptr[i]; p < g.rowptr[i + 1]; ++p) {
    const int j = g.col[p];
    if (strong(g, i, p, j)) near2 = max(near2, max(win[j], near1[j]));
  }
  if (win[i]) root[i] = 1;
  const bool hit = win[i] != 0 || near1[i] != 0 || near2 != 0;
  const int64_t mine = hit ? 0 : pri[i];
  pri[i] = mine;
  if (mine > 0) atomicAdd(remaining, 1ull);
}

__global__ __launch_bounds__(kBlock) void number_kernel(int32_t m, const int64_t* __restrict__ flag,
                                                         const int64_t* __restrict__ rank, int64_t base,
                                                         int64_t* __restrict__ agg, int keep_others) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= m) return;
  if (flag[i]) agg[i] = base + rank[i];
  else if (!keep_others) agg[i] = -1;
}

// an unaggregated node joins the aggregate of its first aggregated strong neighbour (CSR order)
__global__ __launch_bounds__(kBlock) void join_kernel(Graph g, const int64_t* __restrict__ in,
                                                       int64_t* __restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= g.m) return;
  int64_t a = in[i];
  if (a < 0) {
    for (int p = g.rowptr[i]; p < g.rowptr[i + 1]; ++p) {
      const int j = g.col[p];
      if (strong(g, i, p, j) && in[j] >= 0) {
        a = in[j];
        break;
      }
    }
  }
  out[i] = a;
}

__global__ __launch_bounds__(kBlock) void left_flag_kernel(int32_t m, const int64_t* __restrict__ agg,
                                                            int64_t* __restrict__ flag) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < m) flag[i] = agg[i] < 0 ? 1 : 0;
}

// ---- expand / sort / compress ---------------------------------------------------------------
// number of products of each X entry = length of the Y row it meets
__global__ __launch_bounds__(kBlock) void count_kernel(int64_t nnzx, const int32_t* __restrict__ xcol,
                                                        const int32_t* __restrict__ yrowptr,
                                                        int64_t* __restrict__ cnt) {
  const int64_t e = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (e < nnzx) {
    const int k = xcol[e];
    cnt[e] = yrowptr[k + 1] - yrowptr[k];
  }
}

__global__ __launch_bounds__(kBlock) void gather_offsets_kernel(int32_t m, const int32_t* __restrict__ rowptr,
                                                                 const int64_t* __restrict__ off,
                                                                 int64_t* __restrict__ roff) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i <= m) roff[i] = off[rowptr[i]];
}

// products of the X entries [e_begin, e_end): key = ((row - ra) << cbits | column) << ebits | position
// of the X entry in its row -- unique, so the order of equal (row, column) pairs after the sort is
// the k order whatever the sort does with ties; value = x * y.  One lane per X entry.
__global__ __launch_bounds__(kBlock) void expand_kernel(int64_t e_begin, int64_t e_end, int32_t ra,
                                                         const int32_t* __restrict__ xrowptr,
                                                         const int32_t* __restrict__ xrows,
                                                         const int32_t* __restrict__ xcol,
                                                         const double* __restrict__ xval,
                                                         const int32_t* __restrict__ yrowptr,
                                                         const int32_t* __restrict__ ycol,
                                                         const double* __restrict__ yval,
                                                         const int64_t* __restrict__ off, int64_t off0, int cbits,
                                                         int ebits, uint64_t* __restrict__ keys,
                                                         double* __restrict__ vals) {
  const int64_t e = e_begin + int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= e_end) return;
  const int i = xrows[e];
  const uint64_t hi = uint64_t(i - ra) << cbits;
  const uint64_t lo = uint64_t(e - xrowptr[i]);
  const int k = xcol[e];
  const double xv = xval[e];
  int64_t o = off[e] - off0;
  for (int q = yrowptr[k]; q < yrowptr[k + 1]; ++q, ++o) {
    keys[o] = ((hi | uint64_t(uint32_t(ycol[q]))) << ebits) | lo;
    vals[o] = __dmul_rn(xv, yval[q]);
  }
}

// head[p] = 1 where a new (row, column) pair starts in the sorted products
__global__ __launch_bounds__(kBlock) void head_flag_kernel(int64_t count, const uint64_t* __restrict__ keys,
                                                            int ebits, uint32_t* __restrict__ head) {
  const int64_t p = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (p < count) head[p] = (p == 0 || (keys[p] >> ebits) != (keys[p - 1] >> ebits)) ? 1u : 0u;
}

__global__ __launch_bounds__(kBlock) void run_start_kernel(int64_t count, const uint32_t* __restrict__ head,
                                                            const uint32_t* __restrict__ pos,
                                                            uint32_t* __restrict__ start) {
  const int64_t p = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (p < count && head[p]) start[pos[p]] = uint32_t(p);
  if (p == count) start[pos[count]] = uint32_t(count);
}

// one lane per (row, column) pair adds its products in their stored (= ascending k) order
__global__ __launch_bounds__(kBlock) void run_sum_kernel(int64_t runs, const uint32_t* __restrict__ start,
                                                          const uint64_t* __restrict__ keys,
                                                          const double* __restrict__ vals, uint64_t cmask,
                                                          int ebits, int32_t* __restrict__ out_col,
                                                          double* __restrict__ out_val) {
  const int64_t u = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (u >= runs) return;
  const uint32_t a = start[u], b = start[u + 1];
  double s = vals[a];
  for (uint32_t p = a + 1; p < b; ++p) s = __dadd_rn(s, vals[p]);
  out_col[u] = int32_t((keys[a] >> ebits) & cmask);
  out_val[u] = s;
}

// row pointers of the rows [ra, rb] of this chunk: pairs before the first product of the row
__global__ __launch_bounds__(kBlock) void chunk_rowptr_kernel(int32_t ra, int32_t rb, const int64_t* __restrict__ roff,
                                                               int64_t off0, const uint32_t* __restrict__ pos,
                                                               int64_t base, int32_t* __restrict__ out_rowptr) {
  const int i = ra + blockIdx.x * kBlock + threadIdx.x;
  if (i <= rb) out_rowptr[i] = int32_t(base + pos[roff[i] - off0]);
}

// tentative prolongator T as CSR: row i holds the single entry (agg[i], 1); *bad counts ids out of range
__global__ __launch_bounds__(kBlock) void tentative_kernel(int32_t m, const int64_t* __restrict__ agg, int64_t nagg,
                                                            int32_t* __restrict__ rowptr, int32_t* __restrict__ col,
                                                            double* __restrict__ val,
                                                            unsigned long long* __restrict__ bad) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i > m) return;
  rowptr[i] = i;
  if (i == m) return;
  const int64_t a = agg[i];
  const bool ok = a >= 0 && a < nagg;
  if (!ok) atomicAdd(bad, 1ull);
  col[i] = ok ? int32_t(a) : 0;
  val[i] = 1.0;
}

// P = T - w D^-1 (A T) on the compressed A T
__global__ __launch_bounds__(kBlock) void smooth_prolongator_kernel(int32_t m, const int32_t* __restrict__ rowptr,
                                                                     const int32_t* __restrict__ col,
                                                                     double* __restrict__ val,
                                                                     const double* __restrict__ diag,
                                                                     const int64_t* __restrict__ agg, double omega) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= m) return;
  const double dinv = 1.0 / diag[i];
  const int32_t own = int32_t(agg[i]);
  for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) {
    const double x = __dmul_rn(omega, __dmul_rn(dinv, val[p]));
    val[p] = col[p] == own ? __dsub_rn(1.0, x) : -x;
  }
}

__global__ __launch_bounds__(kBlock) void nonzero_flag_kernel(int64_t nnz, const double* __restrict__ val,
                                                               int32_t* __restrict__ flag) {
  const int64_t p = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (p < nnz) flag[p] = val[p] != 0.0 ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void compact_kernel(int64_t nnz, const int32_t* __restrict__ flag,
                                                          const int32_t* __restrict__ pos,
                                                          const int32_t* __restrict__ col,
                                                          const double* __restrict__ val,
                                                          int32_t* __restrict__ out_col,
                                                          double* __restrict__ out_val) {
  const int64_t p = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (p < nnz && flag[p]) {
    out_col[pos[p]] = col[p];
    out_val[pos[p]] = val[p];
  }
}

__global__ __launch_bounds__(kBlock) void remap_rowptr_kernel(int32_t m, const int32_t* __restrict__ rowptr,
                                                               const int32_t* __restrict__ pos,
                                                               int32_t* __restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i <= m) out[i] = pos[rowptr[i]];
}

__global__ __launch_bounds__(kBlock) void row_index_kernel(int32_t m, const int32_t* __restrict__ rowptr,
                                                            int32_t* __restrict__ rows) {
  const int i = blockIdx.x * (kBlock / 8) + threadIdx.x / 8;
  if (i >= m) return;
  for (int p = rowptr[i] + (threadIdx.x & 7); p < rowptr[i + 1]; p += 8) rows[p] = i;
}

__global__ __launch_bounds__(kBlock) void histogram_kernel(int64_t nnz, const int32_t* __restrict__ col,
                                                            int32_t* __restrict__ cnt) {
  const int64_t p = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (p < nnz) atomicAdd(&cnt[col[p]], 1);
}

__global__ __launch_bounds__(kBlock) void iota_kernel(int64_t n, uint32_t* __restrict__ v) {
  const int64_t p = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (p < n) v[p] = uint32_t(p);
}

__global__ __launch_bounds__(kBlock) void permute_kernel(int64_t nnz, const uint32_t* __restrict__ perm,
                                                          const int32_t* __restrict__ rows,
                                                          const double* __restrict__ val,
                                                          int32_t* __restrict__ tcol, double* __restrict__ tval) {
  const int64_t p = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (p < nnz) {
    const uint32_t s = perm[p];
    tcol[p] = rows[s];
    tval[p] = val[s];
  }
}



// ---- multicolour ordering (block Gauss-Seidel set-up) -------------------------------------------
// max over the neighbours of row i in one or two graphs (second graph: the transpose, for a
// structurally non-symmetric matrix); the diagonal is skipped, values are ignored
__global__ __launch_bounds__(kBlock) void nbr_max_kernel(int32_t m, const int32_t* __restrict__ rp1,
                                                          const int32_t* __restrict__ c1,
                                                          const int32_t* __restrict__ rp2,
                                                          const int32_t* __restrict__ c2,
                                                          const int64_t* __restrict__ x, int64_t* __restrict__ y) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= m) return;
  int64_t best = 0;
  for (int p = rp1[i]; p < rp1[i + 1]; ++p)
    if (c1[p] != i) best = max(best, x[c1[p]]);
  if (rp2)
    for (int p = rp2[i]; p < rp2[i + 1]; ++p)
      if (c2[p] != i) best = max(best, x[c2[p]]);
  y[i] = best;
}

// candidates of the current colour: priority of the uncoloured nodes, 0 for the others
__global__ __launch_bounds__(kBlock) void color_reset_kernel(int32_t m, const int32_t* __restrict__ colors,
                                                              const int64_t* __restrict__ priority,
                                                              int64_t* __restrict__ pri,
                                                              unsigned long long* __restrict__ count) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= m) return;
  const bool open = colors[i] < 0;
  pri[i] = open ? priority[i] : 0;
  if (open) atomicAdd(count, 1ull);
}

// a candidate whose priority beats all neighbouring candidates takes the colour
__global__ __launch_bounds__(kBlock) void color_win_kernel(int32_t m, const int64_t* __restrict__ pri,
                                                            const int64_t* __restrict__ nmax, int32_t color,
                                                            int32_t* __restrict__ colors, int64_t* __restrict__ win) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= m) return;
  const bool w = pri[i] > 0 && pri[i] > nmax[i];
  win[i] = w ? 1 : 0;
  if (w) colors[i] = color;
}

// winners and their neighbours stop being candidates of this colour
__global__ __launch_bounds__(kBlock) void color_drop_kernel(int32_t m, const int64_t* __restrict__ win,
                                                             const int64_t* __restrict__ hit, int64_t* __restrict__ pri,
                                                             unsigned long long* __restrict__ remaining) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= m) return;
  if (win[i] || hit[i] > 0) pri[i] = 0;
  if (pri[i] > 0) atomicAdd(remaining, 1ull);
}

__global__ __launch_bounds__(kBlock) void ones_kernel(int64_t n, double* __restrict__ v) {
  const int64_t p = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (p < n) v[p] = 1.0;
}

__global__ __launch_bounds__(kBlock) void row_length_kernel(int32_t m, const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ rows,
                                                             int32_t* __restrict__ len) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < m) len[i] = rowptr[rows[i] + 1] - rowptr[rows[i]];
}

__global__ __launch_bounds__(kBlock) void row_copy_kernel(int32_t m, const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ col,
                                                           const double* __restrict__ val,
                                                           const int32_t* __restrict__ rows,
                                                           const int32_t* __restrict__ out_rowptr,
                                                           int32_t* __restrict__ out_col, double* __restrict__ out_val) {
  const int i = blockIdx.x * (kBlock / 8) + threadIdx.x / 8;     // 8 lanes per row
  if (i >= m) return;
  const int src = rowptr[rows[i]], n = rowptr[rows[i] + 1] - src, dst = out_rowptr[i];
  for (int k = threadIdx.x & 7; k < n; k += 8) {
    out_col[dst + k] = col[src + k];
    out_val[dst + k] = val[src + k];
  }
}

}  // namespace

// Build the handle around device arrays (ownership passes to the handle).  col / val must have
// been allocated with 4 spare entries (see nss_csr_create_cuts).
nss_csr_s* adopt_csr(int32_t m, int32_t n, int64_t nnz, Dev<int32_t>& rowptr, Dev<int32_t>& col, Dev<double>& val,
                     const int32_t* cuts = nullptr, int ncuts = 0, int max_rows = 0, const uint8_t* row_pos = nullptr) {
  std::vector<int32_t> h_rowptr(size_t(m) + 1);
  NSS_HIP(hipMemcpy(h_rowptr.data(), rowptr.p, sizeof(int32_t) * (size_t(m) + 1), hipMemcpyDeviceToHost));
  std::vector<int32_t> blk;
  int32_t rg = 1, chunk = kChunk;
  plan_row_blocks(m, nnz, h_rowptr.data(), &rg, &chunk, blk, cuts, ncuts, kChunk, max_rows, row_pos);
  Dev<int32_t> rowblk(blk.size(), true);
  NSS_HIP(hipMemcpy(rowblk.p, blk.data(), sizeof(int32_t) * blk.size(), hipMemcpyHostToDevice));
  nss_csr_s* A = new nss_csr_s;
  A->m = m;
  A->n = n;
  A->nnz = nnz;
  A->rg = rg;
  A->chunk = chunk;
  if (cuts && ncuts > 0) A->cuts.assign(cuts, cuts + ncuts);
  A->nblk = int32_t(blk.size()) - 1;
  A->rowptr = rowptr.take();
  A->col = col.take();
  A->val = val.take();
  A->rowblk = rowblk.take();
  try {
    compress_columns(*A, nullptr);
  } catch (...) {
    nss_csr_destroy(A);
    throw;
  }
  return A;
}

struct RawCsr {
  int32_t m = 0, n = 0;
  int64_t nnz = 0;
  Dev<int32_t> rowptr{0, true}, col{0, true};
  Dev<double> val{0, true};
  nss_csr_s* adopt() { return adopt_csr(m, n, nnz, rowptr, col, val); }
};

// remove the entries that are exactly zero (what a CPU sparse product / sparse subtraction does)
static void drop_zeros(RawCsr& c, hipStream_t st) {
  if (c.nnz == 0) return;
  Dev<int32_t> flag(size_t(c.nnz) + 1), pos(size_t(c.nnz) + 1);
  NSS_HIP(hipMemsetAsync(flag.p, 0, sizeof(int32_t) * (size_t(c.nnz) + 1), st));
  hipLaunchKernelGGL(nonzero_flag_kernel, dim3(grid_for(c.nnz)), dim3(kBlock), 0, st, c.nnz, c.val.p, flag.p);
  NSS_CHECK_LAUNCH();
  exclusive_sum(flag.p, pos.p, size_t(c.nnz) + 1, st);
  const int64_t kept = fetch(pos.p + c.nnz, st);
  if (kept == c.nnz) return;
  Dev<int32_t> rowptr(size_t(c.m) + 1, true), col(size_t(kept) + 4, true);
  Dev<double> val(size_t(kept) + 4, true);
  NSS_HIP(hipMemsetAsync(col.p, 0, sizeof(int32_t) * (size_t(kept) + 4), st));
  NSS_HIP(hipMemsetAsync(val.p, 0, sizeof(double) * (size_t(kept) + 4), st));
  hipLaunchKernelGGL(compact_kernel, dim3(grid_for(c.nnz)), dim3(kBlock), 0, st, c.nnz, flag.p, pos.p, c.col.p,
                     c.val.p, col.p, val.p);
  hipLaunchKernelGGL(remap_rowptr_kernel, dim3(grid_for(int64_t(c.m) + 1)), dim3(kBlock), 0, st, c.m, c.rowptr.p,
                     pos.p, rowptr.p);
  NSS_CHECK_LAUNCH();
  NSS_HIP(hipStreamSynchronize(st));
  std::swap(c.rowptr.p, rowptr.p);
  std::swap(c.col.p, col.p);
  std::swap(c.val.p, val.p);
  c.nnz = kept;
}

static Graph graph_of(const nss_csr_s& A, const double* dabs, double theta) {
  return Graph{A.m, A.rowptr, A.col, A.val, dabs, theta};
}

// C = X Y (expand / sort / compress), rows processed in chunks of at most `cap` products.  Sums that
// come out exactly zero stay in the result as stored entries; drop_zeros() removes them.
static void spgemm(const nss_csr_s& X, const nss_csr_s& Y, hipStream_t st, int64_t cap, RawCsr& result) {
  NSS_REQUIRE(X.n == Y.m, "spgemm: inner dimensions differ");
  const int32_t m = X.m;
  const int cbits = bits_for(uint64_t(std::max(Y.n, 1)));
  Dev<int64_t> cnt(size_t(X.nnz) + 1), off(size_t(X.nnz) + 1), roff(size_t(m) + 1);
  NSS_HIP(hipMemsetAsync(cnt.p, 0, sizeof(int64_t) * (size_t(X.nnz) + 1), st));
  if (X.nnz > 0) {
    hipLaunchKernelGGL(count_kernel, dim3(grid_for(X.nnz)), dim3(kBlock), 0, st, X.nnz, X.col, Y.rowptr, cnt.p);
    NSS_CHECK_LAUNCH();
  }
  exclusive_sum(cnt.p, off.p, size_t(X.nnz) + 1, st);
  hipLaunchKernelGGL(gather_offsets_kernel, dim3(grid_for(int64_t(m) + 1)), dim3(kBlock), 0, st, m, X.rowptr, off.p,
                     roff.p);
  NSS_CHECK_LAUNCH();
  std::vector<int64_t> h_roff(size_t(m) + 1);
  std::vector<int32_t> h_xrow(size_t(m) + 1);
  NSS_HIP(hipMemcpyAsync(h_roff.data(), roff.p, sizeof(int64_t) * (size_t(m) + 1), hipMemcpyDeviceToHost, st));
  NSS_HIP(hipMemcpyAsync(h_xrow.data(), X.rowptr, sizeof(int32_t) * (size_t(m) + 1), hipMemcpyDeviceToHost, st));
  NSS_HIP(hipStreamSynchronize(st));
  cnt.release();
  int32_t max_xrow = 1;
  for (int32_t i = 0; i < m; ++i) max_xrow = std::max(max_xrow, h_xrow[size_t(i) + 1] - h_xrow[i]);
  const int ebits = bits_for(uint64_t(max_xrow));

  Dev<int32_t> xrows{size_t(std::max<int64_t>(X.nnz, 1))};
  if (X.nnz > 0) {
    hipLaunchKernelGGL(row_index_kernel, dim3((m + kBlock / 8 - 1) / (kBlock / 8)), dim3(kBlock), 0, st, m, X.rowptr,
                       xrows.p);
    NSS_CHECK_LAUNCH();
  }
  struct Chunk {
    int32_t ra, rb;
    int64_t runs = 0;
    Dev<int32_t> col;
    Dev<double> val;
  };
  std::vector<Chunk> chunks;
  for (int32_t ra = 0; ra < m;) {
    int32_t rb = ra + 1;
    while (rb < m && h_roff[size_t(rb) + 1] - h_roff[ra] <= cap) ++rb;
    NSS_REQUIRE(h_roff[rb] - h_roff[ra] < (int64_t(1) << 32) - 1, "spgemm: one row block exceeds 2^32 products");
    chunks.emplace_back();
    chunks.back().ra = ra;
    chunks.back().rb = rb;
    ra = rb;
  }
  int64_t max_products = 0;
  int32_t max_rows = 1;
  for (const Chunk& c : chunks) {
    max_products = std::max(max_products, h_roff[c.rb] - h_roff[c.ra]);
    max_rows = std::max(max_rows, c.rb - c.ra);
  }
  const int end_bit = ebits + cbits + bits_for(uint64_t(max_rows));
  NSS_REQUIRE(end_bit <= 64, "spgemm: key does not fit 64 bits");
  const size_t cap_items = size_t(std::max<int64_t>(max_products, 1));
  Dev<uint64_t> keys_a(cap_items), keys_b(cap_items);
  Dev<double> vals_a(cap_items), vals_b(cap_items);
  Dev<uint32_t> head(cap_items + 1), pos(cap_items + 1), start(cap_items + 1);
  size_t sort_bytes = 0;
  NSS_HIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, keys_a.p, keys_b.p, vals_a.p, vals_b.p, cap_items, 0, end_bit,
                                    st));
  Dev<char> sort_tmp(sort_bytes);
  const uint64_t cmask = (uint64_t(1) << cbits) - 1;
  Dev<int32_t> out_rowptr(size_t(m) + 1, true);
  NSS_HIP(hipMemsetAsync(out_rowptr.p, 0, sizeof(int32_t) * (size_t(m) + 1), st));

  int64_t nnz = 0;
  for (Chunk& c : chunks) {
    const int64_t off0 = h_roff[c.ra], count = h_roff[c.rb] - off0;
    const int64_t e0 = h_xrow[c.ra], e1 = h_xrow[c.rb];
    NSS_HIP(hipMemsetAsync(head.p, 0, sizeof(uint32_t) * (size_t(count) + 1), st));
    if (count > 0) {
      hipLaunchKernelGGL(expand_kernel, dim3(grid_for(e1 - e0)), dim3(kBlock), 0, st, e0, e1, c.ra, X.rowptr, xrows.p,
                         X.col, X.val, Y.rowptr, Y.col, Y.val, off.p, off0, cbits, ebits, keys_a.p, vals_a.p);
      NSS_CHECK_LAUNCH();
      size_t bytes = sort_bytes;
      NSS_HIP(rocprim::radix_sort_pairs(sort_tmp.p, bytes, keys_a.p, keys_b.p, vals_a.p, vals_b.p, size_t(count), 0,
                                        end_bit, st));
      hipLaunchKernelGGL(head_flag_kernel, dim3(grid_for(count)), dim3(kBlock), 0, st, count, keys_b.p, ebits, head.p);
      NSS_CHECK_LAUNCH();
    }
    exclusive_sum(head.p, pos.p, size_t(count) + 1, st);
    c.runs = fetch(pos.p + count, st);
    NSS_REQUIRE(nnz + c.runs < (int64_t(1) << 31), "spgemm: result has more than 2^31 non-zeros");
    hipLaunchKernelGGL(chunk_rowptr_kernel, dim3(grid_for(int64_t(c.rb - c.ra) + 1)), dim3(kBlock), 0, st, c.ra, c.rb,
                       roff.p, off0, pos.p, nnz, out_rowptr.p);
    NSS_CHECK_LAUNCH();
    if (c.runs > 0) {
      c.col.alloc(size_t(c.runs));
      c.val.alloc(size_t(c.runs));
      hipLaunchKernelGGL(run_start_kernel, dim3(grid_for(count + 1)), dim3(kBlock), 0, st, count, head.p, pos.p,
                         start.p);
      hipLaunchKernelGGL(run_sum_kernel, dim3(grid_for(c.runs)), dim3(kBlock), 0, st, c.runs, start.p, keys_b.p,
                         vals_b.p, cmask, ebits, c.col.p, c.val.p);
      NSS_CHECK_LAUNCH();
    }
    nnz += c.runs;
    NSS_HIP(hipStreamSynchronize(st));
  }
  // one chunk: its arrays are the result (re-allocated with the 4 spare entries); several: concatenate
  Dev<int32_t> out_col(size_t(nnz) + 4, true);
  Dev<double> out_val(size_t(nnz) + 4, true);
  NSS_HIP(hipMemsetAsync(out_col.p, 0, sizeof(int32_t) * (size_t(nnz) + 4), st));
  NSS_HIP(hipMemsetAsync(out_val.p, 0, sizeof(double) * (size_t(nnz) + 4), st));
  int64_t at = 0;
  for (Chunk& c : chunks) {
    if (c.runs == 0) continue;
    NSS_HIP(hipMemcpyAsync(out_col.p + at, c.col.p, sizeof(int32_t) * c.runs, hipMemcpyDeviceToDevice, st));
    NSS_HIP(hipMemcpyAsync(out_val.p + at, c.val.p, sizeof(double) * c.runs, hipMemcpyDeviceToDevice, st));
    at += c.runs;
  }
  NSS_HIP(hipStreamSynchronize(st));
  result.m = m;
  result.n = Y.n;
  result.nnz = nnz;
  std::swap(result.rowptr.p, out_rowptr.p);
  std::swap(result.col.p, out_col.p);
  std::swap(result.val.p, out_val.p);
}

static nss_csr_s* transpose(const nss_csr_s& A, hipStream_t st) {
  const int32_t m = A.m, n = A.n;
  const int64_t nnz = A.nnz;
  NSS_REQUIRE(nnz < (int64_t(1) << 32), "transpose: too many non-zeros");
  Dev<int32_t> cnt(size_t(n) + 1), trow(size_t(n) + 1, true);
  NSS_HIP(hipMemsetAsync(cnt.p, 0, sizeof(int32_t) * (size_t(n) + 1), st));
  Dev<int32_t> tcol(size_t(nnz) + 4, true);
  Dev<double> tval(size_t(nnz) + 4, true);
  NSS_HIP(hipMemsetAsync(tcol.p, 0, sizeof(int32_t) * (size_t(nnz) + 4), st));
  NSS_HIP(hipMemsetAsync(tval.p, 0, sizeof(double) * (size_t(nnz) + 4), st));
  if (nnz > 0) {
    hipLaunchKernelGGL(histogram_kernel, dim3(grid_for(nnz)), dim3(kBlock), 0, st, nnz, A.col, cnt.p);
    NSS_CHECK_LAUNCH();
  }
  exclusive_sum(cnt.p, trow.p, size_t(n) + 1, st);
  if (nnz > 0) {
    Dev<int32_t> rows{size_t(nnz)};
    Dev<uint32_t> idx_a{size_t(nnz)}, idx_b{size_t(nnz)};
    Dev<uint32_t> key_b{size_t(nnz)};
    hipLaunchKernelGGL(row_index_kernel, dim3((m + kBlock / 8 - 1) / (kBlock / 8)), dim3(kBlock), 0, st, m, A.rowptr,
                       rows.p);
    NSS_CHECK_LAUNCH();
    hipLaunchKernelGGL(iota_kernel, dim3(grid_for(nnz)), dim3(kBlock), 0, st, nnz, idx_a.p);
    NSS_CHECK_LAUNCH();
    const uint32_t* key_a = reinterpret_cast<const uint32_t*>(A.col);
    const int end_bit = bits_for(uint64_t(std::max(n, 1)));
    size_t bytes = 0;
    NSS_HIP(rocprim::radix_sort_pairs(nullptr, bytes, key_a, key_b.p, idx_a.p, idx_b.p, size_t(nnz), 0, end_bit, st));
    Dev<char> tmp(bytes);
    NSS_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, key_a, key_b.p, idx_a.p, idx_b.p, size_t(nnz), 0, end_bit, st));
    hipLaunchKernelGGL(permute_kernel, dim3(grid_for(nnz)), dim3(kBlock), 0, st, nnz, idx_b.p, rows.p, A.val, tcol.p,
                       tval.p);
    NSS_CHECK_LAUNCH();
    NSS_HIP(hipStreamSynchronize(st));
  }
  return adopt_csr(n, m, nnz, trow, tcol, tval);
}

}  // namespace nss

using namespace nss;

extern "C" {

int nss_csr_transpose(nss_csr_t a, nss_csr_t* out) {
  return guarded([&] {
    NSS_REQUIRE(a != nullptr && out != nullptr, "csr_transpose: NULL argument");
    *out = transpose(*a, nullptr);
  });
}

int nss_csr_spgemm(nss_csr_t x, nss_csr_t y, int64_t max_products_per_pass, nss_csr_t* out, nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(x != nullptr && y != nullptr && out != nullptr, "csr_spgemm: NULL argument");
    const int64_t cap = max_products_per_pass > 0 ? max_products_per_pass : (int64_t(1) << 27);
    RawCsr c;
    spgemm(*x, *y, as_stream(stream), cap, c);
    drop_zeros(c, as_stream(stream));
    *out = c.adopt();
  });
}

int nss_csr_ones_like(nss_csr_t a, nss_csr_t* out, nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(a != nullptr && out != nullptr, "csr_ones_like: NULL argument");
    hipStream_t st = as_stream(stream);
    Dev<int32_t> rowptr(size_t(a->m) + 1, true), col(size_t(a->nnz) + 4, true);
    Dev<double> val(size_t(a->nnz) + 4, true);
    NSS_HIP(hipMemcpyAsync(rowptr.p, a->rowptr, sizeof(int32_t) * (size_t(a->m) + 1), hipMemcpyDeviceToDevice, st));
    NSS_HIP(hipMemsetAsync(col.p, 0, sizeof(int32_t) * (size_t(a->nnz) + 4), st));
    NSS_HIP(hipMemsetAsync(val.p, 0, sizeof(double) * (size_t(a->nnz) + 4), st));
    if (a->nnz > 0) {
      NSS_HIP(hipMemcpyAsync(col.p, a->col, sizeof(int32_t) * a->nnz, hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(ones_kernel, dim3(grid_for(a->nnz)), dim3(kBlock), 0, st, a->nnz, val.p);
      NSS_CHECK_LAUNCH();
    }
    NSS_HIP(hipStreamSynchronize(st));
    *out = adopt_csr(a->m, a->n, a->nnz, rowptr, col, val);
  });
}

int nss_csr_select_rows(nss_csr_t a, int32_t nrows, const int32_t* d_rows, int32_t ncuts, const int32_t* h_cuts,
                        nss_csr_t* out, nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(a != nullptr && out != nullptr && nrows >= 0 && (nrows == 0 || d_rows != nullptr),
                "csr_select_rows: NULL argument");
    NSS_REQUIRE(ncuts >= 0 && (ncuts == 0 || h_cuts != nullptr), "csr_select_rows: bad cuts");
    for (int i = 0; i < ncuts; ++i)
      NSS_REQUIRE(h_cuts[i] >= 0 && h_cuts[i] <= nrows && (i == 0 || h_cuts[i] >= h_cuts[i - 1]),
                  "csr_select_rows: cuts must be ascending row positions");
    hipStream_t st = as_stream(stream);
    Dev<int32_t> len(size_t(nrows) + 1), rowptr(size_t(nrows) + 1, true);
    NSS_HIP(hipMemsetAsync(len.p, 0, sizeof(int32_t) * (size_t(nrows) + 1), st));
    if (nrows > 0) {
      hipLaunchKernelGGL(row_length_kernel, dim3(grid_for(nrows)), dim3(kBlock), 0, st, nrows, a->rowptr, d_rows, len.p);
      NSS_CHECK_LAUNCH();
    }
    exclusive_sum(len.p, rowptr.p, size_t(nrows) + 1, st);
    const int64_t nnz = fetch(rowptr.p + nrows, st);
    Dev<int32_t> col(size_t(nnz) + 4, true);
    Dev<double> val(size_t(nnz) + 4, true);
    NSS_HIP(hipMemsetAsync(col.p, 0, sizeof(int32_t) * (size_t(nnz) + 4), st));
    NSS_HIP(hipMemsetAsync(val.p, 0, sizeof(double) * (size_t(nnz) + 4), st));
    if (nrows > 0 && nnz > 0) {
      hipLaunchKernelGGL(row_copy_kernel, dim3((nrows + kBlock / 8 - 1) / (kBlock / 8)), dim3(kBlock), 0, st, nrows,
                         a->rowptr, a->col, a->val, d_rows, rowptr.p, col.p, val.p);
      NSS_CHECK_LAUNCH();
    }
    NSS_HIP(hipStreamSynchronize(st));
    *out = adopt_csr(nrows, a->n, nnz, rowptr, col, val, h_cuts, ncuts);
  });
}

// col[p] = map[col[p]]
__global__ __launch_bounds__(kBlock) void map_columns_kernel(int64_t nnz, const int32_t* __restrict__ map,
                                                              int32_t* __restrict__ col) {
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  for (int64_t p = int64_t(blockIdx.x) * kBlock + threadIdx.x; p < nnz; p += stride) col[p] = map[col[p]];
}

int nss_csr_permute(nss_csr_t a, int32_t nrows, const int32_t* d_rows, const int32_t* d_colmap, int32_t ncols_out,
                    int32_t ncuts, const int32_t* h_cuts, int32_t max_rows, const uint8_t* h_row_pos, nss_csr_t* out,
                    nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(a != nullptr && out != nullptr && nrows >= 0 && (nrows == 0 || d_rows != nullptr) && d_colmap != nullptr,
                "csr_permute: NULL argument");
    NSS_REQUIRE(ncols_out >= 1 && max_rows >= 0, "csr_permute: bad shape");
    NSS_REQUIRE(ncuts >= 0 && (ncuts == 0 || h_cuts != nullptr), "csr_permute: bad cuts");
    for (int i = 0; i < ncuts; ++i)
      NSS_REQUIRE(h_cuts[i] >= 0 && h_cuts[i] <= nrows && (i == 0 || h_cuts[i] >= h_cuts[i - 1]),
                  "csr_permute: cuts must be ascending row positions");
    hipStream_t st = as_stream(stream);
    Dev<int32_t> len(size_t(nrows) + 1), rowptr(size_t(nrows) + 1, true);
    NSS_HIP(hipMemsetAsync(len.p, 0, sizeof(int32_t) * (size_t(nrows) + 1), st));
    if (nrows > 0) {
      hipLaunchKernelGGL(row_length_kernel, dim3(grid_for(nrows)), dim3(kBlock), 0, st, nrows, a->rowptr, d_rows, len.p);
      NSS_CHECK_LAUNCH();
    }
    exclusive_sum(len.p, rowptr.p, size_t(nrows) + 1, st);
    const int64_t nnz = fetch(rowptr.p + nrows, st);
    Dev<int32_t> col(size_t(nnz) + 4, true);
    Dev<double> val(size_t(nnz) + 4, true);
    NSS_HIP(hipMemsetAsync(col.p, 0, sizeof(int32_t) * (size_t(nnz) + 4), st));
    NSS_HIP(hipMemsetAsync(val.p, 0, sizeof(double) * (size_t(nnz) + 4), st));
    if (nrows > 0 && nnz > 0) {
      hipLaunchKernelGGL(row_copy_kernel, dim3((nrows + kBlock / 8 - 1) / (kBlock / 8)), dim3(kBlock), 0, st, nrows,
                         a->rowptr, a->col, a->val, d_rows, rowptr.p, col.p, val.p);
      hipLaunchKernelGGL(map_columns_kernel, dim3(grid_for(nnz)), dim3(kBlock), 0, st, nnz, d_colmap, col.p);
      NSS_CHECK_LAUNCH();
    }
    NSS_HIP(hipStreamSynchronize(st));
    *out = adopt_csr(nrows, ncols_out, nnz, rowptr, col, val, h_cuts, ncuts, max_rows, h_row_pos);
  });
}

// First-fit colouring in the given node order, on the host (a sequential algorithm: set-up only).  On grid-like block
// graphs it finds the parity colouring -- 2 to 4 balanced colours where Luby's maximal independent sets give 5 to 6
// with a tail of tiny ones, each of which costs the Gauss-Seidel sweep a launch.
int nss_graph_color_greedy(nss_csr_t g, nss_csr_t g_transposed, int32_t* h_colors, int32_t* ncolors_out) {
  return guarded([&] {
    NSS_REQUIRE(g != nullptr && h_colors != nullptr && ncolors_out != nullptr, "graph_color_greedy: NULL argument");
    NSS_REQUIRE(g->m == g->n && (!g_transposed || (g_transposed->m == g->m && g_transposed->n == g->n)),
                "graph_color_greedy: the graph must be square");
    const int32_t m = g->m;
    std::vector<int32_t> rp(size_t(m) + 1), cl(size_t(g->nnz)), rp2, cl2;
    NSS_HIP(hipMemcpy(rp.data(), g->rowptr, sizeof(int32_t) * rp.size(), hipMemcpyDeviceToHost));
    if (g->nnz) NSS_HIP(hipMemcpy(cl.data(), g->col, sizeof(int32_t) * cl.size(), hipMemcpyDeviceToHost));
    if (g_transposed) {
      rp2.resize(size_t(m) + 1);
      cl2.resize(size_t(g_transposed->nnz));
      NSS_HIP(hipMemcpy(rp2.data(), g_transposed->rowptr, sizeof(int32_t) * rp2.size(), hipMemcpyDeviceToHost));
      if (g_transposed->nnz) NSS_HIP(hipMemcpy(cl2.data(), g_transposed->col, sizeof(int32_t) * cl2.size(), hipMemcpyDeviceToHost));
    }
    int32_t ncol = 0;
    std::vector<int32_t> mark;                                // mark[c] == i: colour c is taken by a neighbour of node i
    for (int32_t i = 0; i < m; ++i) {
      h_colors[i] = -1;
      auto visit = [&](const std::vector<int32_t>& ptr, const std::vector<int32_t>& col) {
        for (int32_t p = ptr[size_t(i)]; p < ptr[size_t(i) + 1]; ++p) {
          const int32_t j = col[size_t(p)];
          if (j < i && h_colors[j] >= 0) mark[size_t(h_colors[j])] = i;   // nodes j > i are not coloured yet
        }
      };
      visit(rp, cl);
      if (g_transposed) visit(rp2, cl2);
      int32_t c = 0;
      while (c < ncol && mark[size_t(c)] == i) ++c;
      if (c == ncol) {
        mark.push_back(-1);
        ++ncol;
      }
      h_colors[i] = c;
    }
    *ncolors_out = ncol;
  });
}

int nss_graph_color(nss_csr_t g, nss_csr_t g_transposed, const int64_t* d_priority, int32_t* d_colors,
                    int32_t* ncolors_out, nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(g != nullptr && d_priority != nullptr && d_colors != nullptr && ncolors_out != nullptr,
                "graph_color: NULL argument");
    NSS_REQUIRE(g->m == g->n && (!g_transposed || (g_transposed->m == g->m && g_transposed->n == g->n)),
                "graph_color: the graph must be square");
    hipStream_t st = as_stream(stream);
    const int32_t m = g->m;
    *ncolors_out = 0;
    if (m == 0) return;
    const int32_t* rp2 = g_transposed ? g_transposed->rowptr : nullptr;
    const int32_t* c2 = g_transposed ? g_transposed->col : nullptr;
    Dev<int64_t> pri(m), nmax(m), win(m), hit(m);
    Dev<unsigned long long> counter(1);
    const dim3 grid(grid_for(m)), block(kBlock);
    NSS_HIP(hipMemsetAsync(d_colors, 0xff, sizeof(int32_t) * m, st));      // -1: uncoloured
    for (int32_t color = 0; color <= m; ++color) {
      NSS_HIP(hipMemsetAsync(counter.p, 0, sizeof(unsigned long long), st));
      hipLaunchKernelGGL(color_reset_kernel, grid, block, 0, st, m, d_colors, d_priority, pri.p, counter.p);
      NSS_CHECK_LAUNCH();
      if (fetch(counter.p, st) == 0) {
        *ncolors_out = color;
        return;
      }
      for (int round = 0; round <= m; ++round) {         // Luby rounds: a maximal independent set of the rest
        hipLaunchKernelGGL(nbr_max_kernel, grid, block, 0, st, m, g->rowptr, g->col, rp2, c2, pri.p, nmax.p);
        hipLaunchKernelGGL(color_win_kernel, grid, block, 0, st, m, pri.p, nmax.p, color, d_colors, win.p);
        hipLaunchKernelGGL(nbr_max_kernel, grid, block, 0, st, m, g->rowptr, g->col, rp2, c2, win.p, hit.p);
        NSS_HIP(hipMemsetAsync(counter.p, 0, sizeof(unsigned long long), st));
        hipLaunchKernelGGL(color_drop_kernel, grid, block, 0, st, m, win.p, hit.p, pri.p, counter.p);
        NSS_CHECK_LAUNCH();
        if (fetch(counter.p, st) == 0) break;
      }
    }
    throw Error("graph_color: did not terminate");
  });
}

int nss_scratch_trim(void) {
  return guarded([&] {
    NSS_HIP(hipDeviceSynchronize());
    pool().trim();
  });
}

int nss_csr_download(nss_csr_t a, int32_t* h_rowptr, int32_t* h_col, double* h_val) {
  return guarded([&] {
    NSS_REQUIRE(a != nullptr && h_rowptr != nullptr, "csr_download: NULL argument");
    NSS_REQUIRE(a->nnz == 0 || (h_col != nullptr && h_val != nullptr), "csr_download: NULL argument");
    NSS_HIP(hipDeviceSynchronize());
    NSS_HIP(hipMemcpy(h_rowptr, a->rowptr, sizeof(int32_t) * (size_t(a->m) + 1), hipMemcpyDeviceToHost));
    if (a->nnz > 0) {
      NSS_HIP(hipMemcpy(h_col, a->col, sizeof(int32_t) * a->nnz, hipMemcpyDeviceToHost));
      NSS_HIP(hipMemcpy(h_val, a->val, sizeof(double) * a->nnz, hipMemcpyDeviceToHost));
    }
  });
}

int nss_amg_aggregate(nss_csr_t a, double theta, const int64_t* d_priority, int64_t* d_agg, int64_t* nagg_out,
                      nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(a != nullptr && d_priority != nullptr && d_agg != nullptr && nagg_out != nullptr,
                "amg_aggregate: NULL argument");
    NSS_REQUIRE(a->m == a->n, "amg_aggregate: matrix must be square");
    hipStream_t st = as_stream(stream);
    const int32_t m = a->m;
    *nagg_out = 0;
    if (m == 0) return;
    const size_t n1 = size_t(m) + 1;
    Dev<double> dabs(m);
    Dev<int64_t> pri(n1), one(n1), win(n1), near1(n1), root(n1), rank(n1), agg2(n1);
    Dev<unsigned long long> remaining(1);
    const dim3 grid(grid_for(m)), block(kBlock);
    hipLaunchKernelGGL(diag_kernel, grid, block, 0, st, m, a->rowptr, a->col, a->val, (double*)nullptr, dabs.p);
    NSS_CHECK_LAUNCH();
    const Graph g = graph_of(*a, dabs.p, theta);
    NSS_HIP(hipMemcpyAsync(pri.p, d_priority, sizeof(int64_t) * m, hipMemcpyDeviceToDevice, st));
    NSS_HIP(hipMemsetAsync(root.p, 0, sizeof(int64_t) * n1, st));
    // Luby rounds; the candidate count strictly decreases (the largest remaining priority always wins)
    for (int round = 0; round <= m; ++round) {
      hipLaunchKernelGGL(gmax_kernel, grid, block, 0, st, g, pri.p, (const int64_t*)nullptr, one.p);
      hipLaunchKernelGGL(mis_winner_kernel, grid, block, 0, st, g, pri.p, one.p, win.p);
      hipLaunchKernelGGL(gmax_kernel, grid, block, 0, st, g, win.p, (const int64_t*)nullptr, near1.p);
      NSS_HIP(hipMemsetAsync(remaining.p, 0, sizeof(unsigned long long), st));
      hipLaunchKernelGGL(mis_update_kernel, grid, block, 0, st, g, pri.p, win.p, near1.p, root.p, remaining.p);
      NSS_CHECK_LAUNCH();
      if (fetch(remaining.p, st) == 0) break;
    }
    // roots are numbered in index order
    exclusive_sum(root.p, rank.p, n1, st);
    const int64_t nroots = fetch(rank.p + m, st);
    hipLaunchKernelGGL(number_kernel, grid, block, 0, st, m, root.p, rank.p, int64_t(0), d_agg, 0);
    NSS_CHECK_LAUNCH();
    int64_t* cur = d_agg;
    int64_t* nxt = agg2.p;
    for (int sweep = 0; sweep < 4; ++sweep) {
      hipLaunchKernelGGL(join_kernel, grid, block, 0, st, g, cur, nxt);
      NSS_CHECK_LAUNCH();
      std::swap(cur, nxt);
    }
    if (cur != d_agg) NSS_HIP(hipMemcpyAsync(d_agg, cur, sizeof(int64_t) * m, hipMemcpyDeviceToDevice, st));
    // whatever is still unaggregated becomes a singleton
    NSS_HIP(hipMemsetAsync(root.p, 0, sizeof(int64_t) * n1, st));
    hipLaunchKernelGGL(left_flag_kernel, grid, block, 0, st, m, d_agg, root.p);
    NSS_CHECK_LAUNCH();
    exclusive_sum(root.p, rank.p, n1, st);
    const int64_t nleft = fetch(rank.p + m, st);
    hipLaunchKernelGGL(number_kernel, grid, block, 0, st, m, root.p, rank.p, nroots, d_agg, 1);
    NSS_CHECK_LAUNCH();
    NSS_HIP(hipStreamSynchronize(st));
    *nagg_out = nroots + nleft;
  });
}

int nss_amg_prolongator(nss_csr_t a, const int64_t* d_agg, int64_t nagg, double omega, nss_csr_t* out,
                        nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(a != nullptr && d_agg != nullptr && out != nullptr, "amg_prolongator: NULL argument");
    NSS_REQUIRE(a->m == a->n && nagg > 0 && nagg < (int64_t(1) << 31), "amg_prolongator: bad sizes");
    hipStream_t st = as_stream(stream);
    const int32_t m = a->m;
    // tentative prolongator T as a CSR matrix: row i has the single entry (agg[i], 1)
    Dev<int32_t> trow(size_t(m) + 1, true), tcol(size_t(m) + 4, true);
    Dev<double> tval(size_t(m) + 4, true);
    {
      Dev<unsigned long long> bad(1);
      NSS_HIP(hipMemsetAsync(bad.p, 0, sizeof(unsigned long long), st));
      NSS_HIP(hipMemsetAsync(tcol.p, 0, sizeof(int32_t) * (size_t(m) + 4), st));
      NSS_HIP(hipMemsetAsync(tval.p, 0, sizeof(double) * (size_t(m) + 4), st));
      hipLaunchKernelGGL(tentative_kernel, dim3(grid_for(int64_t(m) + 1)), dim3(kBlock), 0, st, m, d_agg, nagg, trow.p,
                         tcol.p, tval.p, bad.p);
      NSS_CHECK_LAUNCH();
      NSS_REQUIRE(fetch(bad.p, st) == 0, "amg_prolongator: aggregate id out of range");
    }
    nss_csr_s* T = adopt_csr(m, int32_t(nagg), m, trow, tcol, tval);
    RawCsr p;
    try {
      spgemm(*a, *T, st, int64_t(1) << 27, p);            // A T: the products are a_ik * 1
      Dev<double> diag(m);
      hipLaunchKernelGGL(diag_kernel, dim3(grid_for(m)), dim3(kBlock), 0, st, m, a->rowptr, a->col, a->val, diag.p,
                         (double*)nullptr);
      hipLaunchKernelGGL(smooth_prolongator_kernel, dim3(grid_for(m)), dim3(kBlock), 0, st, m, p.rowptr.p, p.col.p,
                         p.val.p, diag.p, d_agg, omega);
      NSS_CHECK_LAUNCH();
      NSS_HIP(hipStreamSynchronize(st));
      drop_zeros(p, st);
    } catch (...) {
      nss_csr_destroy(T);
      throw;
    }
    nss_csr_destroy(T);
    *out = p.adopt();
  });
}

}  // extern "C"
