// Smoothed-aggregation AMG V(1,1)-cycle on the device (hierarchy built by the host,
// hipla/amg.py).  Every operation of the cycle is a CSR-stream SpMV with a fused epilogue or a
// diagonal scale, all on one stream:
//
//   level l:  x  = w D^-1 b                      diag_scale            (pre-smoothing from x = 0)
//             r  = b - A x                       SpMV, EpiResidual
//             b' = R r                           SpMV                  (restriction)
//             x' = V_{l+1}(b')                   recursion; coarsest: x' = A^-1 b' (dense inverse, SpMV)
//             x += P x'                          SpMV, beta = 1        (prolongation + correction)
//             y  = x + w D^-1 (b - A x)          SpMV, EpiJacobi       (post-smoothing, into the output)
//
// One pre- and one post-smoothing step with the same damping make the cycle symmetric, so it is
// an SPD preconditioner for the Bramble-Pasciak CG.
#include "amg.h"

#include <vector>

namespace nss {

// y = s * d .* x
__global__ __launch_bounds__(kBlock) void amg_diag_kernel(int32_t n, double s, const double* __restrict__ d,
                                                           const double* __restrict__ x, double* __restrict__ y,
                                                           const int32_t* __restrict__ done) {
  if (done != nullptr && *done != 0) return;
  const int stride = gridDim.x * kBlock;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) y[i] = s * (d[i] * x[i]);
}

// out = scale * V_l(b); the cycle is linear, so the scale factor of the outermost call (preA = k V)
// rides in the last kernel instead of a pass over the right-hand side
static void cycle(const nss_amg_s& a, int l, const double* b, double* out, hipStream_t st, const int32_t* done,
                  double scale = 1.0, bool accumulate = false) {
  const AmgLevel& lv = a.levels[l];
  if (l == int(a.levels.size()) - 1) {
    launch_csr_stream(*a.coarse_inverse, b, EpiAxpby{scale, accumulate ? 1.0 : 0.0, out, done}, st);   // x = A^-1 b
    return;
  }
  const int n = lv.n;
  hipLaunchKernelGGL(amg_diag_kernel, dim3(stream_grid(n, kBlock * 4)), dim3(kBlock), 0, st, n, a.omega, lv.dinv, b,
                     lv.x, done);
  NSS_CHECK_LAUNCH();
  launch_csr_stream(*lv.A, lv.x, EpiResidual{b, lv.r, done}, st);
  const AmgLevel& next = a.levels[l + 1];
  launch_csr_stream(*lv.R, lv.r, EpiAxpby{1.0, 0.0, next.b, done}, st);
  cycle(a, l + 1, next.b, next.y, st, done);
  launch_csr_stream(*lv.P, next.y, EpiAxpby{1.0, 1.0, lv.x, done}, st);
  launch_csr_stream(*lv.A, lv.x, EpiJacobi{b, lv.x, lv.dinv, out, a.omega, scale, done, accumulate}, st);
}


// ---- K right-hand sides at once: components that share one hierarchy ---------------------------------------------------
// y(:, k) = A x(:, k) for k < K with every entry of A read ONCE: the products of a row block (<= kMultiChunk entries by
// the plan, nss::replan_row_blocks) go to LDS as K planes, then one lane per row adds them up; the operand is
// interleaved [row][K] (one 8 K-byte gather per entry), what the epilogue reads and writes is addressed with strides
// (VecK) so that the stacked auxiliary vectors at the two ends of the cycle need no transposition pass.
constexpr int kMultiChunk = 1024;
constexpr int kMultiPlane = kMultiChunk + 1;   // the K planes of products in LDS, one bank apart: the lanes of the K pairs of
                                               // a row read the same offset of their planes at the same time

struct VecK {
  double* p;
  int64_t si, sk;                    // element (i, k) at p[i * si + k * sk]: stacked (1, n) or interleaved (K, 1)
  __device__ double& at(int64_t i, int k) const { return p[i * si + k * sk]; }
};

// Phase 2 gives every (row, k) pair a lane (k fastest: the interleaved vectors are then read and written with unit
// stride, the stacked ones as K unit-stride streams); the pair's row bounds and epilogue operands are requested BEFORE
// the matrix stream so that their latency hides behind it.
// L lanes share a pair when rows are long (the Galerkin operators of the coarser levels: ~30 entries per row, where one
// lane per pair would leave most of the workgroup idle while it adds up its products).
typedef const int32_t __attribute__((address_space(4)))* MultiDesc;   // (constant address space: scalar loads)

__global__ __launch_bounds__(kBlock) void multi_desc_kernel(int32_t nblk, const int32_t* __restrict__ rowblk,
                                                             const int32_t* __restrict__ rowptr, int32_t* __restrict__ desc) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= nblk) return;
  const int r0 = rowblk[b], r1 = rowblk[b + 1], p0 = rowptr[r0];
  desc[4 * b] = r0;
  desc[4 * b + 1] = r1;
  desc[4 * b + 2] = p0;
  desc[4 * b + 3] = rowptr[r1] - p0;
}

template <int K, int L, class Epi>
__global__ __launch_bounds__(kBlock) void csr_multi_kernel(CsrView a, const int32_t* __restrict__ desc,
                                                            const double* __restrict__ x, Epi epi) {
  __shared__ double prod[K * kMultiPlane];
  __shared__ double red[kRedDoubles];
  if (epi.skip()) return;
  const int tid = threadIdx.x, wg = int(blockIdx.x);
  const int lb = (wg & (kXcds - 1)) * a.per_xcd + (wg >> 3);          // the XCD-aware map of the other kernels
  if (lb >= a.nblk) return;
  const int b = a.blk0 + lb;
  const MultiDesc d = (MultiDesc)(desc + size_t(b) * 4);
  const int r0 = d[0], r1 = d[1], p0 = d[2], cnt = d[3];
  if (cnt <= kMultiChunk) {
    constexpr int kPF = 2;                                             // pairs per lane whose operands are prefetched
    constexpr int kLanePairs = kBlock / L;                             // pairs per pass of the workgroup
    const int pairs = (r1 - r0) * K;
    const int sub = tid % L, slot = tid / L;
    int ps[kPF], pe[kPF];
    typename Epi::Pre pre[kPF];
#pragma unroll
    for (int q = 0; q < kPF; ++q) {
      const int idx = slot + q * kLanePairs;
      if (idx < pairs) {
        const int i = r0 + idx / K, k = idx % K;
        ps[q] = a.rowptr[i];
        pe[q] = a.rowptr[i + 1];
        if (sub == 0) pre[q] = epi.fetch(i, k);
      }
    }
    // all loads of the lane's entries are requested before the first product (as the single-vector stream kernel does)
    constexpr int kPer = kMultiChunk / kBlock;
    int cc[kPer];
    double vv[kPer], xv[kPer][K];
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      const int i = tid + q * kBlock;
      cc[q] = i < cnt ? __builtin_nontemporal_load(a.col + p0 + i) : -1;
      vv[q] = i < cnt ? __builtin_nontemporal_load(a.val + p0 + i) : 0.0;
    }
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
#pragma unroll
      for (int k = 0; k < K; ++k) xv[q][k] = cc[q] >= 0 ? x[size_t(cc[q]) * K + k] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      const int i = tid + q * kBlock;
      if (i < cnt) {
#pragma unroll
        for (int k = 0; k < K; ++k) prod[k * kMultiPlane + i] = vv[q] * xv[q][k];
      }
    }
    __syncthreads();
    // (whole groups of L lanes take the same branch: the shuffles below stay inside a group)
#pragma unroll
    for (int q = 0; q < kPF; ++q) {
      const int idx = slot + q * kLanePairs;
      if (idx < pairs) {
        const int i = r0 + idx / K, k = idx % K;
        const double* __restrict__ plane = prod + k * kMultiPlane - p0;
        double sum = 0.0;
        for (int j = ps[q] + sub; j < pe[q]; j += L) sum += plane[j];
#pragma unroll
        for (int off = L / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off, kWave);
        if (sub == 0) epi.elem(i, k, sum, pre[q]);
      }
    }
    for (int idx = slot + kPF * kLanePairs; idx < pairs; idx += kLanePairs) {   // more pairs than prefetch slots
      const int i = r0 + idx / K, k = idx % K;
      const double* __restrict__ plane = prod + k * kMultiPlane - p0;
      double sum = 0.0;
      for (int j = a.rowptr[i] + sub; j < a.rowptr[i + 1]; j += L) sum += plane[j];
#pragma unroll
      for (int off = L / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off, kWave);
      if (sub == 0) epi.elem(i, k, sum, epi.fetch(i, k));
    }
  } else {                            // one row longer than the chunk (the dense coarse inverse): the workgroup reduces it
    double acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.0;
    for (int i = tid; i < cnt; i += kBlock) {
      const int c = a.col[p0 + i];
      const double v = a.val[p0 + i];
#pragma unroll
      for (int k = 0; k < K; ++k) acc[k] = fma(v, x[size_t(c) * K + k], acc[k]);
    }
    double sum[K];
#pragma unroll
    for (int k = 0; k < K; ++k) sum[k] = block_sum(acc[k], red);
    if (tid < K) {
      for (int r = r0; r < r1; ++r) {                    // (r1 == r0 + 1: such a row has a block of its own)
        double mine = sum[0];
#pragma unroll
        for (int k = 1; k < K; ++k) mine = tid == k ? sum[k] : mine;
        epi.elem(r, tid, mine, epi.fetch(r, tid));
      }
    }
  }
}

template <int K>
struct MResidual {     // r = b - A x
  VecK b;
  double* __restrict__ r;
  const int32_t* __restrict__ done;
  __device__ bool skip() const { return done != nullptr && *done != 0; }
  struct Pre { double b = 0.0; };
  __device__ Pre fetch(int i, int k) const { return Pre{b.at(i, k)}; }
  __device__ void elem(int i, int k, double ax, const Pre& p) const { r[size_t(i) * K + k] = p.b - ax; }
};

template <int K>
struct MAxpby {        // y = alpha A x + beta y
  double alpha, beta;
  VecK y;
  const int32_t* __restrict__ done;
  __device__ bool skip() const { return done != nullptr && *done != 0; }
  struct Pre { double y = 0.0; };
  __device__ Pre fetch(int i, int k) const { return Pre{beta != 0.0 ? y.at(i, k) : 0.0}; }
  __device__ void elem(int i, int k, double ax, const Pre& p) const {
    y.at(i, k) = beta != 0.0 ? fma(alpha, ax, beta * p.y) : alpha * ax;
  }
};

template <int K>
struct MJacobi {       // y (+)= scale * (x + w dinv (b - A x))
  VecK b;
  const double* __restrict__ x;
  const double* __restrict__ dinv;
  VecK y;
  double w, scale;
  const int32_t* __restrict__ done;
  bool accumulate;
  __device__ bool skip() const { return done != nullptr && *done != 0; }
  struct Pre { double b = 0.0, x = 0.0, dinv = 0.0, y = 0.0; };
  __device__ Pre fetch(int i, int k) const {
    return Pre{b.at(i, k), x[size_t(i) * K + k], dinv[i], accumulate ? y.at(i, k) : 0.0};
  }
  __device__ void elem(int i, int k, double ax, const Pre& p) const {
    const double t = scale * fma(w * p.dinv, p.b - ax, p.x);
    y.at(i, k) = accumulate ? p.y + t : t;
  }
};

// x(i, k) = s * d[i] * b(i, k), x interleaved
template <int K>
__global__ __launch_bounds__(kBlock) void amg_diag_multi_kernel(int32_t n, double s, const double* __restrict__ d, VecK b,
                                                                 double* __restrict__ x, const int32_t* __restrict__ done) {
  if (done != nullptr && *done != 0) return;
  const int stride = gridDim.x * kBlock;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const double sd = d[i];
#pragma unroll
    for (int k = 0; k < K; ++k) x[size_t(i) * K + k] = s * (sd * b.at(i, k));
  }
}

static const int32_t* multi_desc_of(const nss_amg_s& aux, const nss_csr_s& A) {
  for (const auto& e : aux.multi_desc)
    if (e.mat == &A) {
      if (e.rowblk != A.rowblk) throw Error("amg: a level matrix of the joint cycle was re-planned after the handle was created");
      return e.desc;
    }
  throw Error("amg: a matrix of the joint cycle has no row-block descriptors");
}

template <int K, class Epi>
static void launch_multi(const nss_amg_s& aux, const nss_csr_s& A, const double* x, const Epi& epi, hipStream_t st) {
  if (A.m == 0 || A.nblk == 0) return;
  const dim3 grid(nss_csr_s::grid(A.nblk)), block(kBlock);
  const CsrView v = A.view(0, A.nblk, 0);
  const int32_t* desc = multi_desc_of(aux, A);
  const double mean = double(A.nnz) / double(A.m);                     // lanes per (row, k) pair by the mean row length
  if (mean >= 48.0) hipLaunchKernelGGL((csr_multi_kernel<K, 4, Epi>), grid, block, 0, st, v, desc, x, epi);
  else if (mean >= 20.0) hipLaunchKernelGGL((csr_multi_kernel<K, 2, Epi>), grid, block, 0, st, v, desc, x, epi);
  else hipLaunchKernelGGL((csr_multi_kernel<K, 1, Epi>), grid, block, 0, st, v, desc, x, epi);
  NSS_CHECK_LAUNCH();
}

// the K component cycles of the shared hierarchy `h` as one: out = scale * V(b) column by column
template <int K>
static void cycle_multi(const nss_amg_s& aux, const nss_amg_s& h, int l, VecK b, VecK out, hipStream_t st,
                        const int32_t* done, double scale, bool accumulate) {
  const AmgLevel& lv = h.levels[l];
  const nss_amg_s::MultiLevel& w = aux.multi[size_t(l)];
  if (l == int(h.levels.size()) - 1) {
    // (b is interleaved here unless the hierarchy has one level only)
    launch_multi<K>(aux, *h.coarse_inverse, b.p, MAxpby<K>{scale, accumulate ? 1.0 : 0.0, out, done}, st);
    return;
  }
  const int n = lv.n;
  hipLaunchKernelGGL((amg_diag_multi_kernel<K>), dim3(stream_grid(n, kBlock * 2)), dim3(kBlock), 0, st, n, h.omega, lv.dinv,
                     b, w.x, done);
  NSS_CHECK_LAUNCH();
  launch_multi<K>(aux, *lv.A, w.x, MResidual<K>{b, w.r, done}, st);
  const nss_amg_s::MultiLevel& nx = aux.multi[size_t(l) + 1];
  launch_multi<K>(aux, *lv.R, w.r, MAxpby<K>{1.0, 0.0, VecK{nx.b, K, 1}, done}, st);
  cycle_multi<K>(aux, h, l + 1, VecK{nx.b, K, 1}, VecK{nx.y, K, 1}, st, done, 1.0, false);
  launch_multi<K>(aux, *lv.P, nx.y, MAxpby<K>{1.0, 1.0, VecK{w.x, K, 1}, done}, st);
  launch_multi<K>(aux, *lv.A, w.x, MJacobi<K>{b, w.x, lv.dinv, out, h.omega, scale, done, accumulate}, st);
}

static int g_amg_batch = 1;

void amg_apply(const nss_amg_s& a, double bscale, const double* b, double* x, hipStream_t st, const int32_t* done,
               bool accumulate) {
  if (a.T) {                                         // auxiliary-space mode: x (+)= T (sum_c V_c) T^T (bscale b)
    launch_csr_stream(*a.TT, b, EpiAxpby{bscale, 0.0, a.aux_r, done}, st);
    if (!a.multi.empty() && g_amg_batch) {           // one shared hierarchy: all components in one cycle
      const nss_amg_s& h = *a.comps[0];
      const int64_t n0 = h.levels[0].n;
      const VecK rhs{a.aux_r, 1, n0}, sol{a.aux_z, 1, n0};       // stacked [component][node]
      if (a.comps.size() == 2) cycle_multi<2>(a, h, 0, rhs, sol, st, done, 1.0, false);
      else cycle_multi<3>(a, h, 0, rhs, sol, st, done, 1.0, false);
    } else
    for (size_t c = 0; c < a.comps.size(); ++c)
      cycle(*a.comps[c], 0, a.aux_r + a.comp_off[c], a.aux_z + a.comp_off[c], st, done, 1.0);
    launch_csr_stream(*a.T, a.aux_z, EpiAxpby{1.0, accumulate ? 1.0 : 0.0, x, done}, st);
    return;
  }
  cycle(a, 0, b, x, st, done, bscale, accumulate);
}

}  // namespace nss

using namespace nss;

extern "C" {

int nss_amg_create(int32_t nlevels, const nss_amg_level_t* h_levels, nss_csr_t coarse_inverse, double omega,
                   nss_amg_t* out) {
  return guarded([&] {
    NSS_REQUIRE(out && h_levels && coarse_inverse, "amg_create: NULL argument");
    NSS_REQUIRE(nlevels >= 1 && nlevels <= 32, "amg_create: 1 <= nlevels <= 32");
    nss_amg_s* a = new nss_amg_s;
    try {
      a->omega = omega;
      a->coarse_inverse = coarse_inverse;
      for (int l = 0; l < nlevels; ++l) {
        const nss_amg_level_t& in = h_levels[l];
        NSS_REQUIRE(in.A && in.dinv, "amg_create: level without operator / diagonal");
        NSS_REQUIRE(in.A->m == in.A->n, "amg_create: level operator must be square");
        AmgLevel lv;
        lv.n = in.A->m;
        lv.A = in.A;
        lv.dinv = in.dinv;
        if (l + 1 < nlevels) {
          NSS_REQUIRE(in.P && in.R, "amg_create: missing transfer operators");
          const nss_csr_s* nextA = h_levels[l + 1].A;
          NSS_REQUIRE(nextA && in.P->m == lv.n && in.P->n == nextA->m && in.R->m == nextA->m && in.R->n == lv.n,
                      "amg_create: transfer operator shapes do not chain");
          lv.P = in.P;
          lv.R = in.R;
        }
        a->levels.push_back(lv);
      }
      NSS_REQUIRE(coarse_inverse->m == a->levels.back().n && coarse_inverse->n == a->levels.back().n,
                  "amg_create: coarse inverse has the wrong size");
      for (auto& lv : a->levels) {
        const size_t bytes = sizeof(double) * size_t(std::max(1, lv.n));
        NSS_HIP(hipMalloc(&lv.x, bytes));
        NSS_HIP(hipMalloc(&lv.r, bytes));
        NSS_HIP(hipMalloc(&lv.b, bytes));
        NSS_HIP(hipMalloc(&lv.y, bytes));
      }
    } catch (...) {
      nss_amg_destroy(a);
      throw;
    }
    *out = a;
  });
}

int nss_amg_create_auxiliary(nss_csr_t T, nss_csr_t TT, int32_t ncomp, const nss_amg_t* h_comps, nss_amg_t* out) {
  return guarded([&] {
    NSS_REQUIRE(out && T && TT && h_comps, "amg_create_auxiliary: NULL argument");
    NSS_REQUIRE(ncomp >= 1 && ncomp <= 16, "amg_create_auxiliary: 1 <= ncomp <= 16");
    NSS_REQUIRE(TT->m == T->n && TT->n == T->m, "amg_create_auxiliary: TT is not the transpose shape of T");
    nss_amg_s* a = new nss_amg_s;
    try {
      a->T = T;
      a->TT = TT;
      int64_t off = 0;
      a->comp_off.push_back(0);
      for (int c = 0; c < ncomp; ++c) {
        NSS_REQUIRE(h_comps[c] != nullptr && h_comps[c]->T == nullptr && !h_comps[c]->levels.empty(),
                    "amg_create_auxiliary: components must be plain V-cycle handles");
        a->comps.push_back(h_comps[c]);
        off += h_comps[c]->levels[0].n;
        a->comp_off.push_back(int32_t(off));
      }
      NSS_REQUIRE(off == T->n, "amg_create_auxiliary: component sizes do not add up to the columns of T");
      AmgLevel lv;
      lv.n = T->m;
      a->levels.push_back(lv);
      const size_t bytes = sizeof(double) * size_t(std::max<int64_t>(1, off));
      NSS_HIP(hipMalloc(&a->aux_r, bytes));
      NSS_HIP(hipMalloc(&a->aux_z, bytes));
      bool shared = ncomp == 2 || ncomp == 3;
      for (int c = 1; c < ncomp; ++c) shared = shared && h_comps[c] == h_comps[0];
      if (shared && h_comps[0]->levels.size() >= 2) {
        // every level operator is read once for all components: row blocks of at most kMultiChunk products (the plan
        // of the matrices changes -- set-up only; per-row sums of the single-vector kernels keep their bits)
        const nss_amg_s& h = *h_comps[0];
        for (const AmgLevel& lv : h.levels) {
          replan_row_blocks(*const_cast<nss_csr_s*>(lv.A), kMultiChunk);
          if (lv.P) replan_row_blocks(*const_cast<nss_csr_s*>(lv.P), kMultiChunk);
          if (lv.R) replan_row_blocks(*const_cast<nss_csr_s*>(lv.R), kMultiChunk);
        }
        replan_row_blocks(*const_cast<nss_csr_s*>(h.coarse_inverse), kMultiChunk);
        auto describe = [&](const nss_csr_s* M) {
          if (!M || M->nblk == 0) return;
          int32_t* desc = nullptr;
          NSS_HIP(hipMalloc(&desc, sizeof(int32_t) * 4 * size_t(M->nblk)));
          a->multi_desc.push_back({M, M->rowblk, desc});
          hipLaunchKernelGGL(multi_desc_kernel, dim3((M->nblk + kBlock - 1) / kBlock), dim3(kBlock), 0, nullptr, M->nblk,
                             M->rowblk, M->rowptr, desc);
          NSS_CHECK_LAUNCH();
        };
        for (const AmgLevel& lv : h.levels) {
          describe(lv.A);
          describe(lv.P);
          describe(lv.R);
        }
        describe(h.coarse_inverse);
        NSS_HIP(hipDeviceSynchronize());
        a->multi.resize(h.levels.size());
        for (size_t l = 0; l < h.levels.size(); ++l) {
          const size_t lb = sizeof(double) * size_t(ncomp) * size_t(std::max(1, h.levels[l].n));
          NSS_HIP(hipMalloc(&a->multi[l].x, lb));
          NSS_HIP(hipMalloc(&a->multi[l].r, lb));
          NSS_HIP(hipMalloc(&a->multi[l].b, lb));
          NSS_HIP(hipMalloc(&a->multi[l].y, lb));
        }
      }
    } catch (...) {
      nss_amg_destroy(a);
      throw;
    }
    *out = a;
  });
}

int nss_amg_destroy(nss_amg_t a) {
  return guarded([&] {
    if (!a) return;
    (void)hipFree(a->aux_r);
    (void)hipFree(a->aux_z);
    for (auto& e : a->multi_desc) (void)hipFree(e.desc);
    for (auto& m : a->multi) {
      (void)hipFree(m.x);
      (void)hipFree(m.r);
      (void)hipFree(m.b);
      (void)hipFree(m.y);
    }
    for (auto& lv : a->levels) {
      (void)hipFree(lv.x);
      (void)hipFree(lv.r);
      (void)hipFree(lv.b);
      (void)hipFree(lv.y);
    }
    delete a;
  });
}

int nss_amg_batch_components(int32_t on) {
  return guarded([&] { g_amg_batch = on != 0; });
}

int nss_amg_apply_f64(nss_amg_t a, double bscale, const double* b, double* x, nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(a != nullptr, "amg_apply: NULL handle");
    NSS_REQUIRE(b != x, "amg_apply: b must not alias x");
    amg_apply(*a, bscale, b, x, as_stream(stream));
  });
}

}  // extern "C"
