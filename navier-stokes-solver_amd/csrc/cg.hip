// Fused, device-resident preconditioned conjugate gradients (the reference's inner solver
// `CGSolver`, templates/NavierStokesSIMPLE_iterative.py:92,130; BASELINE config 1).
//
//   C1  rows of A    : q = A p, partial <p, q>
//   S1  one workgroup: <p,q> = sum
//   C2  element-wise : alpha = rz / <p,q>;  x += alpha p;  r -= alpha q;  [z = dinv r, partial <r,z>]
//   [P] block Jacobi / Gauss-Seidel / AMG: z = pre r, then partial <r, z>
//   S2  one workgroup: rz_new = sum
//   C3  element-wise : beta = rz_new / rz;  p = z + beta p;  one lane: history, stop test
#include "bpcg2.h"

namespace nss {

enum { G_RZ = 0, G_PQ = 1, G_RZN = 2, G_ERR0 = 3, G_TOL = 4, G_RZ_ODD = 5 };
enum { GC_DONE = 0, GC_ITFINAL = 1, GC_LAST = 2 };
__device__ __forceinline__ int rz_slot(int it) { return (it & 1) ? G_RZ_ODD : G_RZ; }

struct EpiCgQ {
  const int32_t* __restrict__ ctrl;
  const double* __restrict__ p;
  double* __restrict__ q;
  double* __restrict__ partials;
  double acc = 0.0;
  __device__ bool skip() const { return ctrl[GC_DONE] != 0; }
  struct Pre { double p = 0.0; };
  __device__ Pre fetch(int r) const { return Pre{p[r]}; }
  __device__ void row(int r, double ap, const Pre& pre) {
    q[r] = ap;
    acc = fma(pre.p, ap, acc);
  }
  __device__ void finish(int b, double* lds) {
    const double s = block_sum(acc, lds);
    if (threadIdx.x == 0 && b >= 0) partials[b] = s;
  }
};

constexpr int kCgSum = 1024;
__global__ __launch_bounds__(kCgSum) void cg_sum_kernel(const int32_t* __restrict__ ctrl, int n,
                                                         const double* __restrict__ part, double* __restrict__ scal,
                                                         int slot) {
  __shared__ double lds[kCgSum / kWave];
  if (ctrl[GC_DONE] != 0) return;
  double a = 0.0, a2 = 0.0;
  int i = threadIdx.x;
  for (; i + kCgSum < n; i += 2 * kCgSum) {
    a += part[i];
    a2 += part[i + kCgSum];
  }
  for (; i < n; i += kCgSum) a += part[i];
  const double s = wave_sum(a + a2);
  if ((threadIdx.x & (kWave - 1)) == 0) lds[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < kCgSum / kWave; ++w) t += lds[w];
    scal[slot] = t;
  }
}

struct CgArgs {
  int32_t* ctrl;
  double* scal;
  double* hist;
  int32_t n, it;
  double *x, *r, *z, *p;
  const double *q, *dinv;
  double* partials;
};

// x += alpha p, r -= alpha q; fused point-Jacobi / identity preconditioner with partial <r, z>
__global__ __launch_bounds__(kBlock) void cg_update_kernel(CgArgs a, int fused_pre) {
  __shared__ double lds[kBlock / kWave];
  if (a.ctrl[GC_DONE] != 0) return;
  const double alpha = a.scal[rz_slot(a.it)] / a.scal[G_PQ];
  const int stride = gridDim.x * kBlock;
  double acc = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < a.n; i += stride) {
    NSS_ST(a.x[i], fma(alpha, a.p[i], a.x[i]));
    const double rn = fma(-alpha, a.q[i], a.r[i]);
    a.r[i] = rn;
    if (fused_pre) {
      const double zn = a.dinv ? a.dinv[i] * rn : rn;
      a.z[i] = zn;
      acc = fma(rn, zn, acc);
    }
  }
  if (fused_pre) {
    const double s = block_sum(acc, lds);
    if (threadIdx.x == 0) a.partials[blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(kBlock) void cg_dot_kernel(const int32_t* __restrict__ ctrl, int32_t n,
                                                         const double* __restrict__ x, const double* __restrict__ y,
                                                         double* __restrict__ partials) {
  __shared__ double lds[kBlock / kWave];
  if (ctrl[GC_DONE] != 0) return;
  const int stride = gridDim.x * kBlock;
  double acc = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) acc = fma(x[i], y[i], acc);
  const double s = block_sum(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// p = z + beta p; one lane: history, stop test, rz of the next iteration
__global__ __launch_bounds__(kBlock) void cg_direction_kernel(CgArgs a) {
  if (a.ctrl[GC_DONE] != 0) return;
  const double rz = a.scal[rz_slot(a.it)], rzn = a.scal[G_RZN];
  const double beta = rzn / rz;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.scal[rz_slot(a.it + 1)] = rzn;
    const double err = sqrt(fabs(rzn));
    a.hist[a.it] = err;
    a.ctrl[GC_LAST] = a.it;
    if (err < a.scal[G_TOL] * a.scal[G_ERR0]) {
      a.ctrl[GC_ITFINAL] = a.it;
      a.ctrl[GC_DONE] = 1;
    }
  }
  const int stride = gridDim.x * kBlock;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < a.n; i += stride) a.p[i] = fma(beta, a.p[i], a.z[i]);
}

static int cg_grid(const nss_cg_t& s) { return stream_grid(s.n, kBlock * 4); }

static void cg_check(const nss_cg_t* s) {
  NSS_REQUIRE(s != nullptr && s->A != nullptr, "cg: NULL state / matrix");
  NSS_REQUIRE(s->A->m == s->n && s->A->n == s->n, "cg: matrix does not match n");
  NSS_REQUIRE(int(s->pre_diag != nullptr) + int(s->pre_bjac != nullptr) + int(s->pre_amg != nullptr) <= 1,
              "cg: at most one preconditioner");
  NSS_REQUIRE(!s->pre_bjac || s->pre_bjac->n == s->n, "cg: block preconditioner size mismatch");
  NSS_REQUIRE(!s->pre_amg || s->pre_amg->levels[0].n == s->n, "cg: AMG size mismatch");
  NSS_REQUIRE(s->x && s->r && s->z && s->p && s->q && s->scal && s->ctrl && s->hist && s->partials_a && s->partials_b,
              "cg: NULL buffer");
}

static void cg_iteration(const nss_cg_t& s, int it, hipStream_t st) {
  launch_csr_stream(*s.A, s.p, EpiCgQ{s.ctrl, s.p, s.q, s.partials_a}, st);
  hipLaunchKernelGGL(cg_sum_kernel, dim3(1), dim3(kCgSum), 0, st, s.ctrl, s.A->nblk, s.partials_a, s.scal, int(G_PQ));
  NSS_CHECK_LAUNCH();
  const bool fused_pre = !s.pre_bjac && !s.pre_amg;
  CgArgs a{s.ctrl, s.scal, s.hist, s.n, it, s.x, s.r, s.z, s.p, s.q, s.pre_diag, s.partials_b};
  hipLaunchKernelGGL(cg_update_kernel, dim3(cg_grid(s)), dim3(kBlock), 0, st, a, fused_pre ? 1 : 0);
  NSS_CHECK_LAUNCH();
  int nb = cg_grid(s);
  if (!fused_pre) {
    if (s.pre_bjac && !s.pre_bjac->gs_mat) {   // block Jacobi: <r, z> comes out of the apply kernel
      nb = bjac_apply_dot(*s.pre_bjac, 1.0, s.r, s.z, s.partials_b, s.ctrl, st);
    } else {
      if (s.pre_bjac) bjac_apply(*s.pre_bjac, 1.0, s.r, 0.0, s.z, s.ctrl, st);
      else amg_apply(*s.pre_amg, 1.0, s.r, s.z, st);
      hipLaunchKernelGGL(cg_dot_kernel, dim3(nb), dim3(kBlock), 0, st, s.ctrl, s.n, s.r, s.z, s.partials_b);
      NSS_CHECK_LAUNCH();
    }
  }
  hipLaunchKernelGGL(cg_sum_kernel, dim3(1), dim3(kCgSum), 0, st, s.ctrl, nb, s.partials_b, s.scal, int(G_RZN));
  NSS_CHECK_LAUNCH();
  hipLaunchKernelGGL(cg_direction_kernel, dim3(cg_grid(s)), dim3(kBlock), 0, st, a);
  NSS_CHECK_LAUNCH();
}

}  // namespace nss

using namespace nss;

extern "C" {

int nss_cg_workspace(const nss_cg_t* s, int64_t* partials_a, int64_t* partials_b) {
  return guarded([&] {
    NSS_REQUIRE(s && s->A, "cg_workspace: NULL state / matrix");
    if (partials_a) *partials_a = s->A->nblk;
    if (partials_b) *partials_b = std::max<int64_t>(cg_grid(*s), s->pre_bjac ? bjac_dot_grid(*s->pre_bjac) : 0);
  });
}

int nss_cg_iterate(const nss_cg_t* s, int32_t it_begin, int32_t it_end, nss_stream_t stream) {
  return guarded([&] {
    cg_check(s);
    for (int it = it_begin; it < it_end; ++it) cg_iteration(*s, it, as_stream(stream));
  });
}

int nss_cg_poll(const nss_cg_t* s, int32_t* done, int32_t* it_final, int32_t* last_it, nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(s && s->ctrl, "cg_poll: NULL state");
    int32_t h[4] = {0, 0, 0, 0};
    NSS_HIP(hipMemcpyAsync(h, s->ctrl, sizeof h, hipMemcpyDeviceToHost, as_stream(stream)));
    NSS_HIP(hipStreamSynchronize(as_stream(stream)));
    if (done) *done = h[GC_DONE];
    if (it_final) *it_final = h[GC_ITFINAL];
    if (last_it) *last_it = h[GC_LAST];
  });
}

}  // extern "C"
