// Device-resident preconditioned Lanczos: the Ritz values of pre * A behind `EigenValues_Preconditioner`
// (reference call sites: bramble_pasciak_cg.py:68-74, solvers/bramblepasciak_new.py:111-122 -- the scale factor
// k = 1 / min(lambda) + 1e-3; the reference times it inside its solver time, run.py:34-38).
//
// Recurrence (hipla/eigen.py::lanczos_ritz, oracle/krylov_ref.py::lanczos_ritz): r_0 = start, z_0 = pre r_0,
// gamma_0 = sqrt<z_0, r_0>; with v_j = r_j / gamma_j, z_j normalised alike:
//   p = A z_j;  delta_j = <p, z_j>;  r_{j+1} = p - delta_j v_j - gamma_j v_{j-1};  z_{j+1} = pre r_{j+1};
//   gamma_{j+1} = sqrt<z_{j+1}, r_{j+1}>;   T = tridiag(gamma, delta, gamma).
// Here the vectors stay UN-normalised (rh_j = gamma_j v_j, zh_j = gamma_j z_j) and the factors go into the scalars --
// two vector passes fewer per step, the same numbers up to rounding:
//   L1  rows of A    : ph = A zh_j, partial <ph, zh_j>
//   L2  one workgroup: dh = sum
//   L3  element-wise : delta_j = dh / gamma_j^2 (lane 0 records it);
//                      rh_{j+1} = ph / gamma_j - (delta_j / gamma_j) rh_j - (gamma_j / gamma_{j-1}) rh_{j-1}
//                      [point Jacobi: zh_{j+1} = dinv rh_{j+1}, partial <zh_{j+1}, rh_{j+1}>]
//   L4  preconditioner (block Jacobi with the dot in the same kernel / Gauss-Seidel / V-cycle / their additive or
//                      multiplicative combination) + partial <zh_{j+1}, rh_{j+1}>
//   L5  one workgroup: g2 = sum; gamma_{j+1} = sqrt|g2| recorded; breakdown test (gamma_{j+1} <= 1e-14 max(|delta_0|,
//                      |delta_j|)) sets the stop flag, after which every kernel returns at once.
// Nothing returns to the host inside a batch of steps: the host reads the new (delta, gamma) pairs and the flag once
// per `check_every` steps and solves the small tridiagonal eigenproblem there (round 2: two host-synchronising dots per
// step -- 532 + 553 of them in the set-up of the 1e7-DoF bench).
#include "bpcg2.h"

namespace nss {

enum { L_DH = 0, L_G2 = 1, L_DELTA = 2, L_SCALE0 = 3, L_GAMMA = 4 /* ring of 3 */ };
enum { LC_STOP = 0, LC_JSTOP = 1, LC_LAST = 2 };
__host__ __device__ __forceinline__ int gamma_slot(int j) { return L_GAMMA + ((j % 3) + 3) % 3; }

struct EpiLanczosP {
  const int32_t* __restrict__ ctrl;
  const double* __restrict__ z;
  double* __restrict__ p;
  double* __restrict__ partials;
  double acc = 0.0;
  __device__ bool skip() const { return ctrl[LC_STOP] != 0; }
  struct Pre { double z = 0.0; };
  __device__ Pre fetch(int r) const { return Pre{z[r]}; }
  __device__ void row(int r, double az, const Pre& pre) {
    p[r] = az;
    acc = fma(az, pre.z, acc);
  }
  __device__ void finish(int b, double* lds) {
    const double s = block_sum(acc, lds);
    if (threadIdx.x == 0 && b >= 0) partials[b] = s;
  }
};

constexpr int kLzSum = 1024;
// scal[slot] = sum of the partials; MODE 1: the books of step j (gamma_{j+1}, history entry, breakdown test);
// MODE 2: the start (gamma_0 = sqrt|sum|; 0 stops everything: the operator annihilates the start vector)
template <int MODE>
__global__ __launch_bounds__(kLzSum) void lanczos_sum_kernel(int32_t* __restrict__ ctrl, int n, const double* __restrict__ part,
                                                              double* __restrict__ scal, int slot, int j,
                                                              double* __restrict__ hist) {
  __shared__ double lds[kLzSum / kWave];
  if (ctrl[LC_STOP] != 0) return;
  double a = 0.0, a2 = 0.0;
  int i = threadIdx.x;
  for (; i + kLzSum < n; i += 2 * kLzSum) {
    a += part[i];
    a2 += part[i + kLzSum];
  }
  for (; i < n; i += kLzSum) a += part[i];
  const double s = wave_sum(a + a2);
  if ((threadIdx.x & (kWave - 1)) == 0) lds[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < kLzSum / kWave; ++w) t += lds[w];
    scal[slot] = t;
    if (MODE == 2) {
      const double gamma0 = sqrt(fabs(t));
      scal[gamma_slot(0)] = gamma0;
      if (gamma0 == 0.0) {
        ctrl[LC_JSTOP] = -1;
        ctrl[LC_STOP] = 1;
      }
    }
    if (MODE == 1) {
      const double gamma_new = sqrt(fabs(t));
      const double delta = scal[L_DELTA];
      scal[gamma_slot(j + 1)] = gamma_new;
      hist[2 * j + 1] = gamma_new;
      ctrl[LC_LAST] = j;
      if (gamma_new <= 1e-14 * fmax(scal[L_SCALE0], fabs(delta))) {
        ctrl[LC_JSTOP] = j;
        ctrl[LC_STOP] = 1;
      }
    }
  }
}

struct LzArgs {
  int32_t* ctrl;
  double* scal;
  double* hist;
  int32_t n, j;
  const double *p, *v, *v_old, *dinv;
  double *v_new, *z_new;
  double pre_scale;
  double* partials;
};

// rh_{j+1} = ph / gamma_j - (delta_j / gamma_j) rh_j - (gamma_j / gamma_{j-1}) rh_{j-1}  [+ fused point Jacobi and dot]
__global__ __launch_bounds__(kBlock) void lanczos_combine_kernel(LzArgs a, int fused_pre) {
  __shared__ double lds[kBlock / kWave];
  if (a.ctrl[LC_STOP] != 0) return;
  const double gamma = a.scal[gamma_slot(a.j)];
  const double delta = a.scal[L_DH] / (gamma * gamma);
  const double ca = 1.0 / gamma, cb = -delta / gamma;
  const double cc = a.j > 0 ? -gamma / a.scal[gamma_slot(a.j - 1)] : 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.scal[L_DELTA] = delta;
    if (a.j == 0) a.scal[L_SCALE0] = fabs(delta);
    a.hist[2 * a.j] = delta;
  }
  const int stride = gridDim.x * kBlock;
  double acc = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < a.n; i += stride) {
    double r = fma(cb, a.v[i], ca * a.p[i]);
    if (a.j > 0) r = fma(cc, a.v_old[i], r);
    a.v_new[i] = r;
    if (fused_pre) {
      const double zn = a.pre_scale * (a.dinv[i] * r);
      a.z_new[i] = zn;
      acc = fma(zn, r, acc);
    }
  }
  if (fused_pre) {
    const double s = block_sum(acc, lds);
    if (threadIdx.x == 0) a.partials[blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(kBlock) void lanczos_dot_kernel(const int32_t* __restrict__ ctrl, int32_t n,
                                                              const double* __restrict__ x, const double* __restrict__ y,
                                                              double* __restrict__ partials) {
  __shared__ double lds[kBlock / kWave];
  if (ctrl[LC_STOP] != 0) return;
  const int stride = gridDim.x * kBlock;
  double acc = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) acc = fma(x[i], y[i], acc);
  const double s = block_sum(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// r = x - A y (multiplicative MypreA: the residual between the two sweeps, :379)
struct EpiLzResidual {
  const int32_t* __restrict__ ctrl;
  const double* __restrict__ x;
  double* __restrict__ r;
  __device__ bool skip() const { return ctrl[LC_STOP] != 0; }
  struct Pre { double x = 0.0; };
  __device__ Pre fetch(int i) const { return Pre{x[i]}; }
  __device__ void row(int i, double ay, const Pre& pre) const { r[i] = pre.x - ay; }
  __device__ void finish(int, double*) const {}
};

__global__ __launch_bounds__(kBlock) void lanczos_scale_kernel(const int32_t* __restrict__ ctrl, int32_t n, double a,
                                                                double* __restrict__ x) {
  if (ctrl[LC_STOP] != 0) return;
  const int stride = gridDim.x * kBlock;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) x[i] *= a;
}

static int lz_grid(const nss_lanczos_t& s) { return stream_grid(s.n, kBlock * 4); }

static void lz_check(const nss_lanczos_t* s) {
  NSS_REQUIRE(s != nullptr && s->A != nullptr, "lanczos: NULL state / matrix");
  NSS_REQUIRE(s->A->m == s->n && s->A->n == s->n, "lanczos: matrix does not match n");
  NSS_REQUIRE(!(s->pre_diag && s->pre_bjac), "lanczos: pre_diag and pre_bjac are exclusive");
  NSS_REQUIRE(s->pre_diag || s->pre_bjac || s->pre_amg, "lanczos: no preconditioner");
  NSS_REQUIRE(!s->pre_bjac || s->pre_bjac->n == s->n, "lanczos: block preconditioner size mismatch");
  NSS_REQUIRE(!s->pre_amg || s->pre_amg->levels.empty() || s->pre_amg->levels[0].n == s->n || s->pre_amg->T, "lanczos: AMG size mismatch");
  for (int i = 0; i < 3; ++i) NSS_REQUIRE(s->v[i] != nullptr, "lanczos: NULL vector");
  NSS_REQUIRE(s->z[0] && s->z[1] && s->p && s->scal && s->ctrl && s->hist && s->partials_a && s->partials_b,
              "lanczos: NULL buffer");
}

// z = pre_scale * pre x for everything that is not the fused point Jacobi; leaves the partials of <z, x> in
// partials_b and returns their count.  `scratch` (n doubles) is free for the multiplicative form.
static int lz_precondition(const nss_lanczos_t& s, const double* x, double* z, double* scratch, hipStream_t st) {
  const int32_t* done = s.ctrl;
  const bool multiplicative = s.pre_amg && s.pre_bjac && s.pre_bjac->gs_mat;
  if (!s.pre_amg && s.pre_bjac && !s.pre_bjac->gs_mat)          // block Jacobi: the dot comes out of the apply kernel
    return bjac_apply_dot(*s.pre_bjac, s.pre_scale, x, z, s.partials_b, done, st);
  if (multiplicative) {
    // MypreA with GS=True (templates/NavierStokesSIMPLE_iterative.py:376-381): y = 0; Smooth; r = x - A y; y += M r; SmoothBack
    NSS_HIP(hipMemsetAsync(z, 0, sizeof(double) * size_t(s.n), st));
    bjac_smooth(*s.pre_bjac, 1.0, x, z, false, done, st);
    launch_csr_stream(*s.A, z, EpiLzResidual{done, x, scratch}, st);
    amg_apply(*s.pre_amg, 1.0, scratch, z, st, done, true);
    bjac_smooth(*s.pre_bjac, 1.0, x, z, true, done, st);
    if (s.pre_scale != 1.0) {
      hipLaunchKernelGGL(lanczos_scale_kernel, dim3(lz_grid(s)), dim3(kBlock), 0, st, done, s.n, s.pre_scale, z);
      NSS_CHECK_LAUNCH();
    }
  } else if (s.pre_amg) {                                       // V-cycle [+ Jacobi part]: the additive MypreA (:383)
    amg_apply(*s.pre_amg, s.pre_scale, x, z, st, done);
    if (s.pre_bjac) bjac_apply(*s.pre_bjac, s.pre_scale, x, 1.0, z, done, st);
    if (s.pre_diag) diag_apply(s.n, s.pre_diag, s.pre_scale, x, 1.0, z, done, st);
  } else {                                                      // symmetric Gauss-Seidel sweep as an operator
    bjac_apply(*s.pre_bjac, s.pre_scale, x, 0.0, z, done, st);
  }
  const int nb = lz_grid(s);
  hipLaunchKernelGGL(lanczos_dot_kernel, dim3(nb), dim3(kBlock), 0, st, done, s.n, z, x, s.partials_b);
  NSS_CHECK_LAUNCH();
  return nb;
}

static void lz_step(const nss_lanczos_t& s, int j, hipStream_t st) {
  double* v = s.v[j % 3];
  double* v_old = s.v[(j + 2) % 3];
  double* v_new = s.v[(j + 1) % 3];
  double* z = s.z[j % 2];
  double* z_new = s.z[(j + 1) % 2];
  launch_csr_stream(*s.A, z, EpiLanczosP{s.ctrl, z, s.p, s.partials_a}, st);
  hipLaunchKernelGGL((lanczos_sum_kernel<0>), dim3(1), dim3(kLzSum), 0, st, s.ctrl, s.A->nblk, s.partials_a, s.scal,
                     int(L_DH), j, s.hist);
  NSS_CHECK_LAUNCH();
  const bool fused_pre = s.pre_diag && !s.pre_amg;
  LzArgs a{s.ctrl, s.scal, s.hist, s.n, j, s.p, v, v_old, s.pre_diag, v_new, z_new, s.pre_scale, s.partials_b};
  hipLaunchKernelGGL(lanczos_combine_kernel, dim3(lz_grid(s)), dim3(kBlock), 0, st, a, fused_pre ? 1 : 0);
  NSS_CHECK_LAUNCH();
  const int nb = fused_pre ? lz_grid(s) : lz_precondition(s, v_new, z_new, s.p, st);    // (p is free from here on)
  hipLaunchKernelGGL((lanczos_sum_kernel<1>), dim3(1), dim3(kLzSum), 0, st, s.ctrl, nb, s.partials_b, s.scal,
                     int(L_G2), j, s.hist);
  NSS_CHECK_LAUNCH();
}

}  // namespace nss

using namespace nss;

extern "C" {

int nss_lanczos_workspace(const nss_lanczos_t* s, int64_t* partials_a, int64_t* partials_b) {
  return guarded([&] {
    NSS_REQUIRE(s && s->A, "lanczos_workspace: NULL state / matrix");
    if (partials_a) *partials_a = s->A->nblk;
    if (partials_b) *partials_b = std::max<int64_t>(lz_grid(*s), s->pre_bjac ? bjac_dot_grid(*s->pre_bjac) : 0);
  });
}

int nss_lanczos_start(const nss_lanczos_t* s, nss_stream_t stream) {
  return guarded([&] {
    lz_check(s);
    hipStream_t st = as_stream(stream);
    // zh_0 = pre rh_0 (rh_0 = v[0], given), gamma_0 = sqrt|<zh_0, rh_0>| -> scal; control words cleared
    NSS_HIP(hipMemsetAsync(s->ctrl, 0, sizeof(int32_t) * 4, st));
    NSS_HIP(hipMemsetAsync(s->scal, 0, sizeof(double) * 8, st));
    int nb;
    if (s->pre_diag && !s->pre_amg) {
      diag_apply(s->n, s->pre_diag, s->pre_scale, s->v[0], 0.0, s->z[0], nullptr, st);
      nb = lz_grid(*s);
      hipLaunchKernelGGL(lanczos_dot_kernel, dim3(nb), dim3(kBlock), 0, st, s->ctrl, s->n, s->z[0], s->v[0], s->partials_b);
      NSS_CHECK_LAUNCH();
    } else {
      nb = lz_precondition(*s, s->v[0], s->z[0], s->p, st);
    }
    hipLaunchKernelGGL((lanczos_sum_kernel<2>), dim3(1), dim3(kLzSum), 0, st, s->ctrl, nb, s->partials_b, s->scal,
                       int(L_G2), 0, s->hist);
    NSS_CHECK_LAUNCH();
  });
}

int nss_lanczos_iterate(const nss_lanczos_t* s, int32_t j_begin, int32_t j_end, nss_stream_t stream) {
  return guarded([&] {
    lz_check(s);
    NSS_REQUIRE(j_begin >= 0 && j_end >= j_begin, "lanczos_iterate: bad step range");
    for (int j = j_begin; j < j_end; ++j) lz_step(*s, j, as_stream(stream));
  });
}

int nss_lanczos_poll(const nss_lanczos_t* s, int32_t* stop, int32_t* j_stop, int32_t* last_j, nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(s && s->ctrl, "lanczos_poll: NULL state");
    int32_t h[4] = {0, 0, 0, 0};
    NSS_HIP(hipMemcpyAsync(h, s->ctrl, sizeof h, hipMemcpyDeviceToHost, as_stream(stream)));
    NSS_HIP(hipStreamSynchronize(as_stream(stream)));
    if (stop) *stop = h[LC_STOP];
    if (j_stop) *j_stop = h[LC_JSTOP];
    if (last_j) *last_j = h[LC_LAST];
  });
}

}  // extern "C"
