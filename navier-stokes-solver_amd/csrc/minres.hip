// Fused, device-resident iteration of preconditioned MINRES (reference loop: minres.py:96-144)
// for K = [[A, B^T], [B, 0]] and C = diag(preA, preS).
//
//   MK1  rows of A   : kz0 = A z0                                                      (:97)
//   MK2  rows of B^T : kz0 += B^T z1, partial <kz0, z0>                                (:97-98)
//   MK3  rows of B   : kz1 = B z0,    partial <kz1, z1>                                (:97-98)
//   SUM  delta
//   MK4  element-wise: v_new = kz - delta v - gamma v_old (:99); z_new = C v_new (:101, Jacobi
//        preA and preS fused; block-Jacobi preA as its own kernel); partial <z_new, v_new> (:103)
//   SUM  gamma_new^2
//   SC   one lane: Givens recurrences (:107-113), ResNorm (:122), hist, both stop rules
//   MK5  element-wise: z_new, v_new *= 1/gamma_new (:104-105); w_new = (z - a3 w_old - a2 w)/a1
//        (:115-116); u += c_new eta_old w_new (:118)
//
// Stop handling: SC of iteration k records {stop, k_stop}; kernels of iteration k' return at
// once when stop is set and k' > k_stop, so MK5 of the stopping iteration still applies the
// update of u (the reference tests *after* :118) and nothing later touches the state.
#include "bpcg2.h"

#include <algorithm>

namespace nss {

enum {
  M_DELTA = 0, M_GAMMA = 1, M_G2 = 2, M_ETA_OLD = 3, M_C_OLD = 4, M_C = 5, M_S_OLD = 6, M_S = 7,
  M_RES_OLD = 8, M_ERR0 = 9, M_TOL = 10, M_A1 = 11, M_A2 = 12, M_A3 = 13, M_UCOEF = 14, M_INVG = 15
};
enum { MC_STOP = 0, MC_KSTOP = 1, MC_REASON = 2, MC_LASTK = 3 };

__device__ __forceinline__ bool minres_skip(const int32_t* ctrl, int k) {
  return ctrl[MC_STOP] != 0 && k > ctrl[MC_KSTOP];
}

struct EpiMStore {  // y = A x
  const int32_t* __restrict__ ctrl;
  int k;
  double* __restrict__ y;
  __device__ bool skip() const { return minres_skip(ctrl, k); }
  __device__ void row(int r, double ax) const { y[r] = ax; }
  __device__ void finish(int, double*) const {}
};

struct EpiMAccDot {  // y (+)= A x ; partial <y, z>
  const int32_t* __restrict__ ctrl;
  int k;
  int accumulate;
  double* __restrict__ y;
  const double* __restrict__ z;
  double* __restrict__ partials;
  double acc = 0.0;
  __device__ bool skip() const { return minres_skip(ctrl, k); }
  struct Pre { double y = 0.0, z = 0.0; };
  __device__ Pre fetch(int r) const { return Pre{accumulate ? y[r] : 0.0, z[r]}; }
  __device__ void row(int r, double ax, const Pre& p) {
    const double t = accumulate ? p.y + ax : ax;
    y[r] = t;
    acc = fma(t, p.z, acc);
  }
  __device__ void finish(int b, double* lds) {
    const double s = block_sum(acc, lds);
    if (threadIdx.x == 0 && b >= 0) partials[b] = s;
  }
};

constexpr int kMSum = 1024;
// scal[slot] = sum(pa[0..na)) + sum(pb[0..nb)) in a fixed order
__global__ __launch_bounds__(kMSum) void minres_sum_kernel(const int32_t* __restrict__ ctrl, int k, int na,
                                                            const double* __restrict__ pa, int nb,
                                                            const double* __restrict__ pb, double* __restrict__ scal,
                                                            int slot) {
  __shared__ double lds[2 * kMSum / kWave];
  if (minres_skip(ctrl, k)) return;
  const int tid = threadIdx.x;
  double a = 0.0, a2 = 0.0, b = 0.0, b2 = 0.0;
  int i = tid;
  for (; i + kMSum < na; i += 2 * kMSum) {
    a += pa[i];
    a2 += pa[i + kMSum];
  }
  for (; i < na; i += kMSum) a += pa[i];
  i = tid;
  for (; i + kMSum < nb; i += 2 * kMSum) {
    b += pb[i];
    b2 += pb[i + kMSum];
  }
  for (; i < nb; i += kMSum) b += pb[i];
  const double sa = wave_sum(a + a2), sb = wave_sum(b + b2);
  const int lane = tid & (kWave - 1), wave = tid >> 6;
  if (lane == 0) {
    lds[wave] = sa;
    lds[kMSum / kWave + wave] = sb;
  }
  __syncthreads();
  if (tid == 0) {
    double ta = 0.0, tb = 0.0;
    for (int w = 0; w < kMSum / kWave; ++w) {
      ta += lds[w];
      tb += lds[kMSum / kWave + w];
    }
    scal[slot] = ta + tb;
  }
}

struct MK4Args {
  const int32_t* ctrl;
  const double* scal;
  int32_t n_u, n_p, k;
  const double *kz0, *kz1, *v0, *v1, *vo0, *vo1;
  double *vn0, *vn1, *zn0, *zn1;
  const double *dinv, *minv;  // dinv == nullptr: block-Jacobi handles zn0 afterwards
  double* partials;
};

__global__ __launch_bounds__(kBlock) void minres_k4_kernel(MK4Args a) {
  __shared__ double lds[kBlock / kWave];
  if (minres_skip(a.ctrl, a.k)) return;
  const double delta = a.scal[M_DELTA], gamma = a.scal[M_GAMMA];
  const int stride = gridDim.x * kBlock;
  double acc = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < a.n_u; i += stride) {
    const double vn = fma(-gamma, a.vo0[i], fma(-delta, a.v0[i], a.kz0[i]));
    a.vn0[i] = vn;
    if (a.dinv) {
      const double zn = a.dinv[i] * vn;
      a.zn0[i] = zn;
      acc = fma(zn, vn, acc);
    }
  }
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < a.n_p; i += stride) {
    const double vn = fma(-gamma, a.vo1[i], fma(-delta, a.v1[i], a.kz1[i]));
    const double zn = a.minv[i] * vn;
    a.vn1[i] = vn;
    a.zn1[i] = zn;
    acc = fma(zn, vn, acc);
  }
  const double s = block_sum(acc, lds);
  if (threadIdx.x == 0) a.partials[blockIdx.x] = s;
}

// partial <x, y> of the velocity block after a block-Jacobi apply
__global__ __launch_bounds__(kBlock) void minres_dot_kernel(const int32_t* __restrict__ ctrl, int k, int32_t n,
                                                             const double* __restrict__ x,
                                                             const double* __restrict__ y,
                                                             double* __restrict__ partials) {
  __shared__ double lds[kBlock / kWave];
  if (minres_skip(ctrl, k)) return;
  const int stride = gridDim.x * kBlock;
  double acc = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) acc = fma(x[i], y[i], acc);
  const double s = block_sum(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

__global__ void minres_scalar_kernel(int32_t* __restrict__ ctrl, double* __restrict__ s, double* __restrict__ hist,
                                     int k) {
  if (threadIdx.x != 0 || minres_skip(ctrl, k)) return;
  const double delta = s[M_DELTA], gamma = s[M_GAMMA];
  const double gamma_new = sqrt(s[M_G2]);                              // :103
  const double c = s[M_C], c_old = s[M_C_OLD], sn = s[M_S], s_old = s[M_S_OLD];
  const double alpha0 = c * delta - c_old * sn * gamma;                // :107
  const double alpha1 = sqrt(alpha0 * alpha0 + gamma_new * gamma_new);
  const double alpha2 = sn * delta + c_old * c * gamma;
  const double alpha3 = s_old * gamma;
  const double c_new = alpha0 / alpha1;                                // :112
  const double s_new = gamma_new / alpha1;
  const double eta_old = s[M_ETA_OLD];
  s[M_A1] = alpha1;
  s[M_A2] = alpha2;
  s[M_A3] = alpha3;
  s[M_UCOEF] = c_new * eta_old;                                        // :118
  s[M_INVG] = 1.0 / gamma_new;                                         // :104-105
  const double res = fabs(s_new) * s[M_RES_OLD];                       // :122
  hist[k] = res / s[M_ERR0];                                           // :125
  ctrl[MC_LASTK] = k;
  // shift the scalars (:135-144)
  s[M_ETA_OLD] = -s_new * eta_old;
  s[M_S_OLD] = sn;
  s[M_S] = s_new;
  s[M_C_OLD] = c;
  s[M_C] = c_new;
  s[M_GAMMA] = gamma_new;
  s[M_RES_OLD] = res;
  if (res < s[M_TOL] * s[M_ERR0]) {                                    // relative break (:126)
    ctrl[MC_KSTOP] = k;
    ctrl[MC_REASON] = 1;
    ctrl[MC_STOP] = 1;
  } else if (!(res > s[M_TOL])) {                                      // absolute guard of the while (:96)
    ctrl[MC_KSTOP] = k;
    ctrl[MC_REASON] = 2;
    ctrl[MC_STOP] = 1;
  }
}

struct MK5Args {
  const int32_t* ctrl;
  const double* scal;
  int32_t n_u, n_p, k;
  double *zn0, *zn1, *vn0, *vn1, *wn0, *wn1, *u0, *u1;
  const double *z0, *z1, *wo0, *wo1, *w0, *w1;
};

__device__ __forceinline__ void minres_k5_body(int i, double invg, double a1inv, double a2, double a3, double uc,
                                                double* zn, double* vn, double* wn, double* u, const double* z,
                                                const double* wo, const double* w) {
  zn[i] *= invg;
  NSS_ST(vn[i], vn[i] * invg);
  double t = fma(-a2, w[i], fma(-a3, wo[i], z[i]));   // :115
  t *= a1inv;                                         // :116
  NSS_ST(wn[i], t);                                   // next read one iteration later
  NSS_ST(u[i], fma(uc, t, u[i]));                     // :118
}

__global__ __launch_bounds__(kBlock) void minres_k5_kernel(MK5Args a) {
  if (minres_skip(a.ctrl, a.k)) return;
  const double invg = a.scal[M_INVG], a1inv = 1.0 / a.scal[M_A1], a2 = a.scal[M_A2], a3 = a.scal[M_A3];
  const double uc = a.scal[M_UCOEF];
  const int stride = gridDim.x * kBlock;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < a.n_u; i += stride)
    minres_k5_body(i, invg, a1inv, a2, a3, uc, a.zn0, a.vn0, a.wn0, a.u0, a.z0, a.wo0, a.w0);
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < a.n_p; i += stride)
    minres_k5_body(i, invg, a1inv, a2, a3, uc, a.zn1, a.vn1, a.wn1, a.u1, a.z1, a.wo1, a.w1);
}

static int m_grid(const nss_minres_t& s) { return stream_grid(int64_t(s.n_u) + s.n_p, kBlock * 4); }
static int m_dot_grid(const nss_minres_t& s) { return stream_grid(s.n_u, kBlock * 4); }

static void minres_check(const nss_minres_t* s) {
  NSS_REQUIRE(s != nullptr, "minres: NULL state");
  NSS_REQUIRE(s->A && s->B && s->BT, "minres: NULL matrix handle");
  NSS_REQUIRE(s->A->m == s->n_u && s->BT->m == s->n_u && s->B->m == s->n_p, "minres: matrix rows do not match n_u/n_p");
  NSS_REQUIRE(s->A->n == s->n_u && s->B->n == s->n_u && s->BT->n == s->n_p, "minres: matrix columns do not match n_u/n_p");
  NSS_REQUIRE(!(s->pre_diag && s->pre_bjac), "minres: pre_diag and pre_bjac are exclusive");
  NSS_REQUIRE(s->pre_diag || s->pre_bjac || s->pre_amg, "minres: no preconditioner for the velocity block");
  NSS_REQUIRE(!s->pre_amg || s->pre_amg->levels[0].n == s->n_u, "minres: AMG size mismatch");
  NSS_REQUIRE(!(s->pre_amg && s->pre_bjac && s->pre_bjac->gs_mat), "minres: AMG + Gauss-Seidel mode is not additive");
  NSS_REQUIRE(!s->pre_bjac || s->pre_bjac->n == s->n_u, "minres: block-Jacobi size mismatch");
  NSS_REQUIRE(s->minv && s->scal && s->ctrl && s->hist && s->partials_a && s->partials_b && s->partials_c,
              "minres: NULL work buffer");
  for (int c = 0; c < 2; ++c) {
    NSS_REQUIRE(s->u[c] && s->kz[c] && s->z[0][c] && s->z[1][c], "minres: NULL vector");
    for (int j = 0; j < 3; ++j) NSS_REQUIRE(s->v[j][c] && s->w[j][c], "minres: NULL ring vector");
  }
}

static void minres_iteration(const nss_minres_t& s, int k, hipStream_t st) {
  const int io = (k + 2) % 3, ic = k % 3, in = (k + 1) % 3;   // old, current, new
  const int zc = k % 2, zn = (k + 1) % 2;
  launch_csr_stream(*s.A, s.z[zc][0], EpiMStore{s.ctrl, k, s.kz[0]}, st);
  launch_csr_stream(*s.BT, s.z[zc][1], EpiMAccDot{s.ctrl, k, 1, s.kz[0], s.z[zc][0], s.partials_a}, st);
  launch_csr_stream(*s.B, s.z[zc][0], EpiMAccDot{s.ctrl, k, 0, s.kz[1], s.z[zc][1], s.partials_b}, st);
  hipLaunchKernelGGL(minres_sum_kernel, dim3(1), dim3(kMSum), 0, st, s.ctrl, k, s.BT->nblk, s.partials_a,
                     s.B->nblk, s.partials_b, s.scal, int(M_DELTA));
  NSS_CHECK_LAUNCH();
  MK4Args a4{s.ctrl, s.scal, s.n_u, s.n_p, k, s.kz[0], s.kz[1], s.v[ic][0], s.v[ic][1], s.v[io][0], s.v[io][1],
             s.v[in][0], s.v[in][1], s.z[zn][0], s.z[zn][1], s.pre_amg ? nullptr : s.pre_diag, s.minv, s.partials_c};
  hipLaunchKernelGGL(minres_k4_kernel, dim3(m_grid(s)), dim3(kBlock), 0, st, a4);
  NSS_CHECK_LAUNCH();
  int nb2 = 0;
  if (s.pre_bjac || s.pre_amg) {
    // z_new[0] = preA v_new[0] outside the element-wise kernel; after the stop these launches only
    // touch ring slots nobody reads any more
    if (s.pre_amg) {
      amg_apply(*s.pre_amg, 1.0, s.v[in][0], s.z[zn][0], st);
      if (s.pre_bjac) bjac_apply(*s.pre_bjac, 1.0, s.v[in][0], 1.0, s.z[zn][0], nullptr, st);
      if (s.pre_diag) {
        const int rc = nss_diag_apply_f64(s.n_u, s.pre_diag, 1.0, s.v[in][0], 1.0, s.z[zn][0], st);
        if (rc != 0) throw Error(nss_last_error());
      }
    } else if (s.pre_bjac->gs_mat) {
      bjac_apply(*s.pre_bjac, 1.0, s.v[in][0], 0.0, s.z[zn][0], nullptr, st);
    } else {                          // block Jacobi: <z_new, v_new> comes out of the apply kernel
      nb2 = bjac_apply_dot(*s.pre_bjac, 1.0, s.v[in][0], s.z[zn][0], s.partials_a, nullptr, st);
    }
    if (nb2 == 0) {
      nb2 = m_dot_grid(s);
      hipLaunchKernelGGL(minres_dot_kernel, dim3(nb2), dim3(kBlock), 0, st, s.ctrl, k, s.n_u, s.z[zn][0], s.v[in][0],
                         s.partials_a);
      NSS_CHECK_LAUNCH();
    }
  }
  hipLaunchKernelGGL(minres_sum_kernel, dim3(1), dim3(kMSum), 0, st, s.ctrl, k, nb2, s.partials_a, m_grid(s),
                     s.partials_c, s.scal, int(M_G2));
  NSS_CHECK_LAUNCH();
  hipLaunchKernelGGL(minres_scalar_kernel, dim3(1), dim3(kWave), 0, st, s.ctrl, s.scal, s.hist, k);
  NSS_CHECK_LAUNCH();
  MK5Args a5{s.ctrl, s.scal, s.n_u, s.n_p, k, s.z[zn][0], s.z[zn][1], s.v[in][0], s.v[in][1], s.w[in][0],
             s.w[in][1], s.u[0], s.u[1], s.z[zc][0], s.z[zc][1], s.w[io][0], s.w[io][1], s.w[ic][0], s.w[ic][1]};
  hipLaunchKernelGGL(minres_k5_kernel, dim3(m_grid(s)), dim3(kBlock), 0, st, a5);
  NSS_CHECK_LAUNCH();
}

}  // namespace nss

using namespace nss;

extern "C" {

int nss_minres_workspace(const nss_minres_t* s, int64_t* partials_a, int64_t* partials_b, int64_t* partials_c) {
  return guarded([&] {
    NSS_REQUIRE(s && s->A && s->B && s->BT, "minres_workspace: NULL state / matrices");
    int64_t dotg = m_dot_grid(*s);
    if (s->pre_bjac) dotg = std::max<int64_t>(dotg, bjac_dot_grid(*s->pre_bjac));
    if (partials_a) *partials_a = std::max<int64_t>(s->BT->nblk, dotg);
    if (partials_b) *partials_b = s->B->nblk;
    if (partials_c) *partials_c = m_grid(*s);
  });
}

int nss_minres_iterate(const nss_minres_t* s, int32_t k_begin, int32_t k_end, nss_stream_t stream) {
  return guarded([&] {
    minres_check(s);
    NSS_REQUIRE(k_begin >= 1, "minres_iterate: iterations are counted from 1");
    for (int k = k_begin; k < k_end; ++k) minres_iteration(*s, k, as_stream(stream));
  });
}

int nss_minres_poll(const nss_minres_t* s, int32_t* stop, int32_t* k_stop, int32_t* reason, int32_t* last_k,
                    nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(s && s->ctrl, "minres_poll: NULL state");
    int32_t h[4] = {0, 0, 0, 0};
    NSS_HIP(hipMemcpyAsync(h, s->ctrl, sizeof h, hipMemcpyDeviceToHost, as_stream(stream)));
    NSS_HIP(hipStreamSynchronize(as_stream(stream)));
    if (stop) *stop = h[MC_STOP];
    if (k_stop) *k_stop = h[MC_KSTOP];
    if (reason) *reason = h[MC_REASON];
    if (last_k) *last_k = h[MC_LASTK];
  });
}

}  // extern "C"
