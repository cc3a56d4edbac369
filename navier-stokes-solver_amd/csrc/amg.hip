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

void amg_apply(const nss_amg_s& a, double bscale, const double* b, double* x, hipStream_t st, const int32_t* done,
               bool accumulate) {
  if (a.T) {                                         // auxiliary-space mode: x (+)= T (sum_c V_c) T^T (bscale b)
    launch_csr_stream(*a.TT, b, EpiAxpby{bscale, 0.0, a.aux_r, done}, st);
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
    for (auto& lv : a->levels) {
      (void)hipFree(lv.x);
      (void)hipFree(lv.r);
      (void)hipFree(lv.b);
      (void)hipFree(lv.y);
    }
    delete a;
  });
}

int nss_amg_apply_f64(nss_amg_t a, double bscale, const double* b, double* x, nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(a != nullptr, "amg_apply: NULL handle");
    NSS_REQUIRE(b != x, "amg_apply: b must not alias x");
    amg_apply(*a, bscale, b, x, as_stream(stream));
  });
}

}  // extern "C"
