// AMG hierarchy handle shared by amg.hip and the fused loops.
#pragma once

#include "csr_stream.h"

#include <algorithm>
#include <vector>

namespace nss {
struct AmgLevel {
  int32_t n = 0;
  const nss_csr_s* A = nullptr;
  const nss_csr_s* P = nullptr;
  const nss_csr_s* R = nullptr;
  const double* dinv = nullptr;
  double *x = nullptr, *r = nullptr, *b = nullptr, *y = nullptr;   // work vectors of the level
};
}  // namespace nss

struct nss_amg_s {
  std::vector<nss::AmgLevel> levels;
  const nss_csr_s* coarse_inverse = nullptr;
  double omega = 2.0 / 3.0;
  // Auxiliary-space mode (nss_amg_create_auxiliary): x -> T (sum_c E_c V_c E_c^T) T^T x, the term
  // `transform @ preAh1 @ transform.T` of the reference's MypreA
  // (templates/NavierStokesSIMPLE_iterative.py:291,320-357,380,383): T maps the auxiliary space (one
  // P1-like scalar space per velocity component, stacked) to the velocity dofs, V_c is the V-cycle of
  // component c's Laplacian.  levels[0].n = rows of T, so the fused loops' size checks hold.
  const nss_csr_s* T = nullptr;
  const nss_csr_s* TT = nullptr;
  std::vector<const nss_amg_s*> comps;
  std::vector<int32_t> comp_off;     // comps.size() + 1 offsets into the stacked auxiliary vector
  double *aux_r = nullptr, *aux_z = nullptr;
  // Components that share ONE hierarchy (the same Laplacian with the same boundary conditions for every velocity
  // component: the usual case) are cycled TOGETHER: every level operator is read once for all K right-hand sides
  // (amg.hip: csr_multi_kernel).  The cycle's internal vectors are interleaved [row][K]; the stacked auxiliary vectors
  // at its two ends are addressed with strides.  Work vectors per level of the shared hierarchy (K * n doubles each):
  struct MultiLevel {
    double *x = nullptr, *r = nullptr, *b = nullptr, *y = nullptr;
  };
  std::vector<MultiLevel> multi;     // empty: components are cycled one after the other
  // one 16-byte descriptor {first row, end row, first entry, entries} per row block of every matrix the joint cycle
  // multiplies with: a workgroup of csr_multi_kernel then starts its matrix stream after ONE (scalar) load instead of
  // the rowblk -> rowptr chain
  struct MultiDesc {
    const nss_csr_s* mat;
    const int32_t* rowblk;           // the launch plan the descriptors were made from (checked at every launch)
    int32_t* desc;
  };
  std::vector<MultiDesc> multi_desc;
};

namespace nss {
struct EpiResidual {   // r = b - A x
  const double* __restrict__ b;
  double* __restrict__ r;
  const int32_t* __restrict__ done;
  __device__ bool skip() const { return done != nullptr && *done != 0; }
  struct Pre { double b = 0.0; };
  __device__ Pre fetch(int i) const { return Pre{b[i]}; }
  __device__ void row(int i, double ax, const Pre& p) const { r[i] = p.b - ax; }
  __device__ void finish(int, double*) const {}
};

struct EpiJacobi {     // y (+)= scale * (x + w dinv (b - A x))
  const double* __restrict__ b;
  const double* __restrict__ x;
  const double* __restrict__ dinv;
  double* __restrict__ y;
  double w;
  double scale;
  const int32_t* __restrict__ done;
  bool accumulate = false;
  __device__ bool skip() const { return done != nullptr && *done != 0; }
  struct Pre { double b = 0.0, x = 0.0, dinv = 0.0, y = 0.0; };
  __device__ Pre fetch(int i) const { return Pre{b[i], x[i], dinv[i], accumulate ? y[i] : 0.0}; }
  __device__ void row(int i, double ax, const Pre& p) const {
    const double t = scale * fma(w * p.dinv, p.b - ax, p.x);
    y[i] = accumulate ? p.y + t : t;
  }
  __device__ void finish(int, double*) const {}
};

// x = V(bscale * b); every kernel of the cycle returns at once when `done` (device int, may be NULL)
// is non-zero: a solver that has stopped leaves x untouched.  `accumulate`: x += V(bscale * b).
void amg_apply(const nss_amg_s& a, double bscale, const double* b, double* x, hipStream_t st,
               const int32_t* done = nullptr, bool accumulate = false);
}  // namespace nss
