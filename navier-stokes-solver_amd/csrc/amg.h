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
};

namespace nss {
// x = V(bscale * b); every kernel of the cycle returns at once when `done` (device int, may be NULL)
// is non-zero: a solver that has stopped leaves x untouched
void amg_apply(const nss_amg_s& a, double bscale, const double* b, double* x, hipStream_t st,
               const int32_t* done = nullptr);
}  // namespace nss
