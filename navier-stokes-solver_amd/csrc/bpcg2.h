#pragma once

#include "precond.h"

namespace nss {
inline void bjac_apply_guarded(const nss_bjac_s& j, double k, const double* x, double* y, const int32_t* done,
                               hipStream_t st) {
  bjac_apply(j, k, x, 0.0, y, done, st);
}
}  // namespace nss
