#pragma once

#include "amg.h"
#include "precond.h"

namespace nss {
inline void bjac_apply_guarded(const nss_bjac_s& j, double k, const double* x, double* y, const int32_t* done,
                               hipStream_t st) {
  bjac_apply(j, k, x, 0.0, y, done, st);   // additive block Jacobi, or the symmetric GS sweep
}

// pieces of the fused BPCG iteration shared with the row-partitioned loop (dist.hip)
void bpcg2_check_state(const nss_bpcg2_t* s);
void bpcg2_phase(const nss_bpcg2_t& s, int which, int it, hipStream_t st);
// `ghost_tail`: K2 also forms t4 on B's ghost columns (needs t1's ghosts: only in a launch ordered
// after their arrival)
void bpcg2_spmv_phase(const nss_bpcg2_t& s, int which, int it, hipStream_t st, int b0, int b1, bool ghost_tail = true);
void bpcg2_k1_finish(const nss_bpcg2_t& s, hipStream_t st);
// the size rule / override of nss_bpcg2_fuse_block_jacobi for a system with `rows` velocity rows
bool fuse_block_jacobi_wanted(int64_t rows);
void bpcg2_cphase(const nss_bpcg2_t& s, int which, int it, hipStream_t st);
void gather_launch(int64_t n, const int32_t* idx, const double* src, double* dst, hipStream_t st);
}  // namespace nss
