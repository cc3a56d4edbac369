// Shared pieces of the row-partitioned native loops (dist.hip: BPCG v2; minres.hip: MINRES).
#pragma once

#include "bpcg2.h"
#include "p2p.h"

#include <vector>

struct nss_dist_s {
  void* comm = nullptr;
  int nranks = 1, rank = 0;
  hipStream_t xstream = nullptr;
  hipEvent_t ev_ready[3] = {nullptr, nullptr, nullptr};  // operand produced on C
  hipEvent_t ev_halo[3] = {nullptr, nullptr, nullptr};   // ghost tail filled on X
  void* lib = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  // per-phase profile of the native loop (nss_dist_profile_begin / _end): kProfMarks events per iteration
  std::vector<hipEvent_t> prof_ev;
  int prof_cap = 0, prof_iters = 0;
  // mailbox transport (nss_dist_attach_p2p): exchange() and the one-double all-reduces of every native partitioned
  // loop go through peer-mapped memory instead of RCCL (csrc/p2p.h)
  struct nss_p2p_s* p2p = nullptr;
};


// Smoothed-aggregation V(1,1)-cycle on a row-partitioned operator with REPLICATED coarse levels, applied
// natively inside the partitioned loops (distributed.DistributedAMG is the host-side twin): the finest level
// works on the slab (two halo exchanges of the iterate), the slab's share of the restricted residual is
// all-reduced (one coarse vector) and levels 1.. run redundantly on every rank.
struct nss_dist_amg_s {
  const nss_dist_s* d = nullptr;
  const nss_csr_s* A = nullptr;        // slab rows, columns [owned | ghosts]
  const nss_csr_s* R = nullptr;        // coarse rows x owned columns
  const nss_csr_s* P = nullptr;        // owned rows x coarse columns
  const double* wdinv = nullptr;       // omega / diag(A) on the slab
  const nss_amg_s* coarse = nullptr;   // levels 1.. (replicated)
  nss_halo_t halo{};                   // of the iterate x (ext = x's halo-extended buffer)
  int32_t n = 0, nc = 0;
  double *res = nullptr, *rc = nullptr, *rc_local = nullptr, *ec = nullptr;   // work vectors (owned by the handle)
};

// The auxiliary-space term of MypreA, transform @ V(L) @ transform.T (templates/NavierStokesSIMPLE_iterative.py:336-337,
// 357,380,383), on slabs: the velocity slab's rows of `transform` (columns: this rank's nodal planes + ghosts), the nodal
// slab's rows of its transpose (columns: this rank's velocity dofs + ghosts) and the V-cycle on the stacked nodal
// Laplacian with replicated coarse levels (nss_dist_amg_s).  Per apply: halo exchange of the argument, of the nodal
// correction, the two of the V-cycle, and its coarse all-reduce.
struct nss_dist_aux_s {
  const nss_dist_s* d = nullptr;
  const nss_csr_s* TT = nullptr;       // nodal rows x velocity columns [owned | ghosts]
  const nss_csr_s* T = nullptr;        // velocity rows x nodal columns [owned | ghosts]
  const nss_dist_amg_s* amg = nullptr;
  nss_halo_t halo_x{};                 // of the velocity argument (ext = its halo-extended buffer)
  nss_halo_t halo_e{};                 // of the nodal correction
  nss_halo_t halo_y{};                 // of the loop's iterate t1 as A's operand (multiplicative form: residual x - A y)
  bool has_halo_y = false;
  int32_t n_u = 0, n_nodes = 0;
  double* r_aux = nullptr;             // nodal work vector (owned by the handle)
};

namespace nss {

// y (+)= scale * T V(L) T^T b on the slab
void dist_aux_apply(const nss_dist_aux_s& a, double scale, const double* b, double* y, bool accumulate, hipStream_t st,
                    const int32_t* done);

// y = scale * V(b) on the slab; every kernel returns at once when *done != 0 (the collectives still run on all
// ranks, on unchanged buffers)
void dist_amg_apply(const nss_dist_amg_s& a, double scale, const double* b, double* y, hipStream_t st,
                    const int32_t* done);

constexpr int kNcclFloat64 = 8;
constexpr int kNcclSum = 0;

void nccl_check(const nss_dist_s& d, int rc, const char* what);
void check_halo(const nss_halo_t* h, const nss_csr_s& mat, const char* name);
// halo exchange of one or two SpMV operands (one grouped send/recv phase) on stream `st`
void exchange(const nss_dist_s& d, const nss_halo_t& h, hipStream_t st, const nss_halo_t* second = nullptr);
// dst = sum over the ranks of src (n doubles, device); without a communicator (one rank) a copy
void allreduce_sum(const nss_dist_s& d, const double* src, double* dst, size_t n, hipStream_t st);

}  // namespace nss
