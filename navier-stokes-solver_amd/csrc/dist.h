// Shared pieces of the row-partitioned native loops (dist.hip: BPCG v2; minres.hip: MINRES).
#pragma once

#include "bpcg2.h"

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
};


namespace nss {

constexpr int kNcclFloat64 = 8;
constexpr int kNcclSum = 0;

void nccl_check(const nss_dist_s& d, int rc, const char* what);
void check_halo(const nss_halo_t* h, const nss_csr_s& mat, const char* name);
// halo exchange of one or two SpMV operands (one grouped send/recv phase) on stream `st`
void exchange(const nss_dist_s& d, const nss_halo_t& h, hipStream_t st, const nss_halo_t* second = nullptr);
// dst = sum over the ranks of src (n doubles, device); without a communicator (one rank) a copy
void allreduce_sum(const nss_dist_s& d, const double* src, double* dst, size_t n, hipStream_t st);

}  // namespace nss
