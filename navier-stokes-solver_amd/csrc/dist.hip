// Row-partitioned BPCG iteration with the communication issued natively.
//
// Why native: driving the halo exchanges and the two all-reduces of an iteration from Python costs
// ~100 us of host time per iteration; with 8 GPUs the device side of an iteration of the
// 1e7-DoF system is ~150 us, so the loop has to be issued from C.  RCCL is resolved at run time
// from the librccl already loaded in the process (dlopen/dlsym: no link-time dependency; the
// communicator itself is created by the host and passed in).
//
// Per iteration (compute stream C, communication stream X, events):
//   X: wait(C) . pack(s1) . group{send,recv} ;  C: K1 interior . wait(X) . K1 boundary . [block-Jacobi]
//   X: wait(C) . pack(t1) . group{send,recv} ;  C: K2 interior . wait(X) . K2 boundary
//   X: wait(C) . pack(t4) . group{send,recv} ;  C: K3 interior . wait(X) . K3 boundary   (ghost mode: no exchange)
//   C: SUM1 . allreduce(as_s) . K4 . SUM2 . allreduce(wdn) . K5
// States laid out for it (nss_bpcg2_t.dist_compact) run the COMPACT plan on C instead:
//   C1 (books of the previous iteration from the all-reduced <w, d>; rows of B^T; ghost copies of s0) . preA .
//   exchange(t1) . C23 (rows of A | owned + ghost rows of B on t1 - s0) . sum . allreduce(as_s) . C4 . sum .
//   allreduce(wdn)                                  -- 6 launches + 3 collectives instead of 9 + 3.
// Interior row blocks touch no ghost column, so they overlap the exchange; xGMI is
// point-to-point and only slab neighbours talk.  With overlap == 0 everything runs on C.
#include "dist.h"

#include <dlfcn.h>

namespace nss {

// marks of one profiled iteration: segment i = [mark i, mark i + 1)
//   0 K1 (B^T rows, incl. the s1 exchange when that operand is not kept by recurrence) + preA
//   1 halo exchange of t1      2 K2 (A rows)      3 K3 (B rows, incl. the t4 exchange if any) + local sum
//   4 all-reduce <s, K s>      5 K4 (+ ghost rows of B) + local sum      6 all-reduce <w, d>      7 K5
// compact plan: 0 C1 + preA   1 exchange of t1   2 C23   3 local sum   4 all-reduce   5 C4 + local sum   6 all-reduce   7 -
constexpr int kProfMarks = 9;

enum { S_AS_SLOT = 1, S_WDN_SLOT = 2, S_LOCAL_OFFSET = 8 };   // local sums: slots 9 / 10 (bpcg2.hip)

void nccl_check(const nss_dist_s& d, int rc, const char* what) {
  if (rc != 0) throw Error(std::string(what) + ": " + (d.GetErrorString ? d.GetErrorString(rc) : "RCCL error"));
}

template <class F>
static void resolve(void* lib, const char* name, F& out) {
  out = reinterpret_cast<F>(dlsym(lib, name));
  if (!out) throw Error(std::string("librccl does not export ") + name);
}

void check_halo(const nss_halo_t* h, const nss_csr_s& mat, const char* name) {
  NSS_REQUIRE(h != nullptr, std::string(name) + ": NULL halo");
  NSS_REQUIRE(h->ext != nullptr, std::string(name) + ": NULL operand buffer");
  NSS_REQUIRE(h->n_pack >= 0 && h->n_send >= 0 && h->n_recv >= 0, std::string(name) + ": negative count");
  NSS_REQUIRE(h->n_pack == 0 || (h->send_idx && h->sendbuf), std::string(name) + ": NULL pack buffers");
  if (h->direct) {   // segments are sent straight out of the operand: no pack, offsets into ext
    NSS_REQUIRE(h->n_pack == 0, std::string(name) + ": direct sends take no pack list");
    for (int i = 0; i < h->n_send; ++i)
      NSS_REQUIRE(h->h_send_cnt[i] > 0 && h->h_send_off[i] >= 0 && h->h_send_off[i] + h->h_send_cnt[i] <= mat.n,
                  std::string(name) + ": send segment outside the operand");
    for (int i = 0; i < h->n_recv; ++i)
      NSS_REQUIRE(h->h_recv_cnt[i] > 0 && h->h_recv_off[i] + h->h_recv_cnt[i] <= mat.n,
                  std::string(name) + ": receive segment outside the operand");
    return;
  }
  NSS_REQUIRE(h->int_begin >= 0 && h->int_begin <= h->int_end && h->int_end <= mat.nblk,
              std::string(name) + ": interior row-block range out of bounds");
  int64_t packed = 0;
  for (int i = 0; i < h->n_send; ++i) {
    NSS_REQUIRE(h->h_send_cnt[i] > 0 && h->h_send_off[i] == packed, std::string(name) + ": send segments not contiguous");
    packed += h->h_send_cnt[i];
  }
  NSS_REQUIRE(packed == h->n_pack, std::string(name) + ": send counts do not add up to n_pack");
  for (int i = 0; i < h->n_recv; ++i)
    NSS_REQUIRE(h->h_recv_cnt[i] > 0 && h->h_recv_off[i] + h->h_recv_cnt[i] <= mat.n,
                std::string(name) + ": receive segment outside the operand");
}

// pack + grouped send/recv of up to two operands on stream `st` (one RCCL group: one exchange phase)
void exchange(const nss_dist_s& d, const nss_halo_t& h, hipStream_t st, const nss_halo_t* second) {
  const nss_halo_t* hs[2] = {&h, second};
  bool any = false;
  for (const nss_halo_t* p : hs)
    if (p && (p->n_send > 0 || p->n_recv > 0)) {
      any = true;
      if (!p->direct) gather_launch(p->n_pack, p->send_idx, p->ext, p->sendbuf, st);
    }
  if (!any || d.nranks <= 1) return;
  if (d.p2p) {                       // mailbox transport: put + wait/copy per operand layout
    for (const nss_halo_t* p : hs)
      if (p && (p->n_send > 0 || p->n_recv > 0)) p2p_exchange(*d.p2p, *p, nullptr, st);
    return;
  }
  nccl_check(d, d.GroupStart(), "ncclGroupStart");
  for (const nss_halo_t* p : hs) {
    if (!p) continue;
    const double* src = p->direct ? p->ext : p->sendbuf;
    for (int i = 0; i < p->n_send; ++i)
      nccl_check(d, d.Send(src + p->h_send_off[i], size_t(p->h_send_cnt[i]), kNcclFloat64, p->h_send_peer[i], d.comm, st),
                 "ncclSend");
    for (int i = 0; i < p->n_recv; ++i)
      nccl_check(d, d.Recv(p->ext + p->h_recv_off[i], size_t(p->h_recv_cnt[i]), kNcclFloat64, p->h_recv_peer[i], d.comm, st),
                 "ncclRecv");
  }
  nccl_check(d, d.GroupEnd(), "ncclGroupEnd");
}

// dst[0 .. n) = sum over the ranks of src[0 .. n) (device pointers; out of place or in place)
void allreduce_sum(const nss_dist_s& d, const double* src, double* dst, size_t n, hipStream_t st) {
  if (d.p2p && d.nranks > 1) {
    if (n != 1) throw Error("dist: the mailbox transport all-reduces single doubles (vectors need the RCCL communicator)");
    p2p_allreduce(*d.p2p, src, dst, st);
    return;
  }
  if (d.comm == nullptr) {
    if (d.nranks > 1) throw Error("dist: no communicator");
    if (src != dst) NSS_HIP(hipMemcpyAsync(dst, src, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
    return;
  }
  nccl_check(d, d.AllReduce(src, dst, n, kNcclFloat64, kNcclSum, d.comm, st), "ncclAllReduce");
}

// one SpMV phase with its operand exchange, optionally overlapped.  `mark` (profiling, non-overlapped
// path only) is recorded between the exchange and the SpMV.
static void spmv_with_halo(const nss_bpcg2_t& s, const nss_dist_s& d, const nss_halo_t& h, int slot, int which,
                           int it, const nss_csr_s& mat, int overlap_mode, hipStream_t cs, hipEvent_t mark = nullptr) {
  const bool talks = d.nranks > 1 && (h.n_send > 0 || h.n_recv > 0);
  const bool overlap = overlap_mode != 0;
  if (!talks && overlap_mode < 2) {   // modes 2, 3: keep the split path without peers (tests, measurements)
    if (mark) NSS_HIP(hipEventRecord(mark, cs));
    bpcg2_spmv_phase(s, which, it, cs, 0, -1);
    return;
  }
  if (!overlap) {
    exchange(d, h, cs);
    if (mark) NSS_HIP(hipEventRecord(mark, cs));
    bpcg2_spmv_phase(s, which, it, cs, 0, -1);
    return;
  }
  if (mark) NSS_HIP(hipEventRecord(mark, cs));
  if (overlap_mode == 3) {             // measurement only: the split launches without the second stream
    exchange(d, h, cs);
    const bool has_suffix = h.int_end < mat.nblk;
    bpcg2_spmv_phase(s, which, it, cs, h.int_begin, h.int_end, !has_suffix && h.int_begin == 0);
    bpcg2_spmv_phase(s, which, it, cs, 0, h.int_begin, !has_suffix);
    bpcg2_spmv_phase(s, which, it, cs, h.int_end, mat.nblk, true);
    return;
  }
  NSS_HIP(hipEventRecord(d.ev_ready[slot], cs));
  NSS_HIP(hipStreamWaitEvent(d.xstream, d.ev_ready[slot], 0));
  exchange(d, h, d.xstream);
  NSS_HIP(hipEventRecord(d.ev_halo[slot], d.xstream));
  // interior rows touch no ghost column; K2's ghost tail (t4 on B's ghost columns, from t1's ghosts)
  // rides in exactly one launch that is ordered after the arrival of the halo
  const bool has_prefix = h.int_begin > 0, has_suffix = h.int_end < mat.nblk;
  if (!has_prefix && !has_suffix) {
    NSS_HIP(hipStreamWaitEvent(cs, d.ev_halo[slot], 0));
    bpcg2_spmv_phase(s, which, it, cs, h.int_begin, h.int_end, true);
    return;
  }
  bpcg2_spmv_phase(s, which, it, cs, h.int_begin, h.int_end, false);
  NSS_HIP(hipStreamWaitEvent(cs, d.ev_halo[slot], 0));
  bpcg2_spmv_phase(s, which, it, cs, 0, h.int_begin, !has_suffix);      // boundary prefix
  bpcg2_spmv_phase(s, which, it, cs, h.int_end, mat.nblk, true);        // boundary suffix
}

__global__ void dist_copy_slot_kernel(double* scal, int dst, int src) { scal[dst] = scal[src]; }

static void allreduce_slot(const nss_bpcg2_t& s, const nss_dist_s& d, int slot, hipStream_t cs) {
  if (d.nranks <= 1 && d.comm == nullptr) {
    if (s.local_sums) {   // one rank, no communicator: the "all-reduce" is a copy of the local sum
      hipLaunchKernelGGL(dist_copy_slot_kernel, dim3(1), dim3(1), 0, cs, s.scal, slot, slot + S_LOCAL_OFFSET);
      NSS_CHECK_LAUNCH();
    }
    return;
  }
  const double* local = s.local_sums ? s.scal + slot + S_LOCAL_OFFSET : s.scal + slot;
  if (d.p2p) {
    p2p_allreduce(*d.p2p, local, s.scal + slot, cs);
    return;
  }
  if (d.comm == nullptr) throw Error("dist: no communicator");
  nccl_check(d, d.AllReduce(local, s.scal + slot, 1, kNcclFloat64, kNcclSum, d.comm, cs), "ncclAllReduce");
}

void dist_amg_apply(const nss_dist_amg_s& a, double scale, const double* b, double* y, hipStream_t st,
                    const int32_t* done) {
  double* x = a.halo.ext;                                                    // owned entries first
  diag_apply(a.n, a.wdinv, 1.0, b, 0.0, x, done, st);                        // pre-smoothing from zero
  exchange(*a.d, a.halo, st);
  launch_csr_stream(*a.A, x, EpiResidual{b, a.res, done}, st);               // res = b - A x
  launch_csr_stream(*a.R, a.res, EpiAxpby{1.0, 0.0, a.rc_local, done}, st);  // this slab's share of R res
  // out of place: once *done is set the kernels return at once but the collectives still run -- an in-place
  // all-reduce would multiply the frozen rc by the number of ranks on every further iteration
  allreduce_sum(*a.d, a.rc_local, a.rc, size_t(a.nc), st);
  amg_apply(*a.coarse, 1.0, a.rc, a.ec, st, done);                           // levels 1.. on every rank
  launch_csr_stream(*a.P, a.ec, EpiAxpby{1.0, 1.0, x, done}, st);            // x += P e
  exchange(*a.d, a.halo, st);
  launch_csr_stream(*a.A, x, EpiJacobi{b, x, a.wdinv, y, 1.0, scale, done}, st);   // y = scale (x + w D^-1 (b - A x))
}

void dist_aux_apply(const nss_dist_aux_s& a, double scale, const double* b, double* y, bool accumulate, hipStream_t st,
                    const int32_t* done) {
  NSS_HIP(hipMemcpyAsync(a.halo_x.ext, b, sizeof(double) * size_t(a.n_u), hipMemcpyDeviceToDevice, st));   // (scratch: unguarded)
  exchange(*a.d, a.halo_x, st);
  launch_csr_stream(*a.TT, a.halo_x.ext, EpiAxpby{1.0, 0.0, a.r_aux, done}, st);                // transform.T
  dist_amg_apply(*a.amg, 1.0, a.r_aux, a.halo_e.ext, st, done);                                 // V-cycle on the stacked Laplacian
  exchange(*a.d, a.halo_e, st);
  launch_csr_stream(*a.T, a.halo_e.ext, EpiAxpby{scale, accumulate ? 1.0 : 0.0, y, done}, st);  // transform
}

}  // namespace nss

using namespace nss;

extern "C" {

int nss_dist_aux_create(nss_dist_t d, nss_csr_t tt_loc, const nss_halo_t* halo_x, nss_csr_t t_loc, const nss_halo_t* halo_e,
                        nss_dist_amg_t amg, const nss_halo_t* halo_y, nss_dist_aux_t* out) {
  return guarded([&] {
    NSS_REQUIRE(d && tt_loc && halo_x && t_loc && halo_e && amg && out, "dist_aux_create: NULL argument");
    const int32_t n_u = t_loc->m, n_nodes = tt_loc->m;
    NSS_REQUIRE(tt_loc->n >= n_u && t_loc->n >= n_nodes && amg->n == n_nodes, "dist_aux_create: operator shapes do not chain");
    check_halo(halo_x, *tt_loc, "dist_aux halo_x");
    check_halo(halo_e, *t_loc, "dist_aux halo_e");
    NSS_REQUIRE(amg->d == d, "dist_aux_create: the V-cycle belongs to another communicator handle");
    nss_dist_aux_s* h = new nss_dist_aux_s;
    try {
      h->d = d;
      h->TT = tt_loc;
      h->T = t_loc;
      h->amg = amg;
      h->halo_x = *halo_x;
      h->halo_e = *halo_e;
      if (halo_y) {
        h->halo_y = *halo_y;
        h->has_halo_y = true;
      }
      h->n_u = n_u;
      h->n_nodes = n_nodes;
      NSS_HIP(hipMalloc(&h->r_aux, sizeof(double) * size_t(std::max(1, n_nodes))));
    } catch (...) {
      nss_dist_aux_destroy(h);
      throw;
    }
    *out = h;
  });
}

int nss_dist_aux_destroy(nss_dist_aux_t h) {
  return guarded([&] {
    if (!h) return;
    (void)hipFree(h->r_aux);
    delete h;
  });
}

int nss_dist_aux_apply_f64(nss_dist_aux_t h, double scale, const double* b, double* y, nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(h && b && y && b != y, "dist_aux_apply: bad argument");
    dist_aux_apply(*h, scale, b, y, false, as_stream(stream), nullptr);
  });
}

int nss_dist_amg_create(nss_dist_t d, nss_csr_t a_loc, const nss_halo_t* halo_x, nss_csr_t r_loc, nss_csr_t p_loc,
                        const double* wdinv, nss_amg_t coarse, nss_dist_amg_t* out) {
  return guarded([&] {
    NSS_REQUIRE(d && a_loc && halo_x && r_loc && p_loc && wdinv && coarse && out, "dist_amg_create: NULL argument");
    NSS_REQUIRE(coarse->T == nullptr && !coarse->levels.empty(), "dist_amg_create: coarse must be a plain V-cycle handle");
    const int32_t n = a_loc->m, nc = coarse->levels[0].n;
    NSS_REQUIRE(a_loc->n >= n && r_loc->m == nc && r_loc->n == n && p_loc->m == n && p_loc->n == nc,
                "dist_amg_create: operator shapes do not chain");
    check_halo(halo_x, *a_loc, "dist_amg halo");
    nss_dist_amg_s* h = new nss_dist_amg_s;
    try {
      h->d = d;
      h->A = a_loc;
      h->R = r_loc;
      h->P = p_loc;
      h->wdinv = wdinv;
      h->coarse = coarse;
      h->halo = *halo_x;
      h->n = n;
      h->nc = nc;
      NSS_HIP(hipMalloc(&h->res, sizeof(double) * size_t(std::max(1, n))));
      NSS_HIP(hipMalloc(&h->rc, sizeof(double) * size_t(std::max(1, nc))));
      NSS_HIP(hipMalloc(&h->rc_local, sizeof(double) * size_t(std::max(1, nc))));
      NSS_HIP(hipMalloc(&h->ec, sizeof(double) * size_t(std::max(1, nc))));
    } catch (...) {
      nss_dist_amg_destroy(h);
      throw;
    }
    *out = h;
  });
}

int nss_dist_amg_destroy(nss_dist_amg_t h) {
  return guarded([&] {
    if (!h) return;
    (void)hipFree(h->res);
    (void)hipFree(h->rc);
    (void)hipFree(h->rc_local);
    (void)hipFree(h->ec);
    delete h;
  });
}

int nss_dist_amg_apply_f64(nss_dist_amg_t h, double scale, const double* b, double* y, nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(h && b && y && b != y, "dist_amg_apply: bad argument");
    dist_amg_apply(*h, scale, b, y, as_stream(stream), nullptr);
  });
}

int nss_dist_create(void* nccl_comm, int32_t nranks, int32_t rank, nss_dist_t* out) {
  return guarded([&] {
    NSS_REQUIRE(out != nullptr, "dist_create: out is NULL");
    NSS_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "dist_create: bad rank / size");
    nss_dist_s* d = new nss_dist_s;
    try {
      d->comm = nccl_comm;
      d->nranks = nranks;
      d->rank = rank;
      if (nccl_comm) {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
          d->lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
          if (!d->lib) d->lib = dlopen(name, RTLD_NOW);
          if (d->lib) break;
        }
        if (!d->lib) throw Error("dist_create: cannot open librccl");
        resolve(d->lib, "ncclAllReduce", d->AllReduce);
        resolve(d->lib, "ncclSend", d->Send);
        resolve(d->lib, "ncclRecv", d->Recv);
        resolve(d->lib, "ncclGroupStart", d->GroupStart);
        resolve(d->lib, "ncclGroupEnd", d->GroupEnd);
        resolve(d->lib, "ncclGetErrorString", d->GetErrorString);
      }
      NSS_HIP(hipStreamCreateWithFlags(&d->xstream, hipStreamNonBlocking));
      for (int i = 0; i < 3; ++i) {
        NSS_HIP(hipEventCreateWithFlags(&d->ev_ready[i], hipEventDisableTiming));
        NSS_HIP(hipEventCreateWithFlags(&d->ev_halo[i], hipEventDisableTiming));
      }
    } catch (...) {
      nss_dist_destroy(d);
      throw;
    }
    *out = d;
  });
}

int nss_dist_attach_p2p(nss_dist_t d, nss_p2p_t p) {
  return guarded([&] {
    NSS_REQUIRE(d != nullptr, "dist_attach_p2p: NULL dist handle");
    NSS_REQUIRE(!p || (p->nranks == d->nranks && p->rank == d->rank), "dist_attach_p2p: rank / size mismatch");
    d->p2p = p;
  });
}

int nss_dist_destroy(nss_dist_t d) {
  return guarded([&] {
    if (!d) return;
    for (int i = 0; i < 3; ++i) {
      if (d->ev_ready[i]) (void)hipEventDestroy(d->ev_ready[i]);
      if (d->ev_halo[i]) (void)hipEventDestroy(d->ev_halo[i]);
    }
    if (d->xstream) (void)hipStreamDestroy(d->xstream);
    for (hipEvent_t e : d->prof_ev) (void)hipEventDestroy(e);
    delete d;   // the library handle stays loaded (it is the process-wide librccl)
  });
}

int nss_dist_profile_begin(nss_dist_t d, int32_t max_iterations) {
  return guarded([&] {
    NSS_REQUIRE(d != nullptr && max_iterations >= 1 && max_iterations <= 4096, "dist_profile_begin: bad argument");
    for (hipEvent_t e : d->prof_ev) (void)hipEventDestroy(e);
    d->prof_ev.assign(size_t(max_iterations) * kProfMarks, nullptr);
    for (hipEvent_t& e : d->prof_ev) NSS_HIP(hipEventCreate(&e));
    d->prof_cap = max_iterations;
    d->prof_iters = 0;
  });
}

int nss_dist_profile_end(nss_dist_t d, double* h_segment_ms, int32_t* iterations) {
  return guarded([&] {
    NSS_REQUIRE(d != nullptr && h_segment_ms != nullptr, "dist_profile_end: NULL argument");
    const int n = d->prof_iters;
    for (int sgm = 0; sgm < kProfMarks - 1; ++sgm) h_segment_ms[sgm] = 0.0;
    if (n > 0) NSS_HIP(hipEventSynchronize(d->prof_ev[size_t(n) * kProfMarks - 1]));
    for (int it = 0; it < n; ++it)
      for (int sgm = 0; sgm < kProfMarks - 1; ++sgm) {
        float ms = 0.0f;
        NSS_HIP(hipEventElapsedTime(&ms, d->prof_ev[size_t(it) * kProfMarks + sgm], d->prof_ev[size_t(it) * kProfMarks + sgm + 1]));
        h_segment_ms[sgm] += double(ms) / n;
      }
    if (iterations) *iterations = n;
    for (hipEvent_t e : d->prof_ev) (void)hipEventDestroy(e);
    d->prof_ev.clear();
    d->prof_cap = d->prof_iters = 0;
  });
}

int nss_bpcg2_iterate_dist(const nss_bpcg2_t* s, nss_dist_t d, const nss_halo_t* halo_s1, const nss_halo_t* halo_t1,
                           const nss_halo_t* halo_t4, int32_t overlap, int32_t it_begin, int32_t it_end,
                           nss_stream_t stream) {
  return guarded([&] {
    bpcg2_check_state(s);
    NSS_REQUIRE(d != nullptr, "iterate_dist: NULL dist handle");
    NSS_REQUIRE(d->nranks == 1 || d->comm != nullptr || s->p2p != nullptr || d->p2p != nullptr, "iterate_dist: multi-rank run without a communicator");
    if (s->dist_compact) {
      // Compact plan: C1 (books of the previous iteration from the ALL-REDUCED <w, d>, rows of B^T, ghost copies of
      // s0) . preA . exchange of t1 . C23 (rows of A; owned and ghost rows of B on t1 - s0) . sum . all-reduce .
      // C4 . sum . all-reduce -- six launches and three collectives (eight-phase form: nine launches).
      check_halo(halo_t1, *s->A, "halo_t1");
      NSS_REQUIRE(halo_t1->ext == s->t1, "iterate_dist: the halo buffer is not the loop's SpMV operand t1");
      hipStream_t cs = as_stream(stream);
      for (int it = it_begin; it < it_end; ++it) {
        hipEvent_t* ev = d->prof_iters < d->prof_cap ? &d->prof_ev[size_t(d->prof_iters) * kProfMarks] : nullptr;
        auto mark = [&](int i) {
          if (ev) NSS_HIP(hipEventRecord(ev[i], cs));
        };
        mark(0);
        bpcg2_cphase(*s, NSS_BPCG2C_C1, it, cs);           // + preA
        mark(1);
        if (s->p2p) p2p_exchange(*s->p2p, *halo_t1, s->ctrl, cs);   // put + wait/copy through the landing zone
        else exchange(*d, *halo_t1, cs);
        mark(2);
        bpcg2_cphase(*s, NSS_BPCG2C_C23, it, cs);
        mark(3);
        bpcg2_cphase(*s, NSS_BPCG2C_SUMA, it, cs);          // mailbox transport: the all-reduce is part of this launch
        mark(4);
        if (!s->p2p) allreduce_slot(*s, *d, S_AS_SLOT, cs);
        mark(5);
        bpcg2_cphase(*s, NSS_BPCG2C_C4, it, cs);           // alpha inside
        bpcg2_cphase(*s, NSS_BPCG2C_SUMW, it, cs);
        mark(6);
        if (!s->p2p) allreduce_slot(*s, *d, S_WDN_SLOT, cs);   // read by C1 of the next iteration (or the poll)
        mark(7);
        mark(8);
        if (ev) ++d->prof_iters;
      }
      return;
    }
    check_halo(halo_s1, *s->BT, "halo_s1");
    check_halo(halo_t1, *s->A, "halo_t1");
    check_halo(halo_t4, *s->B, "halo_t4");
    NSS_REQUIRE(halo_s1->ext == s->s1 && halo_t1->ext == s->t1 && halo_t4->ext == s->t4,
                "iterate_dist: halo buffers are not the loop's SpMV operands");
    hipStream_t cs = as_stream(stream);
    const int ov = overlap;
    for (int it = it_begin; it < it_end; ++it) {
      hipEvent_t* ev = d->prof_iters < d->prof_cap ? &d->prof_ev[size_t(d->prof_iters) * kProfMarks] : nullptr;
      auto mark = [&](int i) {
        if (ev) NSS_HIP(hipEventRecord(ev[i], cs));
      };
      mark(0);
      if (s->ghost_p_mode) bpcg2_spmv_phase(*s, NSS_BPCG2_K1, it, cs, 0, -1);   // s1's ghosts are kept up to date locally
      else spmv_with_halo(*s, *d, *halo_s1, 0, NSS_BPCG2_K1, it, *s->BT, ov, cs);
      bpcg2_k1_finish(*s, cs);
      mark(1);
      spmv_with_halo(*s, *d, *halo_t1, 1, NSS_BPCG2_K2, it, *s->A, ov, cs, ev ? ev[2] : nullptr);
      mark(3);
      if (s->ghost_mode) bpcg2_spmv_phase(*s, NSS_BPCG2_K3, it, cs, 0, -1);   // t4's ghosts were computed in K2
      else spmv_with_halo(*s, *d, *halo_t4, 2, NSS_BPCG2_K3, it, *s->B, ov, cs);
      bpcg2_phase(*s, NSS_BPCG2_SUM1, it, cs);
      mark(4);
      allreduce_slot(*s, *d, S_AS_SLOT, cs);
      mark(5);
      bpcg2_phase(*s, NSS_BPCG2_K4, it, cs);       // alpha inside
      bpcg2_phase(*s, NSS_BPCG2_SUM2, it, cs);
      mark(6);
      allreduce_slot(*s, *d, S_WDN_SLOT, cs);
      mark(7);
      bpcg2_phase(*s, NSS_BPCG2_K5, it, cs);       // beta, history, stop test inside
      mark(8);
      if (ev) ++d->prof_iters;
    }
  });
}

}  // extern "C"
