// Fused, device-resident iteration of the recurrence-optimised Bramble-Pasciak CG
// (reference loop: solvers/bramblepasciak_new.py:200-249).
//
// One iteration = 3 CSR-stream SpMVs whose epilogues carry the vector recurrences and
// the dot partials, one fused element-wise update, one small s-update and two
// single-workgroup reductions that also advance the scalars (alpha, beta, stop test):
//
//   K1  rows of B^T :  [it>0] z0 -= a*t2 (:238, deferred), q = b*q + z0_old - a*t2 (:205),
//                      s0 = b*s0 + w0 (:240-241, velocity part, deferred)
//                      t0 = q + B^T s1 (:206-207);  Jacobi preA: t1 = k*dinv*t0 (:209)
//   [J] block-Jacobi preA: t1 = k * J t0 (:209)
//   K2  rows of A   :  t2 = A t1 (:210), t4 = t1 - s0 (:212), partial <s0, t2 - t0> (:218,222)
//   K3  rows of B   :  t3 = B t4 (:213), partial <s1, t3> (:219,222)
//   R1               :  as_s = sum of the partials (one workgroup, fixed order)
//   K4  element-wise:  alpha = wd / as_s (:226); u += a*s (:228), d -= a*v (:229), w0 -= a*t1,
//                      w1 -= a*minv*t3 (:232-233), partial <w, d> (:235)
//   R2               :  wdn = sum of the partials
//   K5  element-wise:  beta = wdn / wd (:236), s1 = b*s1 + w1 (:240-241, pressure part); one lane:
//                      hist[it] = sqrt|wd| (:243), stop test (:246)
//
// The single-GPU loop (nss_bpcg2_iterate) issues the same arithmetic in THREE dependent launches
// (+ the block-Jacobi apply) instead of eight -- the "compact plan", what makes the small configurations
// (1e5 .. 1e6 DoF: bound by kernel boundaries, not bandwidth) twice as fast:
//
//   C1  rows of B^T :  K1, preceded in every workgroup by the books of the previous iteration (what K5
//                      did: wdn = sum of K4's partials, beta, history entry, stop test) and multiplying
//                      with the operand beta*s1 + w1 formed on the fly (s1 is materialised by C23)
//   C23 rows of A and of B in ONE launch:  t2 = A t1 with <s0, t2 - t0>;  t3 = B (t1 - s0) with the
//                      operand formed on the fly (no t4), s1 = beta*s1 + w1 stored, <s1, t3>
//   C4  element-wise:  K4, preceded in every workgroup by as_s = sum of the partials of C23
// Every lane performs exactly the floating-point operations of the eight-kernel form and the sums use
// the same tree (fixed_sum_1024), so both forms -- and the row-partitioned loop, which keeps the
// eight-kernel form because its sums are all-reduced in between -- give identical bits.  Systems with
// more than kFoldMax partials per sum keep the two stand-alone sum kernels (C1 and C4 then read the
// scalar instead of summing).
//
// alpha, beta, wd and the `done` flag live in device memory: nothing is copied to the host
// inside the loop.  Once `done` is set every kernel returns immediately, so the state is
// frozen exactly at the reference's `break` and the host may poll every m iterations.
// R1/R2 only form local sums; alpha and beta are evaluated by the consuming kernels, so the
// row-partitioned multi-GPU loop simply all-reduces scal[as_s] / scal[wdn] in between
// (SURVEY.md section 8e).
#include "dist.h"

// Streaming (non-temporal) LOADS of vector operands that are not read again before they are rewritten: they do
// not displace the operand windows of the SpMV kernels from the 4-MiB XCD L2s.  Interleaved same-box A/B at 1e7
// DoF (profiles/r02_ab_streaming_loads.txt): packed block-Jacobi inverses +2.3 % iterations/s, C1's q / z0 / t2 /
// w0 / u0 +5.4 % (C23 gets 7 % faster: t1 and s0 are still in L2 when it stages them), C4's operands +3.6 %;
// MINRES' element-wise kernels +8 % (minres.hip), BPCG v1's +1.4 % (bpcg1.hip).  At 1e6 DoF, where the vectors
// fit the caches, the same loads cost 5 %: the choice is a template parameter (NT) made per launch from the
// vector length (stream_vector_loads, nss_common.h).

namespace nss {

enum { S_WD = 0, S_AS = 1, S_WDN = 2, S_ALPHA = 3, S_BETA = 4, S_ERR0 = 5, S_TOL = 6, S_REL = 7, S_WD_ODD = 8,
       S_AS_LOCAL = 9, S_WDN_LOCAL = 10 };
// Row-partitioned runs (nss_bpcg2_t.local_sums): the local sums go to S_AS_LOCAL / S_WDN_LOCAL and the
// all-reduce writes S_AS / S_WDN out of place.  Once the loop has stopped the sum kernels return early,
// the local slots keep their last value and every further all-reduce reproduces the same global sum:
// the poll-visible scalars stay frozen (an in-place all-reduce of a frozen slot would multiply it by
// the number of ranks each time).
// wd of iteration `it` lives in slot S_WD (it even) or S_WD_ODD (it odd): K5 of iteration it writes
// the slot of it+1 while its other lanes still read the slot of it.
__device__ __forceinline__ int wd_slot(int it) { return (it & 1) ? S_WD_ODD : S_WD; }
enum { C_DONE = 0, C_IT_FINAL = 1, C_LAST_IT = 2, C_BREAKDOWN = 3, C_PENDING = 4, C_COMPUTED = 5, C_CLOSED = 6 };
// compact plan: C_COMPUTED = number of iterations whose K4 has run, C_CLOSED = number of iterations whose
// books (history entry, beta, stop test) are done.  The books of iteration it are done by C1 of iteration
// it + 1 or, when the host looks first, by nss_bpcg2_poll (idempotent: same inputs, same values).
// C_PENDING: the velocity part of `u += alpha s` (:228) of iteration it is not done by K4 but by K1 of
// iteration it + 1, which reads s0 anyway (one pass over n_u less per iteration).  K4 leaves
// C_PENDING = it + 1; K1(it + 1) applies the update iff it finds its own iteration number there;
// nss_bpcg2_poll applies what is still pending (loop frozen by the stop test, or maxsteps reached) and
// clears the word, so the solution is complete whenever the host looks at it.

struct EpiK1 {
  const int32_t* __restrict__ ctrl;
  const double* __restrict__ scal;
  double* __restrict__ u0;
  double* __restrict__ q;
  double* __restrict__ z0;
  const double* __restrict__ t2;
  double* __restrict__ s0;
  const double* __restrict__ w0;
  double* __restrict__ t0;
  double* __restrict__ t1;
  const double* __restrict__ dinv;  // nullptr when preA is block-Jacobi
  double k;
  int it;
  __device__ bool skip() const { return ctrl[C_DONE] != 0; }
  __device__ void row(int r, double bts) const {
    double qv = q[r];
    if (it != 0) {
      const double alpha = scal[S_ALPHA], beta = scal[S_BETA];
      const double zo = z0[r], t2v = t2[r], so = s0[r];
      if (ctrl[C_PENDING] == it) NSS_ST(u0[r], fma(alpha, so, u0[r]));   // deferred u += alpha s of iteration it - 1
      qv = fma(-alpha, t2v, fma(beta, qv, zo));
      NSS_ST(z0[r], fma(-alpha, t2v, zo));
      NSS_ST(q[r], qv);
      NSS_ST3(s0[r], fma(beta, so, w0[r]));
    }
    const double t = qv + bts;
    NSS_ST3(t0[r], t);
    if (dinv) t1[r] = k * (dinv[r] * t);
  }
  __device__ void finish(int, double*) const {}
};

struct EpiK2 {
  const int32_t* __restrict__ ctrl;
  const double* __restrict__ t0;
  const double* __restrict__ t1;
  const double* __restrict__ s0;
  double* __restrict__ t2;
  double* __restrict__ t4;
  double* __restrict__ partials;
  int32_t ghost_n;                          // partitioned runs: t4 on B's ghost columns (see nss_bpcg2_t)
  const int32_t* __restrict__ ghost_map;
  const double* __restrict__ ghost_s0;
  double* __restrict__ ghost_t4;
  double acc = 0.0;
  __device__ bool skip() const { return ctrl[C_DONE] != 0; }
  struct Pre { double s0 = 0.0, t1 = 0.0, t0 = 0.0; };
  __device__ Pre fetch(int r) const { return Pre{s0[r], t1[r], t0[r]}; }
  __device__ void row(int r, double at1, const Pre& p) {
    NSS_ST(t2[r], at1);
    NSS_ST2(t4[r], p.t1 - p.s0);
    acc = fma(p.s0, at1 - p.t0, acc);
  }
  __device__ void finish(int b, double* lds) {
    const double s = block_sum(acc, lds);
    if (threadIdx.x == 0 && b >= 0) partials[b] = s;
    // ghost_n > 0 only in a launch that is ordered after the arrival of t1's ghosts
    const int stride = gridDim.x * kBlock;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < ghost_n; i += stride) ghost_t4[i] = t1[ghost_map[i]] - ghost_s0[i];
  }
};

// r = scale * x - A y  (multiplicative MypreA: residual between the two sweeps, :379)
struct EpiScaledResidual {
  const int32_t* __restrict__ ctrl;
  double scale;
  const double* __restrict__ x;
  double* __restrict__ r;
  __device__ bool skip() const { return ctrl[C_DONE] != 0; }
  struct Pre { double x = 0.0; };
  __device__ Pre fetch(int i) const { return Pre{x[i]}; }
  __device__ void row(int i, double ay, const Pre& p) const { r[i] = fma(scale, p.x, -ay); }
  __device__ void finish(int, double*) const {}
};

__global__ __launch_bounds__(kBlock) void bpcg2_zero_kernel(const int32_t* __restrict__ ctrl, int32_t n,
                                                             double* __restrict__ y) {
  if (ctrl[C_DONE] != 0) return;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) y[i] = 0.0;
}

// condensed form: f = t0 + H^T t0
struct EpiLift {
  const int32_t* __restrict__ ctrl;
  const double* __restrict__ t0;
  double* __restrict__ f;
  __device__ bool skip() const { return ctrl[C_DONE] != 0; }
  struct Pre { double t0 = 0.0; };
  __device__ Pre fetch(int r) const { return Pre{t0[r]}; }
  __device__ void row(int r, double ax, const Pre& p) const { f[r] = p.t0 + ax; }
  __device__ void finish(int, double*) const {}
};

// condensed form: y += H y in place.  Only rows that receive something are written: H maps
// coupling dofs (its columns) to interior dofs (its non-empty rows), the two sets are disjoint, so
// no lane writes an entry another lane gathers.
struct EpiExtendInPlace {
  const int32_t* __restrict__ ctrl;
  double* __restrict__ y;
  __device__ bool skip() const { return ctrl[C_DONE] != 0; }
  __device__ void row(int r, double ax) const {
    if (ax != 0.0) y[r] += ax;
  }
  __device__ void finish(int, double*) const {}
};

// y += A x with the solver's done flag
struct EpiAddGuarded {
  const int32_t* __restrict__ ctrl;
  double* __restrict__ y;
  __device__ bool skip() const { return ctrl[C_DONE] != 0; }
  struct Pre { double y = 0.0; };
  __device__ Pre fetch(int r) const { return Pre{y[r]}; }
  __device__ void row(int r, double ax, const Pre& p) const { y[r] = p.y + ax; }
  __device__ void finish(int, double*) const {}
};

// y = A x, frozen once the loop has stopped (ghost rows of B)
struct EpiGuardedStore {
  const int32_t* __restrict__ ctrl;
  double* __restrict__ y;
  __device__ bool skip() const { return ctrl[C_DONE] != 0; }
  __device__ void row(int r, double ax) const { y[r] = ax; }
  __device__ void finish(int, double*) const {}
};

struct EpiK3 {
  const int32_t* __restrict__ ctrl;
  const double* __restrict__ s1;
  double* __restrict__ t3;
  double* __restrict__ partials;
  double acc = 0.0;
  __device__ bool skip() const { return ctrl[C_DONE] != 0; }
  struct Pre { double s1 = 0.0; };
  __device__ Pre fetch(int r) const { return Pre{s1[r]}; }
  __device__ void row(int r, double bt4, const Pre& p) {
    NSS_ST2(t3[r], bt4);
    acc = fma(p.s1, bt4, acc);
  }
  __device__ void finish(int b, double* lds) {
    const double s = block_sum(acc, lds);
    if (threadIdx.x == 0 && b >= 0) partials[b] = s;
  }
};

// local sum of the partials of K2 and K3 (or of K4 when nb == 0) into scal[slot]: one workgroup,
// fixed_sum_1024 (nss_common.h) -- the tree the compact plan evaluates inside its kernels.
__global__ __launch_bounds__(kSumLanes) void bpcg2_sum_kernel(const int32_t* __restrict__ ctrl, int na,
                                                               const double* __restrict__ pa, int nb,
                                                               const double* __restrict__ pb,
                                                               double* __restrict__ scal, int slot) {
  __shared__ double lds[kRedDoubles];
  if (ctrl[C_DONE] != 0) return;
  const double t = fixed_sum_1024<kSumLanes>(pa, na, pb, nb, lds);
  if (threadIdx.x == 0) scal[slot] = t;
}

// The same sum followed, in the same launch, by the all-reduce over the ranks through the peer-mapped mailboxes
// (p2p.h): the local sum goes to scal[local_slot], the global one -- the nranks values added in rank order, the same
// bits on every rank -- to scal[slot].  A peer that does not arrive within the timeout stops the loop (breakdown code 3).
__global__ __launch_bounds__(kSumLanes) void bpcg2_sum_p2p_kernel(int32_t* __restrict__ ctrl, int na,
                                                                   const double* __restrict__ pa, int nb,
                                                                   const double* __restrict__ pb, double* __restrict__ scal,
                                                                   int slot, int local_slot, int it, P2pView v) {
  __shared__ double lds[kRedDoubles];
  static_assert(kRedDoubles >= kP2pMaxRanks, "reduction scratch too small for the mailbox values");
  if (ctrl[C_DONE] != 0) return;
  const double local = fixed_sum_1024<kSumLanes>(pa, na, pb, nb, lds);
  const double total = p2p_allreduce_sum(v, local, lds);
  if (threadIdx.x == 0) {
    scal[local_slot] = local;
    scal[slot] = total;
    if (*reinterpret_cast<volatile int32_t*>(v.error) != 0) {
      ctrl[C_BREAKDOWN] = 3;
      ctrl[C_IT_FINAL] = it;
      ctrl[C_PENDING] = 0;
      ctrl[C_DONE] = 1;
    }
  }
}

struct K4Args {
  int32_t* ctrl;
  double* scal;
  int32_t n_u, n_p, it;
  double *u0, *d0, *w0, *u1, *d1, *w1;
  const double *s0, *t0, *t1, *t2, *s1, *t3, *minv;
  double* partials;
  int32_t ghost_n;
  const int32_t* ghost_map;
  double* ghost_w0;
  int32_t ghost_p_n;
  const double* ghost_t3;
  const double* ghost_minv;
  double* ghost_w1;
  int32_t gu, gp;                 // workgroups of the velocity / pressure part (one partial each)
  int32_t fold, na, nb;           // fold != 0: as_s = fixed sum of pa[0..na) and pb[0..nb) in every workgroup
  const double *pa, *pb;
};

// K4 is a one-shot launch: one double2 per lane, no grid-stride loop -- the form that reaches the highest
// HBM rate for an element-wise kernel on this chip (profiles/r02_triad_variants.txt: 6.2 TB/s against
// 5.5 TB/s for <= 2048 striding workgroups).
constexpr int kK4PerBlock = 2 * kBlock;

template <bool VEC2, bool FOLD, bool NT>
__global__ __launch_bounds__(kBlock) void bpcg2_k4_kernel(K4Args a) {
  __shared__ double lds[kRedDoubles];
  if (a.ctrl[C_DONE] != 0) return;
  // the lane's operands are requested before the sum of the partials: both latencies overlap
  const int wg0 = blockIdx.x;
  const int e0 = ((wg0 < a.gu ? wg0 : wg0 - a.gu) * kBlock + int(threadIdx.x)) * 2;
  double2 q0{}, q1{}, q2{}, q3{}, q4{}, q5{};
  const bool vec_u = VEC2 && wg0 < a.gu && e0 + 1 < a.n_u;
  const bool vec_p = VEC2 && wg0 >= a.gu && wg0 < a.gu + a.gp && e0 + 1 < a.n_p;
  // NT: streaming loads for everything that is dead or rewritten after this kernel (t0, t1, t3, d, u1) or next
  // read by a streaming load itself (t2, w0); s1 and w1 stay cached: the rows of B^T gather them next
  if (vec_u) {
    q0 = ld2s<NT>(a.t0 + e0);
    q1 = ld2s<NT>(a.t1 + e0);
    q2 = ld2s<NT>(a.t2 + e0);
    q3 = ld2s<NT>(a.d0 + e0);
    q4 = ld2s<NT>(a.w0 + e0);
  } else if (vec_p) {
    q0 = ld2(a.s1 + e0);
    q1 = ld2s<NT>(a.t3 + e0);
    q2 = ld2s<NT>(a.minv + e0);
    q3 = ld2s<NT>(a.u1 + e0);
    q4 = ld2s<NT>(a.d1 + e0);
    q5 = ld2(a.w1 + e0);
  }
  // alpha = wd / <s, K^ s> (:226), evaluated by every lane from the (all-)reduced sum.
  // <s, K^ s> == 0: the reference raises ZeroDivisionError in Python; freeze the state and report
  // it (ctrl[3]) so that the host can raise the same error.
  double as_s;
  if constexpr (FOLD) as_s = fixed_sum_1024(a.pa, a.na, a.pb, a.nb, lds);
  else as_s = a.scal[S_AS];
  if (as_s == 0.0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      a.ctrl[C_BREAKDOWN] = 1;
      a.ctrl[C_IT_FINAL] = a.it;
      a.ctrl[C_DONE] = 1;
      a.ctrl[C_PENDING] = 0;
    }
    return;
  }
  const double alpha = a.scal[wd_slot(a.it)] / as_s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.scal[S_ALPHA] = alpha;          // K1 of the next iteration
    a.ctrl[C_PENDING] = a.it + 1;     // ... which also applies u0 += alpha s0
    a.ctrl[C_COMPUTED] = a.it + 1;
    if (FOLD) a.scal[S_AS] = as_s;    // for the host's eyes only
  }
  const int wg = blockIdx.x;
  double acc = 0.0;
  if (wg < a.gu) {
    const int i0 = (wg * kBlock + int(threadIdx.x)) * 2;
    if (vec_u) {
      const double2 t0v = q0, t1v = q1, t2v = q2, d = q3, w = q4;
      double2 dn, wn;
      dn.x = fma(-alpha, t2v.x - t0v.x, d.x);
      dn.y = fma(-alpha, t2v.y - t0v.y, d.y);
      wn.x = fma(-alpha, t1v.x, w.x);
      wn.y = fma(-alpha, t1v.y, w.y);
      store2_nt(a.d0 + i0, dn);
      store2_nt(a.w0 + i0, wn);
      acc = fma(wn.x, dn.x, acc);
      acc = fma(wn.y, dn.y, acc);
    } else {
      for (int i = i0; i < i0 + 2 && i < a.n_u; ++i) {
        const double dn = fma(-alpha, a.t2[i] - a.t0[i], a.d0[i]);
        const double wn = fma(-alpha, a.t1[i], a.w0[i]);
        NSS_ST(a.d0[i], dn);
        NSS_ST(a.w0[i], wn);
        acc = fma(wn, dn, acc);
      }
    }
  } else if (wg < a.gu + a.gp) {
    const int i0 = ((wg - a.gu) * kBlock + int(threadIdx.x)) * 2;
    if (vec_p) {
      const double2 sv = q0, t3v = q1, mv = q2, d = q4, w = q5;
      double2 u = q3;
      u.x = fma(alpha, sv.x, u.x);
      u.y = fma(alpha, sv.y, u.y);
      double2 dn, wn;
      dn.x = fma(-alpha, t3v.x, d.x);
      dn.y = fma(-alpha, t3v.y, d.y);
      wn.x = fma(-alpha, mv.x * t3v.x, w.x);
      wn.y = fma(-alpha, mv.y * t3v.y, w.y);
      *reinterpret_cast<double2*>(a.u1 + i0) = u;
      *reinterpret_cast<double2*>(a.d1 + i0) = dn;
      *reinterpret_cast<double2*>(a.w1 + i0) = wn;
      acc = fma(wn.x, dn.x, acc);
      acc = fma(wn.y, dn.y, acc);
    } else {
      for (int i = i0; i < i0 + 2 && i < a.n_p; ++i) {
        const double sv = a.s1[i], t3v = a.t3[i];
        a.u1[i] = fma(alpha, sv, a.u1[i]);
        const double dn = fma(-alpha, t3v, a.d1[i]);
        const double wn = fma(-alpha, a.minv[i] * t3v, a.w1[i]);
        a.d1[i] = dn;
        a.w1[i] = wn;
        acc = fma(wn, dn, acc);
      }
    }
  } else {                                       // row-partitioned runs: ghost copies, same recurrences
    const int first = wg - a.gu - a.gp, stride = (int(gridDim.x) - a.gu - a.gp) * kBlock;
    for (int i = first * kBlock + int(threadIdx.x); i < a.ghost_n; i += stride)
      a.ghost_w0[i] = fma(-alpha, a.t1[a.ghost_map ? a.ghost_map[i] : a.n_u + i], a.ghost_w0[i]);
    for (int i = first * kBlock + int(threadIdx.x); i < a.ghost_p_n; i += stride)
      a.ghost_w1[i] = fma(-alpha, a.ghost_minv[i] * a.ghost_t3[i], a.ghost_w1[i]);
    return;
  }
  const double s = block_sum(acc, lds);
  if (threadIdx.x == 0) a.partials[wg] = s;
}

// K5: beta = wdn / wd (:236) by every lane; lane 0 of workgroup 0 also keeps the books: history
// entry (:243), stop test (:246), wd of the next iteration, beta for K1.  The lanes that see the
// freshly set `done` flag may skip the s1 update of this last iteration: s is not returned.
__global__ __launch_bounds__(kBlock) void bpcg2_k5_kernel(int32_t* __restrict__ ctrl, double* __restrict__ scal,
                                                           double* __restrict__ hist, int32_t it, int32_t n_p,
                                                           double* __restrict__ s1, const double* __restrict__ w1,
                                                           int32_t ghost_n, double* __restrict__ ghost_s0,
                                                           const double* __restrict__ ghost_w0, int32_t ghost_p_n,
                                                           double* __restrict__ ghost_s1,
                                                           const double* __restrict__ ghost_w1) {
  if (ctrl[C_DONE] != 0) return;
  const double wd = scal[wd_slot(it)], wdn = scal[S_WDN];
  const double beta = wdn / wd;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    scal[S_BETA] = beta;
    scal[wd_slot(it + 1)] = wdn;
    const double err = sqrt(fabs(wd));
    hist[it] = err;
    ctrl[C_LAST_IT] = it;
    ctrl[C_CLOSED] = it + 1;
    const double bound = scal[S_TOL] * (scal[S_REL] != 0.0 ? scal[S_ERR0] : 1.0);
    if (err < bound) {
      ctrl[C_IT_FINAL] = it;
      ctrl[C_DONE] = 1;
    }
  }
  const int stride = gridDim.x * kBlock;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n_p; i += stride) s1[i] = fma(beta, s1[i], w1[i]);
  // ghost copies of s0: what K1 of the next iteration does to the owned entries with this beta
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < ghost_n; i += stride) ghost_s0[i] = fma(beta, ghost_s0[i], ghost_w0[i]);
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < ghost_p_n; i += stride) ghost_s1[i] = fma(beta, ghost_s1[i], ghost_w1[i]);
}

// what K1 of the next iteration would have done, for a loop that ends here
__global__ __launch_bounds__(kBlock) void bpcg2_flush_kernel(const int32_t* __restrict__ ctrl,
                                                              const double* __restrict__ scal, int32_t n_u,
                                                              const double* __restrict__ s0, double* __restrict__ u0) {
  if (ctrl[C_PENDING] == 0) return;
  const double alpha = scal[S_ALPHA];
  const int stride = gridDim.x * kBlock;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n_u; i += stride) u0[i] = fma(alpha, s0[i], u0[i]);
}

__global__ void bpcg2_flush_done_kernel(int32_t* __restrict__ ctrl) { ctrl[C_PENDING] = 0; }

// =====================================================================================================
// compact plan (single GPU): C1, C23, C4 -- see the file header
// =====================================================================================================
struct CloseArgs {
  int32_t* ctrl;
  double* scal;
  double* hist;
  const double* partials_c;
  int32_t nc;
  int32_t fold;
};

// The books of iteration it - 1, evaluated by every workgroup of C1(it) from the same inputs (what K5
// did with one lane): wdn, beta = wdn / wd (:236), history entry (:243), stop test (:246), wd of the next
// iteration.  Workgroup 0 records them.  Returns false when the loop stops at iteration it - 1.
template <bool FOLD>
__device__ __forceinline__ bool close_iteration(const CloseArgs& a, int it, double* lds, double* beta_out) {
  const int prev = it - 1;
  double wdn;
  if constexpr (FOLD) wdn = fixed_sum_1024(a.partials_c, a.nc, a.partials_c, 0, lds);
  else wdn = a.scal[S_WDN];
  const double wd = a.scal[wd_slot(prev)];
  const double beta = wdn / wd;
  const double err = sqrt(fabs(wd));
  const double bound = a.scal[S_TOL] * (a.scal[S_REL] != 0.0 ? a.scal[S_ERR0] : 1.0);
  const bool stop = err < bound;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.scal[S_BETA] = beta;
    a.scal[wd_slot(it)] = wdn;
    if (FOLD) a.scal[S_WDN] = wdn;
    a.hist[prev] = err;
    a.ctrl[C_LAST_IT] = prev;
    a.ctrl[C_CLOSED] = it;
    if (stop) {
      a.ctrl[C_IT_FINAL] = prev;
      a.ctrl[C_DONE] = 1;
    }
  }
  *beta_out = beta;
  return !stop;
}

// the host looks (nss_bpcg2_poll): close the last computed iteration if C1 of its successor has not
__global__ __launch_bounds__(kBlock) void bpcg2_close_kernel(CloseArgs a) {
  __shared__ double lds[kRedDoubles];
  if (a.ctrl[C_DONE] != 0) return;
  const int computed = a.ctrl[C_COMPUTED];
  if (a.ctrl[C_CLOSED] >= computed) return;
  double beta;
  if (a.fold) close_iteration<true>(a, computed, lds, &beta);
  else close_iteration<false>(a, computed, lds, &beta);
}

// C1: K1 with the books of the previous iteration in front and the operand beta * s1 + w1 on the fly.
// FOLD (short sums) is a template parameter: the registers of the in-kernel sum must not cost the
// bandwidth-bound large systems their occupancy.
template <bool FOLD, bool NT = false>
struct EpiK1c {
  CloseArgs cl;
  double* __restrict__ u0;
  double* __restrict__ q;
  double* __restrict__ z0;
  const double* __restrict__ t2;
  double* __restrict__ s0;
  const double* __restrict__ w0;
  double* __restrict__ t0;
  double* __restrict__ t1;
  const double* __restrict__ dinv;  // nullptr when preA is not a plain point Jacobi
  double k;
  int it;
  const double* __restrict__ s1;
  const double* __restrict__ w1;
  // row-partitioned runs on the compact plan: the ghost copies of s0 follow s0 = beta s0 + w0 here (K5's job in
  // the eight-phase form); 0 entries on one GPU
  int32_t ghost_n = 0;
  double* __restrict__ ghost_s0 = nullptr;
  const double* __restrict__ ghost_w0 = nullptr;
  // block Jacobi applied here (B^T planned around its blocks, nss_csr_plan_for_blocks): t1 = k J t0 from the LDS copy
  // of this row block's t0, one lane per Jacobi block, the arithmetic of bjac_apply_sym_kernel (row i: the chain
  // fma(M_ij, x_j, .) over j ascending) -- no launch of its own, t0 is not read back
  const int32_t* __restrict__ jb_first = nullptr;    // nullptr: not fused
  const int32_t* __restrict__ jb_order = nullptr;
  const int32_t* __restrict__ jb_run = nullptr;
  const double* __restrict__ jb_packed = nullptr;
  int32_t jb_count = 0, jb_bs = 0;
  double alpha = 0.0, beta = 0.0;
  bool pending = false;
  __device__ bool skip() const { return cl.ctrl[C_DONE] != 0; }
  __device__ bool prologue(double* lds) {
    if (it == 0) return true;
    alpha = cl.scal[S_ALPHA];
    pending = cl.ctrl[C_PENDING] == it;
    return close_iteration<FOLD>(cl, it, lds, &beta);
  }
  struct X {
    const double* __restrict__ s1;
    const double* __restrict__ w1;
    double beta;
    bool first;
    static constexpr bool kStageable = false;   // an expression of two vectors: gathered, or pair-staged
    static constexpr bool kStageablePair = true;
    __device__ const double* ptr() const { return s1; }
    __device__ const double* ptr2() const { return w1; }
    __device__ double pair(double s, double w) const { return first ? s : fma(beta, s, w); }
    __device__ double operator()(int c) const { return first ? s1[c] : fma(beta, s1[c], w1[c]); }   // :240-241
  };
  __device__ X xop(const double*) const { return X{s1, w1, beta, it == 0}; }
  // the row's six read-only operands are requested before the matrix stream (and before alpha / beta
  // are known: the prologue runs later)
  // requesting the row's six operands before the matrix stream costs 12 registers (occupancy 8 -> 6 waves per
  // SIMD) and measured 0.7 % slower at 1e7 DoF, no different at 1e5 (profiles/r02_ab_k1_prefetch.txt): off
#ifndef NSS_K1C_PREFETCH
#define NSS_K1C_PREFETCH 0
#endif

#if NSS_K1C_PREFETCH
  struct Pre { double q = 0.0, z0 = 0.0, t2 = 0.0, s0 = 0.0, w0 = 0.0, u0 = 0.0; };
  __device__ Pre fetch(int r) const {
    if (it == 0) return Pre{q[r], 0.0, 0.0, 0.0, 0.0, 0.0};
    return Pre{q[r], z0[r], t2[r], s0[r], w0[r], u0[r]};
  }
  __device__ void row(int r, double bts, const Pre& p) const {
#else
  struct PreLate { double q, z0, t2, s0, w0, u0; };
  __device__ void row(int r, double bts) const {
    // NT: q, z0, t2, w0, u0 are read here and nowhere else before they are rewritten -- streaming loads keep
    // them out of the XCD L2s (s0 stays: the rows of A and B read it next)
    const PreLate p{ld1s<NT>(q + r), it ? ld1s<NT>(z0 + r) : 0.0, it ? ld1s<NT>(t2 + r) : 0.0, it ? s0[r] : 0.0,
                    it ? ld1s<NT>(w0 + r) : 0.0, (it && pending) ? ld1s<NT>(u0 + r) : 0.0};
#endif
    double qv = p.q;
    if (it != 0) {
      const double zo = p.z0, t2v = p.t2, so = p.s0;
      if (pending) NSS_ST(u0[r], fma(alpha, so, p.u0));    // deferred u += alpha s of iteration it - 1
      qv = fma(-alpha, t2v, fma(beta, qv, zo));
      NSS_ST(z0[r], fma(-alpha, t2v, zo));
      NSS_ST(q[r], qv);
      NSS_ST3(s0[r], fma(beta, so, p.w0));
    }
    const double t = qv + bts;
    NSS_ST3(t0[r], t);
    if (dinv) t1[r] = k * (dinv[r] * t);
    if (jb_first) {
      extern __shared__ double k1_t0[];
      k1_t0[r & (kBlockRows - 1)] = t;                   // (a row block holds at most kBlockRows consecutive rows)
    }
  }
  __device__ void finish(int b, double*) const {
    if (jb_first && b >= 0) {                            // (uniform over the workgroup)
      extern __shared__ double k1_t0[];
      __syncthreads();
      const int j1 = jb_first[b + 1];
      for (int pos = jb_first[b] + int(threadIdx.x); pos < j1; pos += kBlock) {
        const int jb = jb_order[pos];
        const int32_t w = jb_run[jb], first = w >> 5, len = w & 31;
        for (int i = 0; i < len; ++i) {
          double s = 0.0;
          for (int j = 0; j < len; ++j) {
            const int lo = i < j ? i : j, hi = i < j ? j : i;
            const int tri = lo * jb_bs - (lo * (lo - 1)) / 2 + (hi - lo);      // upper triangle, row-major
            s = fma(jb_packed[size_t(tri) * jb_count + jb], k1_t0[(first + j) & (kBlockRows - 1)], s);
          }
          t1[first + i] = k * s;
        }
      }
    }
    if (it == 0 || ghost_n == 0) return;
    const int stride = gridDim.x * kBlock;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < ghost_n; i += stride) ghost_s0[i] = fma(beta, ghost_s0[i], ghost_w0[i]);
  }
};

// C23, rows of A: K2 without t4
struct EpiK2c {
  const int32_t* __restrict__ ctrl;
  const double* __restrict__ t0;
  const double* __restrict__ s0;
  double* __restrict__ t2;
  double* __restrict__ partials;
  double acc = 0.0;
  __device__ bool skip() const { return ctrl[C_DONE] != 0; }
  struct Pre { double s0 = 0.0, t0 = 0.0; };
  __device__ Pre fetch(int r) const { return Pre{s0[r], t0[r]}; }   // (t0 as a streaming load: no difference measured)
  __device__ void row(int r, double at1, const Pre& p) {
    NSS_ST(t2[r], at1);
    acc = fma(p.s0, at1 - p.t0, acc);
  }
  __device__ void finish(int b, double* lds) {
    const double s = block_sum(acc, lds);
    if (threadIdx.x == 0 && b >= 0) partials[b] = s;
  }
};

// C23, rows of B: K3 multiplying with t4 = t1 - s0 formed on the fly (:212-213) and storing
// s1 = beta * s1 + w1 (:240-241, pressure part -- K5's job)
struct EpiK3c {
  const int32_t* __restrict__ ctrl;
  const double* __restrict__ scal;
  const double* __restrict__ t1;
  const double* __restrict__ s0;
  double* __restrict__ s1;
  const double* __restrict__ w1;
  double* __restrict__ t3;
  double* __restrict__ partials;
  int it;
  int n_own;          // rows from n_own on are the ghost pressure rows of a row-partitioned run: same recurrences,
                      // nothing added to the inner product (one GPU: n_own = rows of B)
  double acc = 0.0;
  double beta = 0.0;
  __device__ bool skip() const { return ctrl[C_DONE] != 0; }
  __device__ bool prologue(double*) {
    if (it != 0) beta = scal[S_BETA];
    return true;
  }
  struct X {
    const double* __restrict__ t1;
    const double* __restrict__ s0;
    static constexpr bool kStageable = false;   // an expression of two vectors: gathered, or pair-staged
    static constexpr bool kStageablePair = true;
    __device__ const double* ptr() const { return t1; }
    __device__ const double* ptr2() const { return s0; }
    __device__ double pair(double a, double b) const { return a - b; }
    __device__ double operator()(int c) const { return t1[c] - s0[c]; }
  };
  __device__ X xop(const double*) const { return X{t1, s0}; }
  struct Pre { double s1 = 0.0, w1 = 0.0; };
  __device__ Pre fetch(int r) const { return Pre{s1[r], it != 0 ? w1[r] : 0.0}; }
  __device__ void row(int r, double bt4, const Pre& p) {
    double sv = p.s1;
    if (it != 0) {
      sv = fma(beta, p.s1, p.w1);
      s1[r] = sv;
    }
    NSS_ST2(t3[r], bt4);
    if (r < n_own) acc = fma(sv, bt4, acc);
  }
  __device__ void finish(int b, double* lds) {
    const double s = block_sum(acc, lds);
    if (threadIdx.x == 0 && b >= 0) partials[b] = s;
  }
};

static int k4_gu(const nss_bpcg2_t& s) { return (s.n_u + kK4PerBlock - 1) / kK4PerBlock; }
static int k4_gp(const nss_bpcg2_t& s) { return (s.n_p + kK4PerBlock - 1) / kK4PerBlock; }
static int k4_partials(const nss_bpcg2_t& s) { return k4_gu(s) + k4_gp(s); }
static int k4_ghost_blocks(const nss_bpcg2_t& s) {
  const int64_t g = std::max<int64_t>(s.ghost_mode ? s.ghost_n : 0, s.ghost_p_mode ? s.ghost_p_n : 0);
  return g > 0 ? stream_grid(g, kBlock) : 0;
}

// Sums with at most this many terms are evaluated inside the consuming kernel by every workgroup
// (compact plan); longer ones by the stand-alone sum kernel.  nss_bpcg2_fold_mode() overrides (tests,
// measurements): -1 automatic, 0 never, 1 always.
// Measured (profiles/r02_ab_fold_threshold.txt, folded vs stand-alone at 1.5e4 ... 1e6 DoF): folding wins by
// 3-30 % up to 2.5e5 DoF (~800 + ~500 partials) and LOSES 11-13 % from 4.4e5 DoF (~1400 + ~850) on -- every one of
// the ~1000-2000 workgroups of the consumer then re-reads all partials from L2, as many bytes as the kernel's own
// HBM traffic.  (Round-2 first value: 4096.)
constexpr int kFoldMax = 1024;
static int g_fold_mode = -1;
static bool fold_sums(const nss_bpcg2_t& s) {
  const int forced = g_fold_mode;
  if (s.local_sums) return false;        // row-partitioned: the sums are all-reduced between the kernels
  if (forced >= 0) return forced != 0;
  return s.A->nblk + s.B->nblk <= kFoldMax && k4_partials(s) <= kFoldMax;
}

static void launch_k4(const nss_bpcg2_t& s, int it, bool fold, hipStream_t st) {
  K4Args a{s.ctrl, s.scal, s.n_u, s.n_p, it, s.u0, s.d0, s.w0, s.u1, s.d1, s.w1,
           s.s0, s.t0, s.t1, s.t2, s.s1, s.t3, s.minv, s.partials_c, s.ghost_mode ? s.ghost_n : 0, s.ghost_map,
           s.ghost_w0, s.ghost_p_mode ? s.ghost_p_n : 0, s.ghost_t3, s.ghost_minv, s.ghost_w1,
           k4_gu(s), k4_gp(s), fold ? 1 : 0, s.A->nblk, s.B->nblk, s.partials_a, s.partials_b};
  const int grid = a.gu + a.gp + k4_ghost_blocks(s);
  bool vec = true;
  for (const void* p : {(const void*)s.u1, (const void*)s.d0, (const void*)s.d1, (const void*)s.w0, (const void*)s.w1,
                        (const void*)s.s1, (const void*)s.t0, (const void*)s.t1, (const void*)s.t2, (const void*)s.t3,
                        (const void*)s.minv})
    vec = vec && aligned16(p);
  const bool nt = vec && stream_vector_loads(s.n_u);
  if (vec && fold && nt) hipLaunchKernelGGL((bpcg2_k4_kernel<true, true, true>), dim3(grid), dim3(kBlock), 0, st, a);
  else if (vec && fold) hipLaunchKernelGGL((bpcg2_k4_kernel<true, true, false>), dim3(grid), dim3(kBlock), 0, st, a);
  else if (vec && nt) hipLaunchKernelGGL((bpcg2_k4_kernel<true, false, true>), dim3(grid), dim3(kBlock), 0, st, a);
  else if (vec) hipLaunchKernelGGL((bpcg2_k4_kernel<true, false, false>), dim3(grid), dim3(kBlock), 0, st, a);
  else if (fold) hipLaunchKernelGGL((bpcg2_k4_kernel<false, true, false>), dim3(grid), dim3(kBlock), 0, st, a);
  else hipLaunchKernelGGL((bpcg2_k4_kernel<false, false, false>), dim3(grid), dim3(kBlock), 0, st, a);
  NSS_CHECK_LAUNCH();
}

void bpcg2_check_state(const nss_bpcg2_t* s) {
  NSS_REQUIRE(s != nullptr, "bpcg2: NULL state");
  NSS_REQUIRE(s->A && s->B && s->BT, "bpcg2: NULL matrix handle");
  NSS_REQUIRE(s->A->m == s->n_u && s->BT->m == s->n_u, "bpcg2: matrix row counts do not match n_u");
  if (s->dist_compact) {
    NSS_REQUIRE(s->local_sums && s->ghost_mode && s->ghost_p_mode && s->ghost_n >= 0 && s->ghost_p_n >= 0,
                "bpcg2: the compact partitioned plan keeps all ghosts by recurrence (ghost_mode, ghost_p_mode, local_sums)");
    NSS_REQUIRE(s->B->m == s->n_p + s->ghost_p_n, "bpcg2: compact partitioned plan: B must hold the owned and the ghost pressure rows");
    NSS_REQUIRE(s->A->n == s->n_u + s->ghost_n && s->B->n == s->A->n && s->BT->n == s->n_p + s->ghost_p_n,
                "bpcg2: compact partitioned plan: operand layouts [owned | ghosts] of A (also B's) and B^T do not match the ghost counts");
    NSS_REQUIRE(s->ghost_map == nullptr && s->ghost_s0 == s->s0 + s->n_u && s->ghost_w0 == s->w0 + s->n_u &&
                    s->ghost_w1 == s->w1 + s->n_p && s->ghost_t3 == s->t3 + s->n_p && (s->ghost_p_n == 0 || s->ghost_minv),
                "bpcg2: compact partitioned plan: ghost copies must sit behind the owned entries of s0, w0, w1, t3");
    NSS_REQUIRE(!s->cond_HT, "bpcg2: compact partitioned plan takes no condensed form");
    NSS_REQUIRE(!s->p2p || s->p2p->connected, "bpcg2: the mailbox transport is not connected");
  } else if (s->p2p) {
    throw Error("bpcg2: the mailbox transport (p2p) serves the compact partitioned plan only");
  } else {
    NSS_REQUIRE(s->B->m == s->n_p, "bpcg2: matrix row counts do not match n_p");
  }
  NSS_REQUIRE(!(s->pre_diag && s->pre_bjac), "bpcg2: pre_diag and pre_bjac are exclusive");
  NSS_REQUIRE(s->pre_diag || s->pre_bjac || s->pre_amg || s->pre_dist_amg || s->pre_dist_aux, "bpcg2: no preconditioner for the velocity block");
  NSS_REQUIRE(!s->pre_dist_aux || (!s->pre_amg && !s->pre_dist_amg && !s->cond_HT && s->pre_dist_aux->n_u == s->n_u),
              "bpcg2: the row-partitioned auxiliary-space term replaces pre_amg / pre_dist_amg, takes no condensed form, and must match n_u");
  NSS_REQUIRE(!s->pre_dist_aux || !(s->pre_bjac && s->pre_bjac->gs_mat) ||
                  (s->pre_dist_aux->has_halo_y && s->pre_dist_aux->halo_y.ext == s->t1),
              "bpcg2: the multiplicative partitioned MypreA needs the halo of t1 (nss_dist_aux_create: halo_y)");
  NSS_REQUIRE(!s->pre_dist_amg || (!s->pre_amg && !s->cond_HT && s->pre_dist_amg->n == s->n_u),
              "bpcg2: the row-partitioned AMG replaces pre_amg, takes no condensed form, and must match n_u");
  NSS_REQUIRE(!s->pre_amg || s->pre_amg->levels[0].n == s->n_u, "bpcg2: AMG size mismatch");
  // AMG (or auxiliary-space) term + a block-Jacobi handle in Gauss-Seidel mode = the MULTIPLICATIVE MypreA
  // (GS=True, :376-381): sweep, residual, correction, back sweep.  Not with the condensed form.
  NSS_REQUIRE(!(s->pre_amg && s->pre_bjac && s->pre_bjac->gs_mat && (s->cond_HT || s->pre_diag)),
              "bpcg2: the multiplicative preconditioner (Gauss-Seidel sweeps around an AMG term) takes neither a "
              "condensed form nor a point-Jacobi part");
  NSS_REQUIRE(!s->pre_bjac || s->pre_bjac->n == s->n_u, "bpcg2: block-Jacobi size mismatch");
  NSS_REQUIRE(s->minv && s->scal && s->ctrl && s->hist && s->partials_a && s->partials_b && s->partials_c,
              "bpcg2: NULL work buffer");
  NSS_REQUIRE(!s->ghost_mode || s->ghost_n == 0 || ((s->ghost_map || s->dist_compact) && s->ghost_s0 && s->ghost_w0),
              "bpcg2: ghost mode without ghost arrays");
  NSS_REQUIRE(s->ghost_n >= 0 && s->ghost_p_n >= 0, "bpcg2: negative ghost count");
  NSS_REQUIRE(!s->ghost_p_mode || s->ghost_p_n == 0 ||
                  (s->ghost_mode && s->ghost_t3 && s->ghost_w1 && s->ghost_minv &&
                   (s->dist_compact || (s->ghost_b && s->ghost_b->m == s->ghost_p_n))),
              "bpcg2: pressure ghost mode without its arrays (it also needs the velocity ghost mode)");
  const bool cond = s->cond_HT || s->cond_H || s->cond_inner || s->cond_f;
  if (cond) {
    NSS_REQUIRE(s->cond_HT && s->cond_H && s->cond_inner && s->cond_f, "bpcg2: condensed form needs H^T, H, A_ii^-1 and a work vector");
    for (const nss_csr_s* m : {s->cond_HT, s->cond_H, s->cond_inner})
      NSS_REQUIRE(m->m == s->n_u && m->n == s->n_u, "bpcg2: condensed operators must be n_u x n_u");
  }
  NSS_REQUIRE(s->u0 && s->u1 && s->d0 && s->d1 && s->w0 && s->w1 && s->s0 && s->s1 && s->z0 && s->q && s->t0 &&
                  s->t1 && s->t2 && s->t3 && s->t4,
              "bpcg2: NULL vector");
}

// SpMV phases over the row blocks [b0, b1) of their matrix (b1 < 0: all).  The block-Jacobi
// apply that completes K1 is a separate step (`bpcg2_k1_finish`) because it needs all of t0.
void bpcg2_spmv_phase(const nss_bpcg2_t& s, int which, int it, hipStream_t st, int b0, int b1, bool ghost_tail) {
  if (s.dist_compact) throw Error("bpcg2: a state laid out for the compact partitioned plan takes the compact phases only");
  switch (which) {
    case NSS_BPCG2_K1: {
      // the point-Jacobi apply rides in the epilogue unless an AMG term comes first
      EpiK1 e{s.ctrl, s.scal, s.u0, s.q, s.z0, s.t2, s.s0, s.w0, s.t0, s.t1,
              (s.pre_amg || s.pre_dist_amg || s.pre_dist_aux || s.cond_HT) ? nullptr : s.pre_diag, s.k, it};
      launch_csr_stream(*s.BT, s.s1, e, st, b0, b1);
      break;
    }
    case NSS_BPCG2_K2: {
      EpiK2 e{s.ctrl, s.t0, s.t1, s.s0, s.t2, s.t4, s.partials_a, (s.ghost_mode && ghost_tail) ? s.ghost_n : 0, s.ghost_map,
              s.ghost_s0, s.t4 + s.n_u};
      launch_csr_stream(*s.A, s.t1, e, st, b0, b1);
      break;
    }
    case NSS_BPCG2_K3: {
      EpiK3 e{s.ctrl, s.s1, s.t3, s.partials_b};
      launch_csr_stream(*s.B, s.t4, e, st, b0, b1);
      break;
    }
    default:
      throw Error("bpcg2: not an SpMV phase");
  }
}

// t1 = k * preA_unscaled t0 for everything that is not fused into K1's epilogue:
// [AMG V-cycle] + [block Jacobi / block Gauss-Seidel | point Jacobi]  (additive MypreA, :383)
// t1 = 0; J.Smooth(t1, k src)  (:377-378).  Colour-major layout: t1 is not zeroed and gathered -- the sweep starts from
// zeros of its own and its first colour needs no pass over A (precond.h: kGsFromZero)
static void gs_forward_from_zero(const nss_bpcg2_t& s, const double* src, hipStream_t st) {
  if (s.pre_bjac->gs_permuted) {
    bjac_smooth(*s.pre_bjac, s.k, src, s.t1, false, s.ctrl, st, kGsFromZero);
    return;
  }
  hipLaunchKernelGGL(bpcg2_zero_kernel, dim3((s.n_u + kBlock - 1) / kBlock), dim3(kBlock), 0, st, s.ctrl, s.n_u, s.t1);
  NSS_CHECK_LAUNCH();
  bjac_smooth(*s.pre_bjac, s.k, src, s.t1, false, s.ctrl, st);
}

void bpcg2_k1_finish(const nss_bpcg2_t& s, hipStream_t st) {
  const double* src = s.t0;
  if (s.cond_HT) {                                   // harmonic_extension(): lift the residual first
    launch_csr_stream(*s.cond_HT, s.t0, EpiLift{s.ctrl, s.t0, s.cond_f}, st);
    src = s.cond_f;
  }
  auto diag = [&](double beta) { diag_apply(s.n_u, s.pre_diag, s.k, src, beta, s.t1, s.ctrl, st); };
  if (s.pre_dist_aux && s.pre_bjac && s.pre_bjac->gs_mat) {
    // multiplicative MypreA on slabs (GS=True, :376-381): the sweeps run inside the slab (additive across slabs), the
    // residual between them with the partitioned A (halo exchange of the iterate), the auxiliary-space term on slabs
    const nss_dist_aux_s& aux = *s.pre_dist_aux;
    gs_forward_from_zero(s, src, st);
    exchange(*aux.d, aux.halo_y, st);
    launch_csr_stream(*s.A, s.t1, EpiScaledResidual{s.ctrl, s.k, src, s.t2}, st);
    dist_aux_apply(aux, 1.0, s.t2, s.t1, true, st, s.ctrl);
    bjac_smooth(*s.pre_bjac, s.k, src, s.t1, true, s.ctrl, st, s.pre_bjac->gs_permuted ? kGsKeepX : 0);   // (src again)
  } else if (s.pre_dist_aux) {                        // additive MypreA on slabs (:383)
    dist_aux_apply(*s.pre_dist_aux, s.k, src, s.t1, false, st, s.ctrl);
    if (s.pre_bjac) bjac_apply(*s.pre_bjac, s.k, src, 1.0, s.t1, s.ctrl, st);
    if (s.pre_diag) diag(1.0);
  } else if (s.pre_dist_amg) {                              // row-partitioned V-cycle (+ additive Jacobi part)
    dist_amg_apply(*s.pre_dist_amg, s.k, src, s.t1, st, s.ctrl);
    if (s.pre_bjac) bjac_apply(*s.pre_bjac, s.k, src, 1.0, s.t1, s.ctrl, st);
    if (s.pre_diag) diag(1.0);
  } else if (s.pre_amg && s.pre_bjac && s.pre_bjac->gs_mat) {
    // multiplicative MypreA (GS=True, :376-381) applied to k * t0:
    //   y = 0; J.Smooth(y, x); r = x - A y; y += M r; J.SmoothBack(y, x)        (t2 is free here: the
    //   previous iteration's t2 was consumed by K1 / C1 and the A-SpMV has not written the new one yet)
    gs_forward_from_zero(s, src, st);
    launch_csr_stream(*s.A, s.t1, EpiScaledResidual{s.ctrl, s.k, src, s.t2}, st);
    amg_apply(*s.pre_amg, 1.0, s.t2, s.t1, st, s.ctrl, true);
    bjac_smooth(*s.pre_bjac, s.k, src, s.t1, true, s.ctrl, st, s.pre_bjac->gs_permuted ? kGsKeepX : 0);   // (src again)
  } else if (s.pre_amg) {
    amg_apply(*s.pre_amg, s.k, src, s.t1, st, s.ctrl);
    if (s.pre_bjac) bjac_apply(*s.pre_bjac, s.k, src, 1.0, s.t1, s.ctrl, st);
    if (s.pre_diag) diag(1.0);
  } else if (s.pre_bjac) {
    bjac_apply_guarded(*s.pre_bjac, s.k, src, s.t1, s.ctrl, st);
  } else if (s.cond_HT) {
    diag(0.0);                                       // uncondensed: rides in K1's epilogue
  }
  if (s.cond_HT) {
    launch_csr_stream(*s.cond_H, s.t1, EpiExtendInPlace{s.ctrl, s.t1}, st);         // t1 += H t1
    launch_csr_stream(*s.cond_inner, s.cond_f, EpiAddGuarded{s.ctrl, s.t1}, st);    // t1 += A_ii^-1 f
  }
}

void bpcg2_phase(const nss_bpcg2_t& s, int which, int it, hipStream_t st) {
  switch (which) {
    case NSS_BPCG2_K1:
      bpcg2_spmv_phase(s, which, it, st, 0, -1);
      bpcg2_k1_finish(s, st);
      break;
    case NSS_BPCG2_K2:
    case NSS_BPCG2_K3:
      bpcg2_spmv_phase(s, which, it, st, 0, -1);
      break;
    case NSS_BPCG2_SUM1:
      if (s.p2p)
        hipLaunchKernelGGL(bpcg2_sum_p2p_kernel, dim3(1), dim3(kSumLanes), 0, st, s.ctrl, s.A->nblk, s.partials_a, s.B->nblk,
                           s.partials_b, s.scal, int(S_AS), int(S_AS_LOCAL), it, s.p2p->view(++s.p2p->seq));
      else
        hipLaunchKernelGGL(bpcg2_sum_kernel, dim3(1), dim3(kSumLanes), 0, st, s.ctrl, s.A->nblk, s.partials_a, s.B->nblk,
                           s.partials_b, s.scal, int(s.local_sums ? S_AS_LOCAL : S_AS));
      NSS_CHECK_LAUNCH();
      break;
    case NSS_BPCG2_ALPHA:   // folded into K4 (kept as a phase id for callers that list all phases)
      break;
    case NSS_BPCG2_K4:
      if (s.dist_compact) throw Error("bpcg2: a state laid out for the compact partitioned plan takes the compact phases only");
      if (s.ghost_p_mode && s.ghost_p_n > 0)     // t3 on the ghost pressure rows (t4 and its ghosts are complete)
        launch_csr_stream(*s.ghost_b, s.t4, EpiGuardedStore{s.ctrl, s.ghost_t3}, st);
      launch_k4(s, it, false, st);
      break;
    case NSS_BPCG2_SUM2:
      if (s.p2p)
        hipLaunchKernelGGL(bpcg2_sum_p2p_kernel, dim3(1), dim3(kSumLanes), 0, st, s.ctrl, k4_partials(s), s.partials_c, 0,
                           s.partials_c, s.scal, int(S_WDN), int(S_WDN_LOCAL), it, s.p2p->view(++s.p2p->seq));
      else
        hipLaunchKernelGGL(bpcg2_sum_kernel, dim3(1), dim3(kSumLanes), 0, st, s.ctrl, k4_partials(s), s.partials_c, 0,
                           s.partials_c, s.scal, int(s.local_sums ? S_WDN_LOCAL : S_WDN));
      NSS_CHECK_LAUNCH();
      break;
    case NSS_BPCG2_BETA:    // folded into K5
      break;
    case NSS_BPCG2_K5:
      if (s.dist_compact) throw Error("bpcg2: a state laid out for the compact partitioned plan takes the compact phases only");
      hipLaunchKernelGGL(bpcg2_k5_kernel, dim3(stream_grid(s.n_p, kBlock * 4)), dim3(kBlock), 0, st, s.ctrl, s.scal,
                         s.hist, it, s.n_p, s.s1, s.w1, s.ghost_mode ? s.ghost_n : 0, s.ghost_s0, s.ghost_w0,
                         s.ghost_p_mode ? s.ghost_p_n : 0, s.s1 + s.n_p, s.ghost_w1);
      NSS_CHECK_LAUNCH();
      break;
    default:
      throw Error("bpcg2: unknown phase");
  }
}

// ---- phases of the compact plan ---------------------------------------------------------------------
static CloseArgs close_args(const nss_bpcg2_t& s, bool fold) {
  return CloseArgs{s.ctrl, s.scal, s.hist, s.partials_c, k4_partials(s), fold ? 1 : 0};
}

// C1 applies the block Jacobi itself when preA is that alone, B^T is planned around its blocks and the system is small:
// measured over 1e4 ... 1e7 DoF (profiles/r03_fuse_bjac_sizes.txt) the launch it removes is worth 25 % of an
// iteration at 1e4 DoF, 8 % at 1e5, 2 % at 2.7e5 -- and from 1e6 DoF on the fused form LOSES 2-3 %: the per-block
// tail of every workgroup, and C23 no longer finds t1 fresh in the memory-side cache (it was the last thing written).
constexpr int kFuseBjacMaxRows = 1 << 18;
static int g_fuse_bjac = -1;          // -1: by size, 0: never, 1: whenever B^T is planned for it
bool fuse_block_jacobi_wanted(int64_t rows) { return g_fuse_bjac == 1 || (g_fuse_bjac == -1 && rows <= kFuseBjacMaxRows); }
bool c1_applies_block_jacobi(const nss_bpcg2_t& s) {
  return fuse_block_jacobi_wanted(s.n_u) && s.pre_bjac && !s.pre_bjac->gs_mat && !s.pre_amg && !s.pre_dist_amg && !s.pre_dist_aux && !s.pre_diag &&
         !s.cond_HT && s.BT->jb_first && s.BT->jb_serial == s.pre_bjac->serial && s.pre_bjac->inv_sym && s.pre_bjac->run;
}

void bpcg2_cphase(const nss_bpcg2_t& s, int which, int it, hipStream_t st) {
  const bool fold = fold_sums(s);
  switch (which) {
    case NSS_BPCG2C_C1: {
      const double* dinv = (s.pre_amg || s.pre_dist_amg || s.pre_dist_aux || s.cond_HT) ? nullptr : s.pre_diag;
      // (streaming operand loads only where B^T takes the row-per-lane kernel: that is the large-system regime,
      // and the stream-kernel instantiations of the epilogue are not doubled)
      const bool nt = s.BT->ell_col != nullptr && stream_vector_loads(s.n_u);
      const bool fj = c1_applies_block_jacobi(s);
      const nss_bjac_s* J = s.pre_bjac;
      const size_t lds = fj ? sizeof(double) * kBlockRows : 0;
#define NSS_C1(FOLD, NT, LAUNCH)                                                                                       \
  LAUNCH(*s.BT, s.s1, EpiK1c<FOLD, NT>{close_args(s, FOLD), s.u0, s.q, s.z0, s.t2, s.s0, s.w0, s.t0, s.t1, dinv, s.k, it, \
                                       s.s1, s.w1, s.dist_compact ? s.ghost_n : 0, s.ghost_s0, s.ghost_w0,                 \
                                       fj ? s.BT->jb_first : nullptr, fj ? s.BT->jb_order : nullptr, fj ? J->run : nullptr, fj ? J->inv_sym : nullptr,   \
                                       fj ? J->nblocks : 0, fj ? J->bs : 0}, st, 0, -1, lds)
      if (nt && fold) NSS_C1(true, true, launch_csr_direct);
      else if (nt) NSS_C1(false, true, launch_csr_direct);
      else if (fold) NSS_C1(true, false, launch_csr_stream);
      else NSS_C1(false, false, launch_csr_stream);
#undef NSS_C1
      if (!fj) bpcg2_k1_finish(s, st);
      break;
    }
    case NSS_BPCG2C_C23: {
      EpiK2c ea{s.ctrl, s.t0, s.s0, s.t2, s.partials_a};
      EpiK3c eb{s.ctrl, s.scal, s.t1, s.s0, s.s1, s.w1, s.t3, s.partials_b, it, s.n_p};
      if (!launch_csr_stream_dual(*s.A, s.t1, ea, *s.B, s.t1, eb, st)) {   // launch plans differ: two launches
        launch_csr_stream(*s.A, s.t1, ea, st);
        launch_csr_stream(*s.B, s.t1, eb, st);
      }
      break;
    }
    case NSS_BPCG2C_SUMA:
      if (!fold) bpcg2_phase(s, NSS_BPCG2_SUM1, it, st);
      break;
    case NSS_BPCG2C_C4:
      launch_k4(s, it, fold, st);
      break;
    case NSS_BPCG2C_SUMW:
      if (!fold) bpcg2_phase(s, NSS_BPCG2_SUM2, it, st);
      break;
    default:
      throw Error("bpcg2: unknown phase of the compact plan");
  }
}

}  // namespace nss

using namespace nss;

extern "C" {

int nss_bpcg2_workspace(const nss_bpcg2_t* s, int64_t* partials_a, int64_t* partials_b, int64_t* partials_c) {
  return guarded([&] {
    NSS_REQUIRE(s && s->A && s->B, "bpcg2_workspace: NULL state / matrices");
    if (partials_a) *partials_a = s->A->nblk;
    if (partials_b) *partials_b = s->B->nblk;
    if (partials_c) *partials_c = k4_partials(*s);
  });
}

int nss_bpcg2_phase(const nss_bpcg2_t* s, int32_t which, int32_t it, nss_stream_t stream) {
  return guarded([&] {
    bpcg2_check_state(s);
    bpcg2_phase(*s, which, it, as_stream(stream));
  });
}

int nss_bpcg2_phases(const nss_bpcg2_t* s, int32_t first, int32_t last, int32_t it, nss_stream_t stream) {
  return guarded([&] {
    bpcg2_check_state(s);
    NSS_REQUIRE(first >= NSS_BPCG2_K1 && last <= NSS_BPCG2_K5 && first <= last, "bpcg2_phases: bad phase range");
    for (int ph = first; ph <= last; ++ph) bpcg2_phase(*s, ph, it, as_stream(stream));
  });
}

int nss_bpcg2_iterate(const nss_bpcg2_t* s, int32_t it_begin, int32_t it_end, nss_stream_t stream) {
  return guarded([&] {
    bpcg2_check_state(s);
    hipStream_t st = as_stream(stream);
    NSS_REQUIRE(!s->ghost_mode && !s->ghost_p_mode && !s->local_sums,
                "bpcg2_iterate: a row-partitioned state takes nss_bpcg2_phase / nss_bpcg2_iterate_dist");
    for (int it = it_begin; it < it_end; ++it)
      for (int ph = NSS_BPCG2C_C1; ph <= NSS_BPCG2C_SUMW; ++ph) bpcg2_cphase(*s, ph, it, st);
  });
}

int nss_bpcg2_iterate_classic(const nss_bpcg2_t* s, int32_t it_begin, int32_t it_end, nss_stream_t stream) {
  return guarded([&] {
    bpcg2_check_state(s);
    hipStream_t st = as_stream(stream);
    for (int it = it_begin; it < it_end; ++it)
      for (int ph = NSS_BPCG2_K1; ph <= NSS_BPCG2_K5; ++ph) bpcg2_phase(*s, ph, it, st);
  });
}

int nss_bpcg2_cphases(const nss_bpcg2_t* s, int32_t first, int32_t last, int32_t it, nss_stream_t stream) {
  return guarded([&] {
    bpcg2_check_state(s);
    NSS_REQUIRE(first >= NSS_BPCG2C_C1 && last <= NSS_BPCG2C_SUMW && first <= last, "bpcg2_cphases: bad phase range");
    NSS_REQUIRE(s->dist_compact || (!s->ghost_mode && !s->ghost_p_mode && !s->local_sums),
                "bpcg2_cphases: single-GPU states, or row-partitioned states laid out for the compact plan");
    for (int ph = first; ph <= last; ++ph) bpcg2_cphase(*s, ph, it, as_stream(stream));
  });
}

int nss_bpcg2_fuse_block_jacobi(int32_t mode) {
  return guarded([&] {
    NSS_REQUIRE(mode >= -1 && mode <= 1, "bpcg2_fuse_block_jacobi: -1 (by size), 0 (never) or 1 (whenever B^T is planned for it)");
    g_fuse_bjac = mode;
  });
}

int nss_bpcg2_c1_applies_preA(const nss_bpcg2_t* s, int32_t* yes) {
  return guarded([&] {
    NSS_REQUIRE(s && s->BT && yes, "bpcg2_c1_applies_preA: NULL argument");
    *yes = c1_applies_block_jacobi(*s) ? 1 : 0;
  });
}

int nss_bpcg2_fold_mode(int32_t mode) {
  return guarded([&] {
    NSS_REQUIRE(mode >= -1 && mode <= 1, "bpcg2_fold_mode: -1 (automatic), 0 (never) or 1 (always)");
    g_fold_mode = mode;
  });
}

int nss_bpcg2_folds_sums(const nss_bpcg2_t* s, int32_t* folds) {
  return guarded([&] {
    NSS_REQUIRE(s && s->A && s->B && folds, "bpcg2_folds_sums: NULL argument");
    *folds = fold_sums(*s) ? 1 : 0;
  });
}

int nss_bpcg2_poll(const nss_bpcg2_t* s, int32_t* done, int32_t* it_final, int32_t* last_it, nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(s && s->ctrl, "bpcg2_poll: NULL state");
    if (s->hist && s->partials_c && s->scal && s->A && s->B) {   // compact plan: the books of the last computed iteration
      hipLaunchKernelGGL(bpcg2_close_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), close_args(*s, fold_sums(*s)));
      NSS_CHECK_LAUNCH();
    }
    if (s->u0 && s->s0 && s->scal && s->n_u > 0) {       // complete the solution (see C_PENDING)
      hipLaunchKernelGGL(bpcg2_flush_kernel, dim3(stream_grid(s->n_u, kBlock * 4)), dim3(kBlock), 0, as_stream(stream),
                         s->ctrl, s->scal, s->n_u, s->s0, s->u0);
      hipLaunchKernelGGL(bpcg2_flush_done_kernel, dim3(1), dim3(1), 0, as_stream(stream), s->ctrl);
      NSS_CHECK_LAUNCH();
    }
    int32_t h[4] = {0, 0, 0, 0};
    NSS_HIP(hipMemcpyAsync(h, s->ctrl, sizeof(int32_t) * 4, hipMemcpyDeviceToHost, as_stream(stream)));
    NSS_HIP(hipStreamSynchronize(as_stream(stream)));
    // 2: alpha = wd / 0 breakdown; 3: a peer of the mailbox transport did not arrive in time
    if (done) *done = h[C_DONE] ? (h[C_BREAKDOWN] == 3 ? 3 : (h[C_BREAKDOWN] ? 2 : 1)) : 0;
    if (it_final) *it_final = h[C_IT_FINAL];
    if (last_it) *last_it = h[C_LAST_IT];
  });
}

}  // extern "C"
