// Fused, device-resident iteration of preconditioned MINRES (reference loop: minres.py:96-144)
// for K = [[A, B^T], [B, 0]] and C = diag(preA, preS).  Four dependent launches per iteration
// (+ one when the preconditioner of the velocity block is not fused, see M3):
//
//   M1  rows of B^T and rows of B in ONE launch (neither depends on the other):
//         kz0 = B^T z1;   kz1 = B z0, partial <kz1, z1>                                 (:97-98)
//   M2  rows of A   : kz0 += A z0, partial <kz0, z0>                                    (:97-98)
//       Launch-bound systems whose B^T has at most two entries per row (the staggered grids) run M1 + M2 as ONE
//       launch: the rows of A add their row of B^T z1 from a fixed-width copy in the epilogue (same products, same
//       order of additions: identical bits), the rows of B share the launch -- three dependent launches per iteration.
//   M3  element-wise, delta = sum of the partials first (in every workgroup when the sum is short,
//        else from the stand-alone sum kernel): v_new = kz - delta v - gamma v_old (:99);
//        z_new = C v_new (:101) -- point Jacobi and preS fused; block Jacobi over runs of
//        consecutive dofs fused too (one lane per block); anything else (index-list blocks,
//        Gauss-Seidel mode, AMG) as its own launches -- partial <z_new, v_new> (:103)
//   M4  element-wise, gamma_new^2 = sum of the partials first; then EVERY workgroup evaluates the
//        Givens recurrences (:107-113), ResNorm (:122) and both stop rules from the same scalars
//        (workgroup 0 records them, history entry and stop flag included) and applies
//        z_new, v_new *= 1/gamma_new (:104-105); w_new = (z - a3 w_old - a2 w)/a1 (:115-116);
//        u += c_new eta_old w_new (:118)
//
// The scalars are double-buffered by the parity of k (M4 of iteration k reads one set while its
// workgroup 0 writes the set of iteration k + 1).
//
// Stop handling: M4 of iteration k records {stop, k_stop}; kernels of iteration k' return at
// once when stop is set and k' > k_stop, so M4 of the stopping iteration still applies the
// update of u (the reference tests *after* :118) and nothing later touches the state.
#include "dist.h"

#include <algorithm>

namespace nss {

enum {
  M_DELTA = 0, M_GAMMA = 1, M_G2 = 2, M_ETA_OLD = 3, M_C_OLD = 4, M_C = 5, M_S_OLD = 6, M_S = 7,
  M_RES_OLD = 8, M_ERR0 = 9, M_TOL = 10, M_A1 = 11, M_A2 = 12, M_A3 = 13, M_UCOEF = 14, M_INVG = 15,
  // z, v and v_old are kept UN-normalised: the reference's `z_new *= 1/gamma_new; v_new *= 1/gamma_new`
  // (:104-105) would cost four vector passes per iteration; instead every consumer multiplies the raw
  // entry by the vector's factor -- fl(raw * factor) is exactly the double the in-place scaling would have
  // stored, so the history is unchanged.  Factors of the current z, the current v and v_old:
  M_SZ = 16, M_SV = 17, M_SVO = 18,
  // row-partitioned runs (nss_minres_t.local_sums): the local sums of delta / gamma_new^2; the all-reduce
  // writes M_DELTA / M_G2 out of place, so the scalars stay frozen after the stop
  M_DELTA_LOC = 19, M_G2_LOC = 20
};
enum { MC_STOP = 0, MC_KSTOP = 1, MC_REASON = 2, MC_LASTK = 3 };

__device__ __forceinline__ bool minres_skip(const int32_t* ctrl, int k) {
  return ctrl[MC_STOP] != 0 && k > ctrl[MC_KSTOP];
}

// operand of the three SpMVs: the current z, stored un-normalised (see M_SZ)
struct XScaledZ {
  const double* __restrict__ z;
  double scale;
  static constexpr bool kStageable = true;
  __device__ const double* ptr() const { return z; }
  __device__ double value(double r) const { return r * scale; }
  __device__ double operator()(int c) const { return z[c] * scale; }
};

struct EpiMStore {  // y = A (scale * x)
  const int32_t* __restrict__ ctrl;
  int k;
  double* __restrict__ y;
  const double* __restrict__ set;    // scalars of iteration k
  double sz = 1.0;
  __device__ bool skip() const { return minres_skip(ctrl, k); }
  __device__ bool prologue(double*) {
    sz = set[M_SZ];
    return true;
  }
  using X = XScaledZ;
  __device__ X xop(const double* x) const { return X{x, sz}; }
  __device__ void row(int r, double ax) const { y[r] = ax; }
  __device__ void finish(int, double*) const {}
};

struct EpiMAccDot {  // y (+)= A (scale * x) ; partial <y, scale * z>
  const int32_t* __restrict__ ctrl;
  int k;
  int accumulate;
  double* __restrict__ y;
  const double* __restrict__ z;
  double* __restrict__ partials;
  const double* __restrict__ set;    // scalars of iteration k
  double acc = 0.0;
  double sz = 1.0;
  __device__ bool skip() const { return minres_skip(ctrl, k); }
  __device__ bool prologue(double*) {
    sz = set[M_SZ];
    return true;
  }
  using X = XScaledZ;
  __device__ X xop(const double* x) const { return X{x, sz}; }
  struct Pre { double y = 0.0, z = 0.0; };
  __device__ Pre fetch(int r) const { return Pre{accumulate ? y[r] : 0.0, z[r]}; }   // (before the prologue)
  __device__ void row(int r, double ax, const Pre& p) {
    const double t = accumulate ? p.y + ax : ax;
    y[r] = t;
    acc = fma(t, p.z * sz, acc);
  }
  __device__ void finish(int b, double* lds) {
    const double s = block_sum(acc, lds);
    if (threadIdx.x == 0 && b >= 0) partials[b] = s;
  }
};

// rows of A with the row of B^T z1 added in the epilogue: kz0 = B^T (sz z1) + A (sz z0), partial <kz0, sz z0>.
// B^T's row comes from its fixed-width copy (two slots per row, column -1 = unused) and is summed exactly as
// csr_direct_kernel sums it (0 + p0 + p1, products rounded on their own); then bts + az as M2 did.
struct EpiMRowsA {
  const int32_t* __restrict__ ctrl;
  int k;
  double* __restrict__ y;
  const double* __restrict__ z;
  double* __restrict__ partials;
  const double* __restrict__ set;
  const int32_t* __restrict__ ecol;
  const double* __restrict__ eval;
  const double* __restrict__ z1;
  double acc = 0.0;
  double sz = 1.0;
  __device__ bool skip() const { return minres_skip(ctrl, k); }
  __device__ bool prologue(double*) {
    sz = set[M_SZ];
    return true;
  }
  using X = XScaledZ;
  __device__ X xop(const double* x) const { return X{x, sz}; }
  struct Pre { double z = 0.0, v0 = 0.0, v1 = 0.0, x0 = 0.0, x1 = 0.0; bool h0 = false, h1 = false; };
  __device__ Pre fetch(int r) const {                        // (before the prologue: raw z1, scaled in row())
    typedef int32_t int2v __attribute__((ext_vector_type(2)));
    const int2v c = reinterpret_cast<const int2v*>(ecol)[r];
    const dbl2v v = reinterpret_cast<const dbl2v*>(eval)[r];
    return Pre{z[r], v.x, v.y, c.x >= 0 ? z1[c.x] : 0.0, c.y >= 0 ? z1[c.y] : 0.0, c.x >= 0, c.y >= 0};
  }
  __device__ void row(int r, double ax, const Pre& p) {
    double bts = 0.0;
    if (p.h0) bts += mul_unfused(p.v0, p.x0 * sz);
    if (p.h1) bts += mul_unfused(p.v1, p.x1 * sz);
    const double t = bts + ax;
    y[r] = t;
    acc = fma(t, p.z * sz, acc);
  }
  __device__ void finish(int b, double* lds) {
    const double s = block_sum(acc, lds);
    if (threadIdx.x == 0 && b >= 0) partials[b] = s;
  }
};

constexpr int kMScal = 32;          // doubles per scalar set; set of iteration k at scal + ((k - 1) & 1) * kMScal
__device__ __host__ __forceinline__ int m_set(int k) { return ((k - 1) & 1) * kMScal; }

// scal_in[slot] = sum(pa[0..na)) + sum(pb[0..nb)) in the fixed order (stand-alone form of the fold)
__global__ __launch_bounds__(kSumLanes) void minres_sum_kernel(const int32_t* __restrict__ ctrl, int k, int na,
                                                                const double* __restrict__ pa, int nb,
                                                                const double* __restrict__ pb,
                                                                double* __restrict__ scal, int slot) {
  __shared__ double lds[kRedDoubles];
  if (minres_skip(ctrl, k)) return;
  const double t = fixed_sum_1024<kSumLanes>(pa, na, pb, nb, lds);
  if (threadIdx.x == 0) scal[slot] = t;
}

struct MK4Args {
  const int32_t* ctrl;
  double* scal;                // set of iteration k
  int32_t n_u, n_p, k;
  const double *kz0, *kz1, *v0, *v1, *vo0, *vo1;
  double *vn0, *vn1, *zn0, *zn1;
  const double *dinv, *minv;   // dinv == nullptr: z_new0 is not a point-Jacobi apply
  double* partials;
  int32_t gu, gp;              // workgroups of the velocity / pressure part (one partial each)
  int32_t fold, na, nb;        // fold: delta = fixed sum of pa[0..na), pb[0..nb) in every workgroup
  const double *pa, *pb;
  // fused block Jacobi over runs of consecutive dofs (BS > 0 instantiations)
  int32_t nblocks;
  const int32_t* run;
  const double* packed;
  int32_t vec;                 // all vectors 16-byte aligned: two doubles per access
};

constexpr int kMPerBlock = 2 * kBlock;     // one-shot element-wise launches: two elements per lane

// (NT = streaming loads of the element-wise operands: see stream_vector_loads, nss_common.h)

__device__ __forceinline__ double m3_delta(const MK4Args& a, double* lds) {
  if (!a.fold) return a.scal[M_DELTA];
  const double d = fixed_sum_1024(a.pa, a.na, a.pb, a.nb, lds);
  if (blockIdx.x == 0 && threadIdx.x == 0) a.scal[M_DELTA] = d;
  return d;
}

// BS == 0: the velocity part is element-wise (point Jacobi when a.dinv, else v_new only);
// BS > 0: one lane per block of <= BS consecutive dofs, z_new0 = J v_new0 from the packed symmetric
// inverse blocks (the arithmetic of bjac_apply_sym_kernel)
template <int BS, bool NT>
__global__ __launch_bounds__(kBlock) void minres_m3_kernel(MK4Args a) {
  __shared__ double lds[kRedDoubles];
  if (minres_skip(a.ctrl, a.k)) return;
  constexpr int B = BS > 0 ? BS : 1;
  const int wg = blockIdx.x;
  const bool velocity = wg < a.gu;
  // ---- stage 1: this lane's operands are requested before the sum of the partials (delta), so that
  //      both latencies overlap (launch-bound small systems) -----------------------------------------
  const int i0 = ((velocity ? wg : wg - a.gu) * kBlock + int(threadIdx.x)) * 2;
  const int n_part = velocity ? a.n_u : a.n_p;
  const bool fast = (BS == 0 || !velocity) && a.vec && i0 + 1 < n_part;
  double2 qkz{}, qv{}, qvo{}, qd{};
  if (fast) {
    // streaming loads: none of these is read again before it is rewritten or a full iteration has passed
    qkz = ld2s<NT>((velocity ? a.kz0 : a.kz1) + i0);
    qv = ld2s<NT>((velocity ? a.v0 : a.v1) + i0);
    qvo = ld2s<NT>((velocity ? a.vo0 : a.vo1) + i0);
    if (!velocity) qd = ld2s<NT>(a.minv + i0);
    else if (a.dinv) qd = ld2s<NT>(a.dinv + i0);
  }
  const int blk = wg * kBlock + int(threadIdx.x);        // BS > 0: one lane per block
  const bool live = BS > 0 && velocity && blk < a.nblocks;
  int32_t first = 0, len = 0;
  double bkz[B], bv[B], bvo[B];
  if (live) {
    const int32_t w = a.run[blk];
    first = w >> 5;
    len = w & 31;
#pragma unroll
    for (int c = 0; c < B; ++c) {
      const bool in = c < len;
      bkz[c] = in ? a.kz0[first + c] : 0.0;
      bv[c] = in ? a.v0[first + c] : 0.0;
      bvo[c] = in ? a.vo0[first + c] : 0.0;
    }
  }
  // ---- stage 2: the scalars ------------------------------------------------------------------------
  const double delta = m3_delta(a, lds), gamma = a.scal[M_GAMMA];
  const double scv = a.scal[M_SV], scvo = a.scal[M_SVO];     // v and v_old are stored un-normalised
  // ---- stage 3 -------------------------------------------------------------------------------------
  double acc = 0.0;
  if (fast) {
    double2 vn;
    vn.x = fma(-gamma, qvo.x * scvo, fma(-delta, qv.x * scv, qkz.x));
    vn.y = fma(-gamma, qvo.y * scvo, fma(-delta, qv.y * scv, qkz.y));
    st2((velocity ? a.vn0 : a.vn1) + i0, vn);
    if (!velocity || a.dinv) {
      double2 zn;
      zn.x = qd.x * vn.x;
      zn.y = qd.y * vn.y;
      st2((velocity ? a.zn0 : a.zn1) + i0, zn);
      acc = fma(zn.x, vn.x, acc);
      acc = fma(zn.y, vn.y, acc);
    }
  } else if (BS == 0 || !velocity) {
    const double* kz = velocity ? a.kz0 : a.kz1;
    const double* v = velocity ? a.v0 : a.v1;
    const double* vo = velocity ? a.vo0 : a.vo1;
    const double* dd = velocity ? a.dinv : a.minv;
    double* vnp = velocity ? a.vn0 : a.vn1;
    double* znp = velocity ? a.zn0 : a.zn1;
    for (int i = i0; i < i0 + 2 && i < n_part; ++i) {
      const double vn = fma(-gamma, vo[i] * scvo, fma(-delta, v[i] * scv, kz[i]));
      vnp[i] = vn;
      if (dd) {
        const double zn = dd[i] * vn;
        znp[i] = zn;
        acc = fma(zn, vn, acc);
      }
    }
  } else if (live) {
    double xv[B], sv[B];
#pragma unroll
    for (int c = 0; c < B; ++c) {
      double vn = 0.0;
      if (c < len) {
        vn = fma(-gamma, bvo[c] * scvo, fma(-delta, bv[c] * scv, bkz[c]));
        a.vn0[first + c] = vn;
      }
      xv[c] = vn;
      sv[c] = 0.0;
    }
    int t = 0;
#pragma unroll
    for (int r = 0; r < B; ++r) {
#pragma unroll
      for (int c = r; c < B; ++c, ++t) {
        const double m = a.packed[size_t(t) * a.nblocks + blk];
        sv[r] = fma(m, xv[c], sv[r]);
        if (c > r) sv[c] = fma(m, xv[r], sv[c]);
      }
    }
#pragma unroll
    for (int r = 0; r < B; ++r) {
      if (r < len) {
        const double zn = sv[r];
        a.zn0[first + r] = zn;
        acc = fma(zn, xv[r], acc);
      }
    }
  }
  const double s = block_sum(acc, lds);
  if (threadIdx.x == 0) a.partials[wg] = s;
}

// partial <x, y> of the velocity block after a preconditioner apply that is not fused
__global__ __launch_bounds__(kBlock) void minres_dot_kernel(const int32_t* __restrict__ ctrl, int k, int32_t n,
                                                             const double* __restrict__ x,
                                                             const double* __restrict__ y,
                                                             double* __restrict__ partials) {
  __shared__ double lds[kRedDoubles];
  if (minres_skip(ctrl, k)) return;
  const int stride = gridDim.x * kBlock;
  double acc = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) acc = fma(x[i], y[i], acc);
  const double s = block_sum(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

struct MK5Args {
  int32_t* ctrl;
  double* scal;                // base of both scalar sets
  double* hist;
  int32_t n_u, n_p, k;
  double *zn0, *zn1, *vn0, *vn1, *wn0, *wn1, *u0, *u1;
  const double *z0, *z1, *wo0, *wo1, *w0, *w1;
  int32_t gu;
  int32_t fold, na, nb;        // fold: gamma_new^2 = fixed sum of pa[0..na), pb[0..nb) in every workgroup
  const double *pa, *pb;
  int32_t vec;
};

__device__ __forceinline__ void minres_k5_body(int i, double sz, double a1inv, double a2, double a3, double uc,
                                                double* wn, double* u, const double* z, const double* wo,
                                                const double* w) {
  double t = fma(-a2, w[i], fma(-a3, wo[i], z[i] * sz));   // :115 (z un-normalised: see M_SZ)
  t *= a1inv;                                         // :116
  NSS_ST(wn[i], t);                                   // next read one iteration later
  NSS_ST(u[i], fma(uc, t, u[i]));                     // :118
}

template <bool NT>
__global__ __launch_bounds__(kBlock) void minres_m4_kernel(MK5Args a) {
  __shared__ double lds[kRedDoubles];
  if (minres_skip(a.ctrl, a.k)) return;
  const double* s = a.scal + m_set(a.k);        // this iteration's scalars (read by every workgroup)
  double* o = a.scal + m_set(a.k + 1);          // next iteration's (written by workgroup 0 only)
  // this lane's operands first, then the sum of the partials: the two latencies overlap
  const int wg = blockIdx.x;
  const bool velocity = wg < a.gu;
  const int i0 = ((velocity ? wg : wg - a.gu) * kBlock + int(threadIdx.x)) * 2;
  const bool fast = a.vec && i0 + 1 < (velocity ? a.n_u : a.n_p);
  double2 qw{}, qwo{}, qz{}, qu{};
  if (fast) {
    qw = ld2s<NT>((velocity ? a.w0 : a.w1) + i0);       // (streaming loads, as in M3)
    qwo = ld2s<NT>((velocity ? a.wo0 : a.wo1) + i0);
    qz = ld2s<NT>((velocity ? a.z0 : a.z1) + i0);
    qu = ld2s<NT>((velocity ? a.u0 : a.u1) + i0);
  }
  const double g2 = a.fold ? fixed_sum_1024(a.pa, a.na, a.pb, a.nb, lds) : s[M_G2];
  const double delta = s[M_DELTA], gamma = s[M_GAMMA];
  const double gamma_new = sqrt(g2);                                   // :103
  const double c = s[M_C], c_old = s[M_C_OLD], sn = s[M_S], s_old = s[M_S_OLD];
  const double alpha0 = c * delta - c_old * sn * gamma;                // :107
  const double alpha1 = sqrt(alpha0 * alpha0 + gamma_new * gamma_new);
  const double alpha2 = sn * delta + c_old * c * gamma;
  const double alpha3 = s_old * gamma;
  const double c_new = alpha0 / alpha1;                                // :112
  const double s_new = gamma_new / alpha1;
  const double eta_old = s[M_ETA_OLD];
  const double uc = c_new * eta_old;                                   // :118
  const double invg = 1.0 / gamma_new;                                 // :104-105
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const double res = fabs(s_new) * s[M_RES_OLD];                     // :122
    a.hist[a.k] = res / s[M_ERR0];                                     // :125
    a.ctrl[MC_LASTK] = a.k;
    // the scalars of iteration k + 1 (:135-144)
    o[M_ETA_OLD] = -s_new * eta_old;
    o[M_S_OLD] = sn;
    o[M_S] = s_new;
    o[M_C_OLD] = c;
    o[M_C] = c_new;
    o[M_GAMMA] = gamma_new;
    o[M_RES_OLD] = res;
    o[M_ERR0] = s[M_ERR0];
    o[M_TOL] = s[M_TOL];
    o[M_SZ] = invg;               // z_new and v_new stay un-normalised (:104-105 applied by their readers)
    o[M_SV] = invg;
    o[M_SVO] = s[M_SV];
    if (res < s[M_TOL] * s[M_ERR0]) {                                  // relative break (:126)
      a.ctrl[MC_KSTOP] = a.k;
      a.ctrl[MC_REASON] = 1;
      a.ctrl[MC_STOP] = 1;
    } else if (!(res > s[M_TOL])) {                                    // absolute guard of the while (:96)
      a.ctrl[MC_KSTOP] = a.k;
      a.ctrl[MC_REASON] = 2;
      a.ctrl[MC_STOP] = 1;
    }
  }
  const double a1inv = 1.0 / alpha1, sz = s[M_SZ];
  if (fast) {
    double2 t;
    t.x = fma(-alpha2, qw.x, fma(-alpha3, qwo.x, qz.x * sz)) * a1inv;      // :115-116
    t.y = fma(-alpha2, qw.y, fma(-alpha3, qwo.y, qz.y * sz)) * a1inv;
    qu.x = fma(uc, t.x, qu.x);                                              // :118
    qu.y = fma(uc, t.y, qu.y);
    store2_nt((velocity ? a.wn0 : a.wn1) + i0, t);
    store2_nt((velocity ? a.u0 : a.u1) + i0, qu);
  } else if (velocity) {
    for (int i = i0; i < i0 + 2 && i < a.n_u; ++i)
      minres_k5_body(i, sz, a1inv, alpha2, alpha3, uc, a.wn0, a.u0, a.z0, a.wo0, a.w0);
  } else {
    for (int i = i0; i < i0 + 2 && i < a.n_p; ++i)
      minres_k5_body(i, sz, a1inv, alpha2, alpha3, uc, a.wn1, a.u1, a.z1, a.wo1, a.w1);
  }
}

static int m_gu(const nss_minres_t& s) { return (s.n_u + kMPerBlock - 1) / kMPerBlock; }
static int m_gp(const nss_minres_t& s) { return (s.n_p + kMPerBlock - 1) / kMPerBlock; }
static int m_dot_grid(const nss_minres_t& s) { return stream_grid(s.n_u, kBlock * 4); }

// block Jacobi that M3 can apply itself: runs of consecutive dofs, symmetric inverse blocks, every dof covered
static bool m_fusable_bjac(const nss_minres_t& s) {
  const nss_bjac_s* j = s.pre_bjac;
  return j != nullptr && !s.pre_amg && !j->gs_mat && j->run != nullptr && j->inv_sym != nullptr && j->n_uncovered == 0;
}
static int g_minres_fuse_mode = -1;
static bool m_fused_bjac(const nss_minres_t& s);
static int m3_gu(const nss_minres_t& s) {
  return m_fused_bjac(s) ? (s.pre_bjac->nblocks + kBlock - 1) / kBlock : m_gu(s);
}

constexpr int kMFoldMax = 1024;      // as kFoldMax of bpcg2.hip (measured there and here: 1e6 DoF +9 % unfolded)
constexpr int kMMergeMaxRows = 1 << 22;   // rows of B^T inside the launch of A's rows up to this many velocity rows
constexpr int kMFuseMax = 4096;      // the block Jacobi fused into M3 still pays at 1e6 DoF (+5 %)
static int g_minres_fold_mode = -1;
static bool m_small(const nss_minres_t& s, int limit) {       // launch-bound regime: every sum of the iteration is short
  int64_t dotg = m_dot_grid(s);
  if (s.pre_bjac) dotg = std::max<int64_t>(dotg, bjac_dot_grid(*s.pre_bjac));
  const int64_t m3 = std::max<int64_t>((s.pre_bjac ? (s.pre_bjac->nblocks + kBlock - 1) / kBlock : 0), m_gu(s)) + m_gp(s);
  return s.A->nblk + s.B->nblk <= limit && m3 <= limit && dotg <= limit;
}
static bool m_fold(const nss_minres_t& s) {
  if (s.local_sums) return false;          // row-partitioned: the sums are all-reduced between the kernels
  if (g_minres_fold_mode >= 0) return g_minres_fold_mode != 0;
  return m_small(s, kMFoldMax);
}
// Fusing the block-Jacobi apply into M3 saves a launch and the re-read of v_new, but turns M3's three
// input streams into 24-byte-strided per-lane accesses: a win where launches dominate, a loss where
// bandwidth does (1e7 DoF: -6 %).  Automatic: fuse exactly in the launch-bound regime.
static bool m_fused_bjac(const nss_minres_t& s) {
  if (!m_fusable_bjac(s)) return false;
  if (g_minres_fuse_mode >= 0) return g_minres_fuse_mode == 1;
  return m_small(s, kMFuseMax);
}

// M1 + M2 as one launch (see the head of the file): in the launch-bound regime, when B^T has a fixed-width copy
static bool m_merged_rows(const nss_minres_t& s) {
  // (row-partitioned runs too: both operands' ghosts arrive in the one grouped exchange in front of the launch, and
  // the fixed-width copy of the slab's B^T carries the column numbers of z1's [owned | ghosts] layout)
  if (g_minres_fuse_mode == 0) return false;
  // measured over 1e4 ... 1e7 DoF (profiles/r03_minres_sizes.txt): -27 % per iteration at 1e5 DoF, -16 % at 1.2e6,
  // -9 % at 4e6, nothing either way at 1e7 -- the launch and the round trip of kz0 it saves against the wider
  // epilogue of A's rows
  if (g_minres_fuse_mode < 0 && s.A->m > kMMergeMaxRows) return false;
  return s.BT->m == s.A->m && fixed_width_copy(*s.BT);
}

static void minres_check(const nss_minres_t* s) {
  NSS_REQUIRE(s != nullptr, "minres: NULL state");
  NSS_REQUIRE(s->A && s->B && s->BT, "minres: NULL matrix handle");
  NSS_REQUIRE(s->A->m == s->n_u && s->BT->m == s->n_u && s->B->m == s->n_p, "minres: matrix rows do not match n_u/n_p");
  if (s->local_sums)      // row-partitioned: the operands carry ghost entries behind the owned ones
    NSS_REQUIRE(s->A->n >= s->n_u && s->B->n >= s->n_u && s->BT->n >= s->n_p, "minres: local matrix narrower than the slab");
  else
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

template <int BS>
static void launch_m3(const MK4Args& a, int grid, hipStream_t st) {
  if (BS == 0 && stream_vector_loads(a.n_u)) hipLaunchKernelGGL((minres_m3_kernel<BS, true>), dim3(grid), dim3(kBlock), 0, st, a);
  else hipLaunchKernelGGL((minres_m3_kernel<BS, false>), dim3(grid), dim3(kBlock), 0, st, a);
}

// phases of one iteration (nss_minres_phases; the row-partitioned schedules all-reduce between them):
//   1 M1 + M2 (the three SpMVs)   2 local sum of delta   3 M3 (+ unfused preA and its dot)
//   4 local sum of gamma_new^2    5 M4
struct MinresDist {
  const nss_dist_s* d;
  const nss_halo_t* hz0;     // z0 in the layout of A's operand (B's local columns are numbered in it too)
  const nss_halo_t* hz1;     // z1 in the layout of B^T's operand
};

static void minres_iteration(const nss_minres_t& s, int k, hipStream_t st, int first = 1, int last = 5,
                             const MinresDist* dist = nullptr) {
  const int io = (k + 2) % 3, ic = k % 3, in = (k + 1) % 3;   // old, current, new
  const int zc = k % 2, zn = (k + 1) % 2;
  const bool fold = m_fold(s);
  double* set = s.scal + m_set(k);
  auto on = [&](int ph) { return first <= ph && ph <= last; };
  if (on(1)) {
  if (dist) {       // both operands in one grouped send/recv phase; z of this iteration sits in ring slot zc
    nss_halo_t h0 = *dist->hz0, h1 = *dist->hz1;
    h0.ext = s.z[zc][0];
    h1.ext = s.z[zc][1];
    exchange(*dist->d, h0, st, &h1);
  }
  // M1: kz0 = B^T z1 and kz1 = B z0 with <kz1, z1>;  M2: kz0 += A z0 with <kz0, z0>
  EpiMAccDot e_b{s.ctrl, k, 0, s.kz[1], s.z[zc][1], s.partials_b, set};
  if (m_merged_rows(s)) {                   // one launch: rows of A (+ their row of B^T z1) and rows of B
    EpiMRowsA e_a{s.ctrl, k, s.kz[0], s.z[zc][0], s.partials_a, set, s.BT->fw_col, s.BT->fw_val, s.z[zc][1]};
    if (!launch_csr_stream_dual(*s.A, s.z[zc][0], e_a, *s.B, s.z[zc][0], e_b, st)) {
      launch_csr_stream(*s.A, s.z[zc][0], e_a, st);
      launch_csr_stream(*s.B, s.z[zc][0], e_b, st);
    }
  } else {
  EpiMStore e_bt{s.ctrl, k, s.kz[0], set};
  if (!launch_csr_stream_dual(*s.BT, s.z[zc][1], e_bt, *s.B, s.z[zc][0], e_b, st)) {
    launch_csr_stream(*s.BT, s.z[zc][1], e_bt, st);
    launch_csr_stream(*s.B, s.z[zc][0], e_b, st);
  }
  launch_csr_stream(*s.A, s.z[zc][0], EpiMAccDot{s.ctrl, k, 1, s.kz[0], s.z[zc][0], s.partials_a, set}, st);
  }
  }
  if (on(2) && !fold) {
    hipLaunchKernelGGL(minres_sum_kernel, dim3(1), dim3(kSumLanes), 0, st, s.ctrl, k, s.A->nblk, s.partials_a,
                       s.B->nblk, s.partials_b, set, int(s.local_sums ? M_DELTA_LOC : M_DELTA));
    NSS_CHECK_LAUNCH();
  }
  if (dist && on(2)) allreduce_sum(*dist->d, set + M_DELTA_LOC, set + M_DELTA, 1, st);
  // M3
  const bool fused = m_fused_bjac(s);
  bool vec = aligned16(s.minv) && (!s.pre_diag || aligned16(s.pre_diag));
  for (int c = 0; c < 2; ++c) {
    vec = vec && aligned16(s.u[c]) && aligned16(s.kz[c]) && aligned16(s.z[0][c]) && aligned16(s.z[1][c]);
    for (int j = 0; j < 3; ++j) vec = vec && aligned16(s.v[j][c]) && aligned16(s.w[j][c]);
  }
  MK4Args a4{s.ctrl, set, s.n_u, s.n_p, k, s.kz[0], s.kz[1], s.v[ic][0], s.v[ic][1], s.v[io][0], s.v[io][1],
             s.v[in][0], s.v[in][1], s.z[zn][0], s.z[zn][1], (s.pre_amg || s.pre_bjac) ? nullptr : s.pre_diag, s.minv,
             s.partials_c, m3_gu(s), m_gp(s), fold ? 1 : 0, s.A->nblk, s.B->nblk, s.partials_a, s.partials_b,
             fused ? s.pre_bjac->nblocks : 0, fused ? s.pre_bjac->run : nullptr, fused ? s.pre_bjac->inv_sym : nullptr,
             vec ? 1 : 0};
  const int g3 = a4.gu + a4.gp;
  int nb2 = 0;
  if (on(3)) {
  if (!fused) {
    launch_m3<0>(a4, g3, st);
  } else {
    switch (s.pre_bjac->bs) {
#define NSS_M3(N) case N: launch_m3<N>(a4, g3, st); break;
      NSS_M3(1) NSS_M3(2) NSS_M3(3) NSS_M3(4) NSS_M3(5) NSS_M3(6) NSS_M3(7) NSS_M3(8)
      NSS_M3(9) NSS_M3(10) NSS_M3(11) NSS_M3(12) NSS_M3(13) NSS_M3(14) NSS_M3(15) NSS_M3(16)
#undef NSS_M3
      default: throw Error("minres: unsupported block size");
    }
  }
  NSS_CHECK_LAUNCH();
  }
  if (!fused && (s.pre_bjac || s.pre_amg)) {
    const bool dot_in_apply = !s.pre_amg && !s.pre_bjac->gs_mat;
    if (!on(3)) {                   // phases 4 / 5 issued on their own: the partial count of phase 3
      nb2 = dot_in_apply ? bjac_dot_grid(*s.pre_bjac) : m_dot_grid(s);
    } else
    // z_new[0] = preA v_new[0] outside the element-wise kernel; after the stop these launches only
    // touch ring slots nobody reads any more
    {
    if (s.pre_amg) {
      amg_apply(*s.pre_amg, 1.0, s.v[in][0], s.z[zn][0], st);
      if (s.pre_bjac) bjac_apply(*s.pre_bjac, 1.0, s.v[in][0], 1.0, s.z[zn][0], nullptr, st);
      if (s.pre_diag) diag_apply(s.n_u, s.pre_diag, 1.0, s.v[in][0], 1.0, s.z[zn][0], nullptr, st);
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
  }
  if (on(4) && !fold) {
    hipLaunchKernelGGL(minres_sum_kernel, dim3(1), dim3(kSumLanes), 0, st, s.ctrl, k, nb2, s.partials_a, g3,
                       s.partials_c, set, int(s.local_sums ? M_G2_LOC : M_G2));
    NSS_CHECK_LAUNCH();
  }
  if (dist && on(4)) allreduce_sum(*dist->d, set + M_G2_LOC, set + M_G2, 1, st);
  if (!on(5)) return;
  // M4
  MK5Args a5{s.ctrl, s.scal, s.hist, s.n_u, s.n_p, k, s.z[zn][0], s.z[zn][1], s.v[in][0], s.v[in][1], s.w[in][0],
             s.w[in][1], s.u[0], s.u[1], s.z[zc][0], s.z[zc][1], s.w[io][0], s.w[io][1], s.w[ic][0], s.w[ic][1],
             m_gu(s), fold ? 1 : 0, nb2, g3, s.partials_a, s.partials_c, vec ? 1 : 0};
  if (stream_vector_loads(s.n_u)) hipLaunchKernelGGL(minres_m4_kernel<true>, dim3(m_gu(s) + m_gp(s)), dim3(kBlock), 0, st, a5);
  else hipLaunchKernelGGL(minres_m4_kernel<false>, dim3(m_gu(s) + m_gp(s)), dim3(kBlock), 0, st, a5);
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
    if (partials_a) *partials_a = std::max<int64_t>(s->A->nblk, dotg);
    if (partials_b) *partials_b = s->B->nblk;
    // M3 in either form (block Jacobi fused: one lane per block; else two entries per lane)
    const int64_t gu_fused = s->pre_bjac ? (s->pre_bjac->nblocks + kBlock - 1) / kBlock : 0;
    if (partials_c) *partials_c = std::max<int64_t>(gu_fused, m_gu(*s)) + m_gp(*s);
  });
}

int nss_minres_iterate(const nss_minres_t* s, int32_t k_begin, int32_t k_end, nss_stream_t stream) {
  return guarded([&] {
    minres_check(s);
    NSS_REQUIRE(k_begin >= 1, "minres_iterate: iterations are counted from 1");
    for (int k = k_begin; k < k_end; ++k) minres_iteration(*s, k, as_stream(stream));
  });
}

int nss_minres_phases(const nss_minres_t* s, int32_t first, int32_t last, int32_t k, nss_stream_t stream) {
  return guarded([&] {
    minres_check(s);
    NSS_REQUIRE(k >= 1 && first >= 1 && last <= 5 && first <= last, "minres_phases: bad phase range / iteration");
    minres_iteration(*s, k, as_stream(stream), first, last);
  });
}

int nss_minres_iterate_dist(const nss_minres_t* s, nss_dist_t d, const nss_halo_t* halo_z0, const nss_halo_t* halo_z1,
                            int32_t k_begin, int32_t k_end, nss_stream_t stream) {
  return guarded([&] {
    minres_check(s);
    NSS_REQUIRE(d != nullptr && s->local_sums, "minres_iterate_dist: needs a dist handle and a row-partitioned state");
    NSS_REQUIRE(k_begin >= 1, "minres_iterate_dist: iterations are counted from 1");
    NSS_REQUIRE(d->nranks == 1 || d->comm != nullptr || d->p2p != nullptr, "minres_iterate_dist: multi-rank run without a communicator");
    // the halo descriptors are checked against ring slot 0; slot 1 has the same layout
    nss_halo_t h0 = *halo_z0, h1 = *halo_z1;
    h0.ext = s->z[0][0];
    h1.ext = s->z[0][1];
    check_halo(&h0, *s->A, "halo_z0");
    check_halo(&h1, *s->BT, "halo_z1");
    MinresDist md{d, halo_z0, halo_z1};
    for (int k = k_begin; k < k_end; ++k) minres_iteration(*s, k, as_stream(stream), 1, 5, &md);
  });
}

int nss_minres_fold_mode(int32_t mode) {
  return guarded([&] {
    NSS_REQUIRE(mode >= -1 && mode <= 1, "minres_fold_mode: -1 (automatic), 0 (never) or 1 (always)");
    g_minres_fold_mode = mode;
  });
}

int nss_minres_fuse_mode(int32_t mode) {
  return guarded([&] {
    NSS_REQUIRE(mode >= -1 && mode <= 2, "minres_fuse_mode: -1 (automatic), 0 (never), 1 (always) or 2 (merged rows only)");
    g_minres_fuse_mode = mode;
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
