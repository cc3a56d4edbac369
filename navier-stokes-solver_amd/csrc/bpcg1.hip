// Fused, device-resident iteration of the textbook Bramble-Pasciak CG (reference loop:
// bramble_pasciak_cg.py:110-143; 6 SpMV per iteration as the reference).
//
//   S0   one lane : hist[it] = err/err0 (:115-118), stop test at the loop top (:119)
//   (V1a + V1c and V3a + V3b are issued as one launch each: same operand, no mutual dependence)
//   V1a  rows of A   : t1u = A du                                                     (:125)
//   V1b  rows of B^T : ku = t1u + B^T dp;  t1u = -ku;  Jacobi: t2u = k dinv ku        (:125-126)
//   [J]  block-Jacobi: t2u = -k J t1u                                                 (:126)
//   V1c  rows of B   : kp = B du;  t1p = -kp;  t2p = kp                               (:125-126)
//   V3a  rows of A   : t1u += A t2u, partial <du, t1u>                                (:127,130)
//   V3b  rows of B   : t1p += B t2u, partial <dp, t1p>                                (:127,130)
//   R1   alpha = rho / sum                                                            (:129)
//   V4   element-wise: x += a d, r -= a t1, a_res -= a t2, partial <a_res_u, r_u>     (:131-133,137)
//   V5   rows of B   : t1p = minv (B a_res_u - a_res_p), partial <t1p, r_p>           (:135,137)
//   R2   rho_new, beta = rho_new / rho                                                (:137-138)
//   V6   element-wise: du = b du + a_res_u, dp = b dp + t1p                           (:140-141)
#include "bpcg2.h"
#include "dist.h"

#include <algorithm>

namespace nss {

// rho of iteration `it` lives in P_RHO (even it) or P_RHO_ODD (odd it): the kernel that forms rho_new writes the
// OTHER slot, so that -- when every workgroup of V6 evaluates beta itself (small systems) -- nobody reads a slot
// that workgroup 0 is rewriting
enum { P_RHO = 0, P_DSUM = 1, P_RHON = 2, P_ALPHA = 3, P_BETA = 4, P_ERR0 = 5, P_TOL = 6, P_RHO_ODD = 7,
       // row-partitioned runs: local totals; the all-reduce writes P_DSUM / P_RHON out of place (frozen after the stop)
       P_DSUM_LOC = 8, P_RHON_LOC = 9 };
enum { PC_STOP = 0, PC_ITSTOP = 1, PC_LAST = 2 };
__host__ __device__ __forceinline__ int rho_slot(int it) { return (it & 1) ? P_RHO_ODD : P_RHO; }

// loop-top bookkeeping (:115-119) evaluated by EVERY workgroup of the first launch of an iteration (small systems):
// hist[it] = err / err0 and the stop test from the same scalars; one thread of the grid records them.  false: stop.
__device__ __forceinline__ bool v1_loop_top(int32_t* ctrl, const double* s, double* hist, int it) {
  const double err = sqrt(fabs(s[rho_slot(it)]));
  const bool stop = err < s[P_TOL] * s[P_ERR0];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    hist[it] = err / s[P_ERR0];
    ctrl[PC_LAST] = it;
    if (stop) {
      ctrl[PC_ITSTOP] = it;
      ctrl[PC_STOP] = 1;
    }
  }
  return !stop;
}

struct EpiStore1 {
  const int32_t* __restrict__ ctrl;
  double* __restrict__ y;
  __device__ bool skip() const { return ctrl[PC_STOP] != 0; }
  __device__ void row(int r, double ax) const { y[r] = ax; }
  __device__ void finish(int, double*) const {}
};

struct EpiV1b {
  const int32_t* __restrict__ ctrl;
  double* __restrict__ t1u;
  double* __restrict__ t2u;
  const double* __restrict__ dinv;
  double k;
  __device__ bool skip() const { return ctrl[PC_STOP] != 0; }
  struct Pre { double t1u = 0.0, dinv = 0.0; };
  __device__ Pre fetch(int r) const { return Pre{t1u[r], dinv ? dinv[r] : 0.0}; }
  __device__ void row(int r, double btp, const Pre& p) const {
    const double ku = p.t1u + btp;
    t1u[r] = -ku;
    if (dinv) t2u[r] = k * (p.dinv * ku);
  }
  __device__ void finish(int, double*) const {}
};

struct EpiV1c {
  const int32_t* __restrict__ ctrl;
  double* __restrict__ t1p;
  double* __restrict__ t2p;
  // small systems: this launch opens the iteration (see v1_loop_top)
  int32_t* top_ctrl = nullptr;
  const double* top_scal = nullptr;
  double* top_hist = nullptr;
  int it = 0;
  __device__ bool skip() const { return ctrl[PC_STOP] != 0; }
  __device__ bool prologue(double*) const { return top_ctrl ? v1_loop_top(top_ctrl, top_scal, top_hist, it) : true; }
  __device__ void row(int r, double kp) const {
    t1p[r] = -kp;
    t2p[r] = kp;
  }
  __device__ void finish(int, double*) const {}
};

// Small systems, B^T with at most two entries per row: the rows of A add their row of B^T dp from B^T's fixed-width
// copy (V1a + V1b in one pass; the row of B^T summed as csr_direct_kernel sums it, then adu + btp as V1b did:
// identical bits) and open the iteration.
struct EpiV1Rows {
  const int32_t* __restrict__ ctrl;
  double* __restrict__ t1u;
  double* __restrict__ t2u;
  const double* __restrict__ dinv;
  double k;
  const int32_t* __restrict__ ecol;
  const double* __restrict__ eval;
  const double* __restrict__ dp;
  int32_t* top_ctrl;
  const double* top_scal;
  double* top_hist;
  int it;
  // block Jacobi applied here (A planned around its blocks, nss_csr_plan_for_blocks): t2u = -k J t1u from the LDS copy
  // of this row block's t1u, one lane per Jacobi block, the arithmetic of bjac_apply_sym_kernel (identical bits)
  const int32_t* __restrict__ jb_first = nullptr;    // nullptr: not fused
  const int32_t* __restrict__ jb_order = nullptr;
  const int32_t* __restrict__ jb_run = nullptr;
  const double* __restrict__ jb_packed = nullptr;
  int32_t jb_count = 0, jb_bs = 0;
  __device__ bool skip() const { return ctrl[PC_STOP] != 0; }
  __device__ bool prologue(double*) const { return top_ctrl ? v1_loop_top(top_ctrl, top_scal, top_hist, it) : true; }
  struct Pre { double dinv = 0.0, v0 = 0.0, v1 = 0.0, x0 = 0.0, x1 = 0.0; bool h0 = false, h1 = false; };
  __device__ Pre fetch(int r) const {
    typedef int32_t int2v __attribute__((ext_vector_type(2)));
    const int2v c = reinterpret_cast<const int2v*>(ecol)[r];
    const dbl2v v = reinterpret_cast<const dbl2v*>(eval)[r];
    return Pre{dinv ? dinv[r] : 0.0, v.x, v.y, c.x >= 0 ? dp[c.x] : 0.0, c.y >= 0 ? dp[c.y] : 0.0, c.x >= 0, c.y >= 0};
  }
  __device__ void row(int r, double adu, const Pre& p) const {
    double btp = 0.0;
    if (p.h0) btp += mul_unfused(p.v0, p.x0);
    if (p.h1) btp += mul_unfused(p.v1, p.x1);
    const double ku = adu + btp;
    t1u[r] = -ku;
    if (dinv) t2u[r] = k * (p.dinv * ku);
    if (jb_first) {
      extern __shared__ double v1_t1u[];
      v1_t1u[r & (kBlockRows - 1)] = -ku;                // (a row block holds at most kBlockRows consecutive rows)
    }
  }
  __device__ void finish(int b, double*) const {
    if (!jb_first || b < 0) return;                      // (uniform over the workgroup)
    extern __shared__ double v1_t1u[];
    __syncthreads();
    const int j1 = jb_first[b + 1];
    for (int pos = jb_first[b] + int(threadIdx.x); pos < j1; pos += kBlock) {
      const int jb = jb_order[pos];
      const int32_t w = jb_run[jb], first = w >> 5, len = w & 31;
      for (int i = 0; i < len; ++i) {
        double s = 0.0;
        for (int j = 0; j < len; ++j) {
          const int lo = i < j ? i : j, hi = i < j ? j : i;
          const int tri = lo * jb_bs - (lo * (lo - 1)) / 2 + (hi - lo);        // upper triangle, row-major
          s = fma(jb_packed[size_t(tri) * jb_count + jb], v1_t1u[(first + j) & (kBlockRows - 1)], s);
        }
        t2u[first + i] = -k * s;
      }
    }
  }
};

struct EpiV3 {  // y += A x ; partial <d, y>
  const int32_t* __restrict__ ctrl;
  double* __restrict__ y;
  const double* __restrict__ d;
  double* __restrict__ partials;
  double acc = 0.0;
  __device__ bool skip() const { return ctrl[PC_STOP] != 0; }
  struct Pre { double y = 0.0, d = 0.0; };
  __device__ Pre fetch(int r) const { return Pre{y[r], d[r]}; }
  __device__ void row(int r, double ax, const Pre& p) {
    const double t = p.y + ax;
    y[r] = t;
    acc = fma(p.d, t, acc);
  }
  __device__ void finish(int b, double* lds) {
    const double s = block_sum(acc, lds);
    if (threadIdx.x == 0 && b >= 0) partials[b] = s;
  }
};

struct EpiV5 {
  const int32_t* __restrict__ ctrl;
  const double* __restrict__ ap;
  const double* __restrict__ minv;
  const double* __restrict__ rp;
  double* __restrict__ t1p;
  double* __restrict__ partials;
  double acc = 0.0;
  __device__ bool skip() const { return ctrl[PC_STOP] != 0; }
  struct Pre { double minv = 0.0, ap = 0.0, rp = 0.0; };
  __device__ Pre fetch(int r) const { return Pre{minv[r], ap[r], rp[r]}; }
  __device__ void row(int r, double bau, const Pre& p) {
    const double t = p.minv * (bau - p.ap);
    t1p[r] = t;
    acc = fma(t, p.rp, acc);
  }
  __device__ void finish(int b, double* lds) {
    const double s = block_sum(acc, lds);
    if (threadIdx.x == 0 && b >= 0) partials[b] = s;
  }
};

constexpr int kPSum = 1024;
// which = 0: loop-top bookkeeping; 1: alpha from sum(pa)+sum(pb); 2: rho_new, beta;
// row-partitioned (local != 0): 1 / 2 only store the local total, 3 / 4 derive alpha / rho_new, beta from the
// all-reduced totals (one lane)
__global__ __launch_bounds__(kPSum) void bpcg1_scalar_kernel(int32_t* __restrict__ ctrl, double* __restrict__ s,
                                                              double* __restrict__ hist, int which, int it, int na,
                                                              const double* __restrict__ pa, int nb,
                                                              const double* __restrict__ pb, int local) {
  __shared__ double lds[2 * kPSum / kWave];
  if (ctrl[PC_STOP] != 0) return;
  const int tid = threadIdx.x;
  if (which == 3 || which == 4) {
    if (tid == 0) {
      if (which == 3) {
        s[P_ALPHA] = s[rho_slot(it)] / s[P_DSUM];
      } else {
        const double total = s[P_RHON];
        s[P_BETA] = total / s[rho_slot(it)];
        s[rho_slot(it + 1)] = total;
      }
    }
    return;
  }
  if (which == 0) {
    if (tid == 0) {
      const double err = sqrt(fabs(s[rho_slot(it)]));
      hist[it] = err / s[P_ERR0];
      ctrl[PC_LAST] = it;
      if (err < s[P_TOL] * s[P_ERR0]) {
        ctrl[PC_ITSTOP] = it;
        ctrl[PC_STOP] = 1;
      }
    }
    return;
  }
  double a = 0.0, b = 0.0;
  for (int i = tid; i < na; i += kPSum) a += pa[i];
  for (int i = tid; i < nb; i += kPSum) b += pb[i];
  const double sa = wave_sum(a), sb = wave_sum(b);
  const int lane = tid & (kWave - 1), wave = tid >> 6;
  if (lane == 0) {
    lds[wave] = sa;
    lds[kPSum / kWave + wave] = sb;
  }
  __syncthreads();
  if (tid == 0) {
    double ta = 0.0, tb = 0.0;
    for (int w = 0; w < kPSum / kWave; ++w) {
      ta += lds[w];
      tb += lds[kPSum / kWave + w];
    }
    const double total = ta + tb;
    if (local) {
      s[which == 1 ? P_DSUM_LOC : P_RHON_LOC] = total;
    } else if (which == 1) {
      s[P_DSUM] = total;
      s[P_ALPHA] = s[rho_slot(it)] / total;
    } else {
      s[P_RHON] = total;
      s[P_BETA] = total / s[rho_slot(it)];
      s[rho_slot(it + 1)] = total;
    }
  }
}

struct V4Args {
  const int32_t* ctrl;
  double* scal;
  int32_t n_u, n_p;
  double *xu, *xp, *ru, *rp, *au, *ap;
  const double *du, *dp, *t1u, *t1p, *t2u, *t2p;
  double* partials;
  // small systems: alpha = rho / (sum(pa) + sum(pb)) evaluated by every workgroup (the tree of bpcg1_scalar_kernel:
  // identical bits), workgroup 0 records it
  int32_t fold, it, na, nb;
  const double *pa, *pb;
};

template <bool NT>      // streaming loads of the operands that are not read again (stream_vector_loads, nss_common.h)
__global__ __launch_bounds__(kBlock) void bpcg1_v4_kernel(V4Args a) {
  __shared__ double lds[kRedDoubles];
  if (a.ctrl[PC_STOP] != 0) return;
  double alpha;
  if (a.fold) {
    const double total = fixed_sum_1024(a.pa, a.na, a.pb, a.nb, lds);
    alpha = a.scal[rho_slot(a.it)] / total;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      a.scal[P_DSUM] = total;
      a.scal[P_ALPHA] = alpha;
    }
  } else {
    alpha = a.scal[P_ALPHA];
  }
  const int stride = gridDim.x * kBlock;
  double acc = 0.0;
  // streaming loads for what is not read again before it is rewritten or an iteration has passed (x, r, t1, t2,
  // a); d stays cached: V6 and the next iteration's SpMVs read it
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < a.n_u; i += stride) {
    NSS_ST(a.xu[i], fma(alpha, a.du[i], ld1s<NT>(&a.xu[i])));
    const double rn = fma(-alpha, ld1s<NT>(&a.t1u[i]), ld1s<NT>(&a.ru[i]));
    const double an = fma(-alpha, ld1s<NT>(&a.t2u[i]), ld1s<NT>(&a.au[i]));
    a.ru[i] = rn;
    a.au[i] = an;
    acc = fma(an, rn, acc);
  }
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < a.n_p; i += stride) {
    NSS_ST(a.xp[i], fma(alpha, a.dp[i], ld1s<NT>(&a.xp[i])));
    a.rp[i] = fma(-alpha, ld1s<NT>(&a.t1p[i]), ld1s<NT>(&a.rp[i]));
    a.ap[i] = fma(-alpha, ld1s<NT>(&a.t2p[i]), ld1s<NT>(&a.ap[i]));
  }
  const double s = block_sum(acc, lds);
  if (threadIdx.x == 0) a.partials[blockIdx.x] = s;
}

// fold: rho_new = sum(pc) + sum(pb), beta = rho_new / rho evaluated by every workgroup; workgroup 0 records them and
// rho_new in the slot of iteration it + 1
template <bool NT>
__global__ __launch_bounds__(kBlock) void bpcg1_v6_kernel(const int32_t* __restrict__ ctrl,
                                                           double* __restrict__ scal, int32_t n_u, int32_t n_p,
                                                           double* __restrict__ du, double* __restrict__ dp,
                                                           const double* __restrict__ au,
                                                           const double* __restrict__ t1p, int fold, int it, int nc,
                                                           const double* __restrict__ pc, int nb,
                                                           const double* __restrict__ pb) {
  __shared__ double lds[kRedDoubles];
  if (ctrl[PC_STOP] != 0) return;
  double beta;
  if (fold) {
    const double total = fixed_sum_1024(pc, nc, pb, nb, lds);
    beta = total / scal[rho_slot(it)];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      scal[P_RHON] = total;
      scal[P_BETA] = beta;
      scal[rho_slot(it + 1)] = total;
    }
  } else {
    beta = scal[P_BETA];
  }
  const int stride = gridDim.x * kBlock;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n_u; i += stride) du[i] = fma(beta, ld1s<NT>(&du[i]), ld1s<NT>(&au[i]));
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n_p; i += stride) dp[i] = fma(beta, ld1s<NT>(&dp[i]), ld1s<NT>(&t1p[i]));
}

static int p_grid(const nss_bpcg1_t& s) { return stream_grid(int64_t(s.n_u) + s.n_p, kBlock * 4); }

static void bpcg1_check(const nss_bpcg1_t* s) {
  NSS_REQUIRE(s != nullptr, "bpcg1: NULL state");
  NSS_REQUIRE(s->A && s->B && s->BT, "bpcg1: NULL matrix handle");
  NSS_REQUIRE(s->A->m == s->n_u && s->BT->m == s->n_u && s->B->m == s->n_p, "bpcg1: matrix rows do not match n_u/n_p");
  if (s->local_sums)      // row-partitioned: the operands carry ghost entries behind the owned ones
    NSS_REQUIRE(s->A->n >= s->n_u && s->B->n == s->A->n && s->BT->n >= s->n_p, "bpcg1: local matrix columns do not match the slab");
  else
    NSS_REQUIRE(s->A->n == s->n_u && s->B->n == s->n_u && s->BT->n == s->n_p, "bpcg1: matrix columns do not match");
  NSS_REQUIRE(!(s->local_sums && s->pre_amg), "bpcg1: the row-partitioned loop takes a (block) Jacobi preconditioner");
  NSS_REQUIRE(!(s->pre_diag && s->pre_bjac), "bpcg1: pre_diag and pre_bjac are exclusive");
  NSS_REQUIRE(s->pre_diag || s->pre_bjac || s->pre_amg, "bpcg1: no preconditioner for the velocity block");
  NSS_REQUIRE(!s->pre_amg || s->pre_amg->levels[0].n == s->n_u, "bpcg1: AMG size mismatch");
  NSS_REQUIRE(!(s->pre_amg && s->pre_bjac && s->pre_bjac->gs_mat), "bpcg1: AMG + Gauss-Seidel mode is not additive");
  NSS_REQUIRE(!s->pre_bjac || s->pre_bjac->n == s->n_u, "bpcg1: block-Jacobi size mismatch");
  NSS_REQUIRE(s->minv && s->scal && s->ctrl && s->hist && s->partials_a && s->partials_b && s->partials_c,
              "bpcg1: NULL work buffer");
  for (int c = 0; c < 2; ++c)
    NSS_REQUIRE(s->x[c] && s->r[c] && s->d[c] && s->a[c] && s->t1[c] && s->t2[c], "bpcg1: NULL vector");
}

struct Bpcg1Dist {
  const nss_dist_s* d;
  const nss_halo_t* hu;      // velocity operands (d[0], t2[0], a[0]): layout of A's operand
  const nss_halo_t* hp;      // d[1]: layout of B^T's operand
};

static void scalar_step(const nss_bpcg1_t& s, int which, int it, int lanes, int na, const double* pa, int nb,
                        const double* pb, hipStream_t st) {
  hipLaunchKernelGGL(bpcg1_scalar_kernel, dim3(1), dim3(lanes), 0, st, s.ctrl, s.scal, s.hist, which, it, na, pa, nb, pb,
                     int(s.local_sums));
  NSS_CHECK_LAUNCH();
}

// One GPU, B^T with a fixed-width copy: the rows of B^T ride in the epilogue of A's rows (`merge`: up to 2^22 velocity
// rows -- measured -10 ... -26 % per iteration up to 1.2e6 DoF, +5 % at 4e6); small systems (every sum of the iteration
// <= 1024 partials) also evaluate the loop-top bookkeeping, alpha and rho_new / beta inside the consuming kernels by
// every workgroup (`fold`) -- 6 dependent launches per iteration instead of 10.  Identical bits in every form
// (nss_bpcg1_fold_mode: -1 by size, 0 neither, 1 both whenever B^T allows).
constexpr int kV1FoldMax = 1024;
constexpr int kV1MergeMaxRows = 1 << 21;
static int g_bpcg1_fold_mode = -1;
static bool v1_merge(const nss_bpcg1_t& s) {
  // (row-partitioned runs too: both operands' ghosts arrive in the grouped exchange in front of the launch, and the
  // slab's copy of B^T indexes dp's [owned | ghosts] layout)
  if (g_bpcg1_fold_mode == 0) return false;
  if (g_bpcg1_fold_mode < 0 && s.A->m > kV1MergeMaxRows) return false;
  return s.BT->m == s.A->m && fixed_width_copy(*s.BT);
}
static bool v1_fold(const nss_bpcg1_t& s) {      // (only together with the merged rows: their launch opens the iteration)
  if (g_bpcg1_fold_mode == 1) return true;
  return s.A->nblk <= kV1FoldMax && s.B->nblk <= kV1FoldMax && p_grid(s) <= kV1FoldMax;
}

// phases first .. last of one iteration (nss_bpcg1_phases); `dist`: exchanges and all-reduces issued from here
static void bpcg1_iteration(const nss_bpcg1_t& s, int it, hipStream_t st, int first = 1, int last = 5,
                            const Bpcg1Dist* dist = nullptr) {
  auto on = [&](int ph) { return first <= ph && ph <= last; };
  auto halo_of = [&](const nss_halo_t* h, double* ext) {
    nss_halo_t c = *h;
    c.ext = ext;
    return c;
  };
  const bool merge = v1_merge(s);                            // rows of B^T inside the launch of A's rows
  bool fused_j = false;                                      // ... and the block Jacobi in the epilogue of that launch
  const bool fast = merge && !dist && !s.local_sums && v1_fold(s);   // ... and the scalar steps inside their consumers
  if (on(1)) {
    if (!fast) scalar_step(s, 0, it, kWave, 0, s.partials_a, 0, s.partials_b, st);
    if (dist) {
      const nss_halo_t h0 = halo_of(dist->hu, s.d[0]), h1 = halo_of(dist->hp, s.d[1]);
      exchange(*dist->d, h0, st, &h1);
    }
    if (merge) {
      // rows of A (+ their row of B^T dp, V1b's combination) and rows of B in one launch, which (fast) opens the iteration
      int32_t* top = fast ? s.ctrl : nullptr;
      const nss_bjac_s* J = s.pre_bjac;
      fused_j = J && !s.pre_amg && !J->gs_mat && J->run && J->inv_sym && s.A->jb_first && s.A->jb_serial == J->serial &&
                !s.A->ell_col && fuse_block_jacobi_wanted(s.A->m);
      const size_t lds = fused_j ? sizeof(double) * kBlockRows : 0;
      const EpiV1Rows e1{s.ctrl, s.t1[0], s.t2[0], s.pre_amg ? nullptr : s.pre_diag, s.k, s.BT->fw_col, s.BT->fw_val,
                         s.d[1], top, s.scal, s.hist, it, fused_j ? s.A->jb_first : nullptr,
                         fused_j ? s.A->jb_order : nullptr, fused_j ? J->run : nullptr, fused_j ? J->inv_sym : nullptr,
                         fused_j ? J->nblocks : 0, fused_j ? J->bs : 0};
      const EpiV1c e1c{s.ctrl, s.t1[1], s.t2[1], top, s.scal, s.hist, it};
      if (!launch_csr_stream_dual(*s.A, s.d[0], e1, *s.B, s.d[0], e1c, st, lds)) {
        launch_csr_stream(*s.A, s.d[0], e1, st, 0, -1, lds);    // (opens the iteration; the second launch repeats the
        launch_csr_stream(*s.B, s.d[0], e1c, st);               //  test from the same scalars: same outcome)
      }
    } else {
    // V1a and V1c multiply the same operand (du) and do not depend on each other: one launch
    const EpiStore1 e1a{s.ctrl, s.t1[0]};
    const EpiV1c e1c{s.ctrl, s.t1[1], s.t2[1]};
    if (!launch_csr_stream_dual(*s.A, s.d[0], e1a, *s.B, s.d[0], e1c, st)) {
      launch_csr_stream(*s.A, s.d[0], e1a, st);
      launch_csr_stream(*s.B, s.d[0], e1c, st);
    }
    launch_csr_stream(*s.BT, s.d[1], EpiV1b{s.ctrl, s.t1[0], s.t2[0], s.pre_amg ? nullptr : s.pre_diag, s.k}, st);
    }
    if (s.pre_amg) {                                         // t2 = -k (AMG + J) t1 (t1 holds -K u here)
      amg_apply(*s.pre_amg, -s.k, s.t1[0], s.t2[0], st);
      if (s.pre_bjac) bjac_apply(*s.pre_bjac, -s.k, s.t1[0], 1.0, s.t2[0], s.ctrl, st);
      if (s.pre_diag) {
        const int rc = nss_diag_apply_f64(s.n_u, s.pre_diag, -s.k, s.t1[0], 1.0, s.t2[0], st);
        if (rc != 0) throw Error(nss_last_error());
      }
    } else if (s.pre_bjac && !fused_j) {
      bjac_apply(*s.pre_bjac, -s.k, s.t1[0], 0.0, s.t2[0], s.ctrl, st);
    }
  }
  if (on(2)) {
    if (dist) exchange(*dist->d, halo_of(dist->hu, s.t2[0]), st);
    // V3a and V3b (operand t2u) likewise
    const EpiV3 e3a{s.ctrl, s.t1[0], s.d[0], s.partials_a}, e3b{s.ctrl, s.t1[1], s.d[1], s.partials_b};
    if (!launch_csr_stream_dual(*s.A, s.t2[0], e3a, *s.B, s.t2[0], e3b, st)) {
      launch_csr_stream(*s.A, s.t2[0], e3a, st);
      launch_csr_stream(*s.B, s.t2[0], e3b, st);
    }
    if (!fast) scalar_step(s, 1, it, kPSum, s.A->nblk, s.partials_a, s.B->nblk, s.partials_b, st);
    if (dist) allreduce_sum(*dist->d, s.scal + P_DSUM_LOC, s.scal + P_DSUM, 1, st);
  }
  if (on(3)) {
    if (s.local_sums) scalar_step(s, 3, it, kWave, 0, s.partials_a, 0, s.partials_b, st);
    V4Args a4{s.ctrl, s.scal, s.n_u, s.n_p, s.x[0], s.x[1], s.r[0], s.r[1], s.a[0], s.a[1],
              s.d[0], s.d[1], s.t1[0], s.t1[1], s.t2[0], s.t2[1], s.partials_c,
              fast ? 1 : 0, it, s.A->nblk, s.B->nblk, s.partials_a, s.partials_b};
    if (stream_vector_loads(s.n_u)) hipLaunchKernelGGL(bpcg1_v4_kernel<true>, dim3(p_grid(s)), dim3(kBlock), 0, st, a4);
    else hipLaunchKernelGGL(bpcg1_v4_kernel<false>, dim3(p_grid(s)), dim3(kBlock), 0, st, a4);
    NSS_CHECK_LAUNCH();
  }
  if (on(4)) {
    if (dist) exchange(*dist->d, halo_of(dist->hu, s.a[0]), st);
    launch_csr_stream(*s.B, s.a[0], EpiV5{s.ctrl, s.a[1], s.minv, s.r[1], s.t1[1], s.partials_b}, st);
    if (!fast) scalar_step(s, 2, it, kPSum, p_grid(s), s.partials_c, s.B->nblk, s.partials_b, st);
    if (dist) allreduce_sum(*dist->d, s.scal + P_RHON_LOC, s.scal + P_RHON, 1, st);
  }
  if (on(5)) {
    if (s.local_sums) scalar_step(s, 4, it, kWave, 0, s.partials_a, 0, s.partials_b, st);
    if (stream_vector_loads(s.n_u))
      hipLaunchKernelGGL(bpcg1_v6_kernel<true>, dim3(p_grid(s)), dim3(kBlock), 0, st, s.ctrl, s.scal, s.n_u, s.n_p, s.d[0],
                         s.d[1], s.a[0], s.t1[1], fast ? 1 : 0, it, p_grid(s), s.partials_c, s.B->nblk, s.partials_b);
    else
      hipLaunchKernelGGL(bpcg1_v6_kernel<false>, dim3(p_grid(s)), dim3(kBlock), 0, st, s.ctrl, s.scal, s.n_u, s.n_p, s.d[0],
                         s.d[1], s.a[0], s.t1[1], fast ? 1 : 0, it, p_grid(s), s.partials_c, s.B->nblk, s.partials_b);
    NSS_CHECK_LAUNCH();
  }
}

}  // namespace nss

using namespace nss;

extern "C" {

int nss_bpcg1_workspace(const nss_bpcg1_t* s, int64_t* partials_a, int64_t* partials_b, int64_t* partials_c) {
  return guarded([&] {
    NSS_REQUIRE(s && s->A && s->B, "bpcg1_workspace: NULL state / matrices");
    if (partials_a) *partials_a = s->A->nblk;
    if (partials_b) *partials_b = s->B->nblk;
    if (partials_c) *partials_c = p_grid(*s);
  });
}

int nss_bpcg1_iterate(const nss_bpcg1_t* s, int32_t it_begin, int32_t it_end, nss_stream_t stream) {
  return guarded([&] {
    bpcg1_check(s);
    for (int it = it_begin; it < it_end; ++it) bpcg1_iteration(*s, it, as_stream(stream));
  });
}

int nss_bpcg1_phases(const nss_bpcg1_t* s, int32_t first, int32_t last, int32_t it, nss_stream_t stream) {
  return guarded([&] {
    bpcg1_check(s);
    NSS_REQUIRE(1 <= first && first <= last && last <= 5, "bpcg1_phases: phases are 1 .. 5");
    bpcg1_iteration(*s, it, as_stream(stream), first, last);
  });
}

int nss_bpcg1_iterate_dist(const nss_bpcg1_t* s, nss_dist_t d, const nss_halo_t* halo_u, const nss_halo_t* halo_p,
                           int32_t it_begin, int32_t it_end, nss_stream_t stream) {
  return guarded([&] {
    bpcg1_check(s);
    NSS_REQUIRE(d != nullptr && s->local_sums, "bpcg1_iterate_dist: needs a dist handle and a row-partitioned state");
    NSS_REQUIRE(halo_u && halo_p, "bpcg1_iterate_dist: NULL halo");
    NSS_REQUIRE(d->nranks == 1 || d->comm != nullptr || d->p2p != nullptr, "bpcg1_iterate_dist: multi-rank run without a communicator");
    nss_halo_t h0 = *halo_u, h1 = *halo_p;
    h0.ext = s->d[0];
    h1.ext = s->d[1];
    check_halo(&h0, *s->A, "halo_u");
    check_halo(&h1, *s->BT, "halo_p");
    Bpcg1Dist bd{d, halo_u, halo_p};
    for (int it = it_begin; it < it_end; ++it) bpcg1_iteration(*s, it, as_stream(stream), 1, 5, &bd);
  });
}

int nss_bpcg1_fold_mode(int32_t mode) {
  return guarded([&] {
    NSS_REQUIRE(mode >= -1 && mode <= 1, "bpcg1_fold_mode: -1 (by size), 0 (never) or 1 (whenever B^T has a fixed-width copy)");
    g_bpcg1_fold_mode = mode;
  });
}

int nss_bpcg1_poll(const nss_bpcg1_t* s, int32_t* stop, int32_t* it_stop, int32_t* last_it, nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(s && s->ctrl, "bpcg1_poll: NULL state");
    int32_t h[4] = {0, 0, 0, 0};
    NSS_HIP(hipMemcpyAsync(h, s->ctrl, sizeof h, hipMemcpyDeviceToHost, as_stream(stream)));
    NSS_HIP(hipStreamSynchronize(as_stream(stream)));
    if (stop) *stop = h[PC_STOP];
    if (it_stop) *it_stop = h[PC_ITSTOP];
    if (last_it) *last_it = h[PC_LAST];
  });
}

}  // extern "C"
