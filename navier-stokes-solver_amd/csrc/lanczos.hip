// Device-resident preconditioned Lanczos: the Ritz values of pre * A behind `EigenValues_Preconditioner`
// (reference call sites: bramble_pasciak_cg.py:68-74, solvers/bramblepasciak_new.py:111-122 -- the scale factor
// k = 1 / min(lambda) + 1e-3; the reference times it inside its solver time, run.py:34-38).
//
// Recurrence (hipla/eigen.py::lanczos_ritz, oracle/krylov_ref.py::lanczos_ritz): r_0 = start, z_0 = pre r_0,
// gamma_0 = sqrt<z_0, r_0>; with v_j = r_j / gamma_j, z_j normalised alike:
//   p = A z_j;  delta_j = <p, z_j>;  r_{j+1} = p - delta_j v_j - gamma_j v_{j-1};  z_{j+1} = pre r_{j+1};
//   gamma_{j+1} = sqrt<z_{j+1}, r_{j+1}>;   T = tridiag(gamma, delta, gamma).
// Here the vectors stay UN-normalised (rh_j = gamma_j v_j, zh_j = gamma_j z_j) and the factors go into the scalars --
// two vector passes fewer per step, the same numbers up to rounding:
//   L1  rows of A    : ph = A zh_j, partial <ph, zh_j>
//   L2  one workgroup: dh = sum
//   L3  element-wise : delta_j = dh / gamma_j^2 (lane 0 records it);
//                      rh_{j+1} = ph / gamma_j - (delta_j / gamma_j) rh_j - (gamma_j / gamma_{j-1}) rh_{j-1}
//                      [point Jacobi: zh_{j+1} = dinv rh_{j+1}, partial <zh_{j+1}, rh_{j+1}>]
//   L4  preconditioner (block Jacobi with the dot in the same kernel / Gauss-Seidel / V-cycle / their additive or
//                      multiplicative combination) + partial <zh_{j+1}, rh_{j+1}>
//   L5  one workgroup: g2 = sum; gamma_{j+1} = sqrt|g2| recorded; breakdown test (gamma_{j+1} <= 1e-14 max(|delta_0|,
//                      |delta_j|)) sets the stop flag, after which every kernel returns at once.
// Small systems (both sums <= 1024 partials, block Jacobi over runs of consecutive dofs) run a step in TWO launches
// instead of five: L1, then one kernel that sums both sets of partials in every workgroup (the fixed tree of
// nss_common.h), does the books of step j - 1 and of L3, and applies L3 + the block Jacobi + the dot from registers
// (one lane per block).  A step is launch-bound at that size: 45 -> ~14 us at 6.7e4 rows.
// Nothing returns to the host inside a batch of steps: the host reads the new (delta, gamma) pairs and the flag once
// per `check_every` steps and solves the small tridiagonal eigenproblem there (round 2: two host-synchronising dots per
// step -- 532 + 553 of them in the set-up of the 1e7-DoF bench).
#include "bpcg2.h"

namespace nss {

enum { L_DH = 0, L_G2 = 1, L_DELTA = 2, L_SCALE0 = 3, L_GAMMA = 4 /* ring of 3 */ };
enum { LC_STOP = 0, LC_JSTOP = 1, LC_LAST = 2 };
__host__ __device__ __forceinline__ int gamma_slot(int j) { return L_GAMMA + ((j % 3) + 3) % 3; }

struct EpiLanczosP {
  const int32_t* __restrict__ ctrl;
  const double* __restrict__ z;
  double* __restrict__ p;
  double* __restrict__ partials;
  double acc = 0.0;
  __device__ bool skip() const { return ctrl[LC_STOP] != 0; }
  struct Pre { double z = 0.0; };
  __device__ Pre fetch(int r) const { return Pre{z[r]}; }
  __device__ void row(int r, double az, const Pre& pre) {
    p[r] = az;
    acc = fma(az, pre.z, acc);
  }
  __device__ void finish(int b, double* lds) {
    const double s = block_sum(acc, lds);
    if (threadIdx.x == 0 && b >= 0) partials[b] = s;
  }
};

constexpr int kLzSum = 1024;
// scal[slot] = sum of the partials; MODE 1: the books of step j (gamma_{j+1}, history entry, breakdown test);
// MODE 2: the start (gamma_0 = sqrt|sum|; 0 stops everything: the operator annihilates the start vector)
template <int MODE>
__global__ __launch_bounds__(kLzSum) void lanczos_sum_kernel(int32_t* __restrict__ ctrl, int n, const double* __restrict__ part,
                                                              double* __restrict__ scal, int slot, int j,
                                                              double* __restrict__ hist) {
  __shared__ double lds[kLzSum / kWave];
  if (ctrl[LC_STOP] != 0) return;
  double a = 0.0, a2 = 0.0;
  int i = threadIdx.x;
  for (; i + kLzSum < n; i += 2 * kLzSum) {
    a += part[i];
    a2 += part[i + kLzSum];
  }
  for (; i < n; i += kLzSum) a += part[i];
  const double s = wave_sum(a + a2);
  if ((threadIdx.x & (kWave - 1)) == 0) lds[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < kLzSum / kWave; ++w) t += lds[w];
    scal[slot] = t;
    if (MODE == 2) {
      const double gamma0 = sqrt(fabs(t));
      scal[gamma_slot(0)] = gamma0;
      if (gamma0 == 0.0) {
        ctrl[LC_JSTOP] = -1;
        ctrl[LC_STOP] = 1;
      }
    }
    if (MODE == 1) {
      const double gamma_new = sqrt(fabs(t));
      const double delta = scal[L_DELTA];
      scal[gamma_slot(j + 1)] = gamma_new;
      hist[2 * j + 1] = gamma_new;
      ctrl[LC_LAST] = j;
      if (gamma_new <= 1e-14 * fmax(scal[L_SCALE0], fabs(delta))) {
        ctrl[LC_JSTOP] = j;
        ctrl[LC_STOP] = 1;
      }
    }
  }
}

struct LzArgs {
  int32_t* ctrl;
  double* scal;
  double* hist;
  int32_t n, j;
  const double *p, *v, *v_old, *dinv;
  double *v_new, *z_new;
  double pre_scale;
  double* partials;
};

// rh_{j+1} = ph / gamma_j - (delta_j / gamma_j) rh_j - (gamma_j / gamma_{j-1}) rh_{j-1}  [+ fused point Jacobi and dot]
__global__ __launch_bounds__(kBlock) void lanczos_combine_kernel(LzArgs a, int fused_pre) {
  __shared__ double lds[kBlock / kWave];
  if (a.ctrl[LC_STOP] != 0) return;
  const double gamma = a.scal[gamma_slot(a.j)];
  const double delta = a.scal[L_DH] / (gamma * gamma);
  const double ca = 1.0 / gamma, cb = -delta / gamma;
  const double cc = a.j > 0 ? -gamma / a.scal[gamma_slot(a.j - 1)] : 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.scal[L_DELTA] = delta;
    if (a.j == 0) a.scal[L_SCALE0] = fabs(delta);
    a.hist[2 * a.j] = delta;
  }
  const int stride = gridDim.x * kBlock;
  double acc = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < a.n; i += stride) {
    double r = fma(cb, a.v[i], ca * a.p[i]);
    if (a.j > 0) r = fma(cc, a.v_old[i], r);
    a.v_new[i] = r;
    if (fused_pre) {
      const double zn = a.pre_scale * (a.dinv[i] * r);
      a.z_new[i] = zn;
      acc = fma(zn, r, acc);
    }
  }
  if (fused_pre) {
    const double s = block_sum(acc, lds);
    if (threadIdx.x == 0) a.partials[blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(kBlock) void lanczos_dot_kernel(const int32_t* __restrict__ ctrl, int32_t n,
                                                              const double* __restrict__ x, const double* __restrict__ y,
                                                              double* __restrict__ partials) {
  __shared__ double lds[kBlock / kWave];
  if (ctrl[LC_STOP] != 0) return;
  const int stride = gridDim.x * kBlock;
  double acc = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) acc = fma(x[i], y[i], acc);
  const double s = block_sum(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// r = x - A y (multiplicative MypreA: the residual between the two sweeps, :379)
struct EpiLzResidual {
  const int32_t* __restrict__ ctrl;
  const double* __restrict__ x;
  double* __restrict__ r;
  __device__ bool skip() const { return ctrl[LC_STOP] != 0; }
  struct Pre { double x = 0.0; };
  __device__ Pre fetch(int i) const { return Pre{x[i]}; }
  __device__ void row(int i, double ay, const Pre& pre) const { r[i] = pre.x - ay; }
  __device__ void finish(int, double*) const {}
};

__global__ __launch_bounds__(kBlock) void lanczos_scale_kernel(const int32_t* __restrict__ ctrl, int32_t n, double a,
                                                                double* __restrict__ x) {
  if (ctrl[LC_STOP] != 0) return;
  const int stride = gridDim.x * kBlock;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) x[i] *= a;
}

// the deterministic, sliceable start vector of hipla/eigen.py::lanczos_start_values, formed where it is used: entry i
// from the global index alone (Knuth's multiplicative hash, 64-bit wrap-around as numpy's uint64) -> [-0.5, 0.5);
// integer arithmetic and one exact division by 2^32: the same bits as the numpy form
__global__ __launch_bounds__(kBlock) void lanczos_start_values_kernel(int64_t offset, int64_t n, double* __restrict__ out) {
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
    unsigned long long h = ((unsigned long long)(offset + i) + 1ull) * 2654435761ull;
    h = (h ^ (h >> 15)) & 0xFFFFFFFFull;
    out[i] = double(h) / 4294967296.0 - 0.5;
  }
}

static int lz_grid(const nss_lanczos_t& s) { return stream_grid(s.n, kBlock * 4); }

// ---- small systems: the step in two launches -----------------------------------------------------------------------
struct LzFusedArgs {
  int32_t* ctrl;
  double* scal;
  double* hist;
  int32_t n, j;
  const double *p, *v, *v_old;
  double *v_new, *z_new;
  double pre_scale;
  int32_t nblocks;
  const int32_t* run;          // first dof * 32 + length of a block
  const double* packed;        // upper triangles of the inverse blocks, [BS (BS + 1) / 2][nblocks]
  const double* pa;            // partials of <ph, zh_j> (na == 0: a stand-alone sum has left the total in scal[L_DH])
  int32_t na;
  const double* pb;            // partials of <zh_j, rh_j>, written by step j - 1 (j > 0)
  int32_t nb;
  double* pout;                // partials of <zh_{j+1}, rh_{j+1}>: the OTHER half of partials_b (late workgroups of
                               // this launch still read pb)
};

// The books of step j - 1 from g2 = <zh_j, rh_j> (what lanczos_sum_kernel<1> does for the five-launch form): gamma_j,
// its history entry, the breakdown test.  Every workgroup derives the same numbers; one thread records them.
__device__ __forceinline__ bool lz_close_step(int32_t* ctrl, double* scal, double* hist, int j, double g2, bool record,
                                              double* gamma_out) {
  const double gamma = sqrt(fabs(g2));
  const double delta_prev = hist[2 * (j - 1)];
  const bool breakdown = gamma <= 1e-14 * fmax(scal[L_SCALE0], fabs(delta_prev));
  if (record) {
    scal[gamma_slot(j)] = gamma;
    hist[2 * (j - 1) + 1] = gamma;
    ctrl[LC_LAST] = j - 1;
    if (breakdown) {
      ctrl[LC_JSTOP] = j - 1;
      ctrl[LC_STOP] = 1;
    }
  }
  *gamma_out = gamma;
  return !breakdown;
}

// end of a batch: the books of the last enqueued step, so that the host finds its gamma in the history (the first
// kernel of the next batch derives the same bits again)
__global__ __launch_bounds__(kBlock) void lanczos_books_kernel(int32_t* ctrl, double* scal, double* hist, int j,
                                                               const double* __restrict__ pb, int nb) {
  __shared__ double lds[kRedDoubles];
  if (ctrl[LC_STOP] != 0) return;
  const SumPair s = fixed_sums_1024(nullptr, 0, pb, nb, lds);
  double gamma;
  lz_close_step(ctrl, scal, hist, j, s.b, threadIdx.x == 0, &gamma);
}

// one lane per block of <= BS consecutive dofs: rh_{j+1} (L3), zh_{j+1} = pre_scale * J rh_{j+1} with the arithmetic of
// bjac_apply_sym_kernel, the partial of their dot
template <int BS>
__global__ __launch_bounds__(kBlock) void lanczos_fused_kernel(LzFusedArgs a) {
  __shared__ double lds[kRedDoubles];
  if (a.ctrl[LC_STOP] != 0) return;
  constexpr int T = BS * (BS + 1) / 2;
  // ---- this lane's operands are requested before the sums: both latencies overlap ---------------------------------
  const int b = blockIdx.x * kBlock + int(threadIdx.x);
  const bool live = b < a.nblocks;
  int32_t first = 0, len = 0;
  double bp[BS], bv[BS], bvo[BS], m[T];
  if (live) {
    const int32_t w = a.run[b];
    first = w >> 5;
    len = w & 31;
#pragma unroll
    for (int c = 0; c < BS; ++c) {
      const bool in = c < len;
      bp[c] = in ? a.p[first + c] : 0.0;
      bv[c] = in ? a.v[first + c] : 0.0;
      bvo[c] = in && a.j > 0 ? a.v_old[first + c] : 0.0;
    }
#pragma unroll
    for (int t = 0; t < T; ++t) m[t] = a.packed[size_t(t) * a.nblocks + b];
  }
  // ---- the scalars ------------------------------------------------------------------------------------------------
  const SumPair s = fixed_sums_1024(a.pa, a.na, a.pb, a.j > 0 ? a.nb : 0, lds);
  const bool record = blockIdx.x == 0 && threadIdx.x == 0;
  double gamma;
  if (a.j > 0) {
    if (!lz_close_step(a.ctrl, a.scal, a.hist, a.j, s.b, record, &gamma)) return;      // (uniform)
  } else {
    gamma = a.scal[gamma_slot(0)];
  }
  const double dh = a.na > 0 ? s.a : a.scal[L_DH];
  const double delta = dh / (gamma * gamma);
  const double ca = 1.0 / gamma, cb = -delta / gamma;
  const double cc = a.j > 0 ? -gamma / a.scal[gamma_slot(a.j - 1)] : 0.0;
  if (record) {
    a.scal[L_DELTA] = delta;
    if (a.j == 0) a.scal[L_SCALE0] = fabs(delta);
    a.hist[2 * a.j] = delta;
  }
  // ---- L3 + block Jacobi + dot ------------------------------------------------------------------------------------
  double acc = 0.0;
  if (live) {
    double r[BS], sum[BS];
#pragma unroll
    for (int c = 0; c < BS; ++c) {
      r[c] = fma(cb, bv[c], ca * bp[c]);
      if (a.j > 0) r[c] = fma(cc, bvo[c], r[c]);
      if (c >= len) r[c] = 0.0;
      sum[c] = 0.0;
    }
    int t = 0;
#pragma unroll
    for (int q = 0; q < BS; ++q) {
#pragma unroll
      for (int c = q; c < BS; ++c, ++t) {
        sum[q] = fma(m[t], r[c], sum[q]);
        if (c > q) sum[c] = fma(m[t], r[q], sum[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < BS; ++c) {
      if (c < len) {
        const double zn = a.pre_scale * sum[c];
        a.v_new[first + c] = r[c];
        a.z_new[first + c] = zn;
        acc = fma(zn, r[c], acc);
      }
    }
  }
  const double part = block_sum(acc, lds);
  if (threadIdx.x == 0) a.pout[blockIdx.x] = part;
}

constexpr int kLzFoldMax = 1024;      // partials per sum up to which every workgroup redoes the sum (as kFoldMax, bpcg2.hip)
constexpr int kLzFusedBs = 8;
static int g_lanczos_fold_mode = -1;  // -1: by size, 0: never, 1: whenever the operands allow

static int lz_fused_grid(const nss_lanczos_t& s) { return (s.pre_bjac->nblocks + kBlock - 1) / kBlock; }
// the two-launch step needs: block Jacobi alone, over runs of consecutive dofs that cover every dof, symmetric inverse
// blocks of at most kLzFusedBs rows; and (by size) at most kLzFoldMax partials of its own dot
static bool lz_fused(const nss_lanczos_t& s) {
  const nss_bjac_s* J = s.pre_bjac;
  if (g_lanczos_fold_mode == 0 || !J || s.pre_amg || s.pre_diag || J->gs_mat || !J->run || !J->inv_sym ||
      J->n_uncovered != 0 || J->bs > kLzFusedBs)
    return false;
  return g_lanczos_fold_mode == 1 || lz_fused_grid(s) <= kLzFoldMax;
}
static bool lz_fold_a(const nss_lanczos_t& s) { return g_lanczos_fold_mode == 1 || s.A->nblk <= kLzFoldMax; }

static void lz_check(const nss_lanczos_t* s) {
  NSS_REQUIRE(s != nullptr && s->A != nullptr, "lanczos: NULL state / matrix");
  NSS_REQUIRE(s->A->m == s->n && s->A->n == s->n, "lanczos: matrix does not match n");
  NSS_REQUIRE(!(s->pre_diag && s->pre_bjac), "lanczos: pre_diag and pre_bjac are exclusive");
  NSS_REQUIRE(s->pre_diag || s->pre_bjac || s->pre_amg, "lanczos: no preconditioner");
  NSS_REQUIRE(!s->pre_bjac || s->pre_bjac->n == s->n, "lanczos: block preconditioner size mismatch");
  NSS_REQUIRE(!s->pre_amg || s->pre_amg->levels.empty() || s->pre_amg->levels[0].n == s->n || s->pre_amg->T, "lanczos: AMG size mismatch");
  for (int i = 0; i < 3; ++i) NSS_REQUIRE(s->v[i] != nullptr, "lanczos: NULL vector");
  NSS_REQUIRE(s->z[0] && s->z[1] && s->p && s->scal && s->ctrl && s->hist && s->partials_a && s->partials_b,
              "lanczos: NULL buffer");
}

// z = pre_scale * pre x for everything that is not the fused point Jacobi; leaves the partials of <z, x> in
// partials_b and returns their count.  `scratch` (n doubles) is free for the multiplicative form.
static int lz_precondition(const nss_lanczos_t& s, const double* x, double* z, double* scratch, hipStream_t st) {
  const int32_t* done = s.ctrl;
  const bool multiplicative = s.pre_amg && s.pre_bjac && s.pre_bjac->gs_mat;
  if (!s.pre_amg && s.pre_bjac && !s.pre_bjac->gs_mat)          // block Jacobi: the dot comes out of the apply kernel
    return bjac_apply_dot(*s.pre_bjac, s.pre_scale, x, z, s.partials_b, done, st);
  if (multiplicative) {
    // MypreA with GS=True (templates/NavierStokesSIMPLE_iterative.py:376-381): y = 0; Smooth; r = x - A y; y += M r; SmoothBack
    if (s.pre_bjac->gs_permuted) {
      bjac_smooth(*s.pre_bjac, 1.0, x, z, false, done, st, kGsFromZero);
    } else {
      NSS_HIP(hipMemsetAsync(z, 0, sizeof(double) * size_t(s.n), st));
      bjac_smooth(*s.pre_bjac, 1.0, x, z, false, done, st);
    }
    launch_csr_stream(*s.A, z, EpiLzResidual{done, x, scratch}, st);
    amg_apply(*s.pre_amg, 1.0, scratch, z, st, done, true);
    bjac_smooth(*s.pre_bjac, 1.0, x, z, true, done, st, s.pre_bjac->gs_permuted ? kGsKeepX : 0);
    if (s.pre_scale != 1.0) {
      hipLaunchKernelGGL(lanczos_scale_kernel, dim3(lz_grid(s)), dim3(kBlock), 0, st, done, s.n, s.pre_scale, z);
      NSS_CHECK_LAUNCH();
    }
  } else if (s.pre_amg) {                                       // V-cycle [+ Jacobi part]: the additive MypreA (:383)
    amg_apply(*s.pre_amg, s.pre_scale, x, z, st, done);
    if (s.pre_bjac) bjac_apply(*s.pre_bjac, s.pre_scale, x, 1.0, z, done, st);
    if (s.pre_diag) diag_apply(s.n, s.pre_diag, s.pre_scale, x, 1.0, z, done, st);
  } else {                                                      // symmetric Gauss-Seidel sweep as an operator
    bjac_apply(*s.pre_bjac, s.pre_scale, x, 0.0, z, done, st);
  }
  const int nb = lz_grid(s);
  hipLaunchKernelGGL(lanczos_dot_kernel, dim3(nb), dim3(kBlock), 0, st, done, s.n, z, x, s.partials_b);
  NSS_CHECK_LAUNCH();
  return nb;
}

static int64_t lz_partials_b(const nss_lanczos_t& s) {
  return std::max<int64_t>(lz_grid(s), s.pre_bjac ? std::max(bjac_dot_grid(*s.pre_bjac), lz_fused_grid(s)) : 0);
}
// the halves of partials_b the two-launch step alternates between
static double* lz_half(const nss_lanczos_t& s, int j) { return s.partials_b + size_t(j & 1) * size_t(lz_partials_b(s)); }

static void lz_step(const nss_lanczos_t& s, int j, bool last, hipStream_t st) {
  double* v = s.v[j % 3];
  double* v_old = s.v[(j + 2) % 3];
  double* v_new = s.v[(j + 1) % 3];
  double* z = s.z[j % 2];
  double* z_new = s.z[(j + 1) % 2];
  launch_csr_stream(*s.A, z, EpiLanczosP{s.ctrl, z, s.p, s.partials_a}, st);
  if (lz_fused(s)) {
    const bool fold_a = lz_fold_a(s);
    if (!fold_a) {
      hipLaunchKernelGGL((lanczos_sum_kernel<0>), dim3(1), dim3(kLzSum), 0, st, s.ctrl, s.A->nblk, s.partials_a, s.scal,
                         int(L_DH), j, s.hist);
      NSS_CHECK_LAUNCH();
    }
    const nss_bjac_s& J = *s.pre_bjac;
    const int g = lz_fused_grid(s);
    LzFusedArgs a{s.ctrl, s.scal, s.hist, s.n, j, s.p, v, v_old, v_new, z_new, s.pre_scale, J.nblocks, J.run, J.inv_sym,
                  s.partials_a, fold_a ? s.A->nblk : 0, lz_half(s, j), g, lz_half(s, j + 1)};
    switch (J.bs) {
#define NSS_LZ(N) case N: hipLaunchKernelGGL((lanczos_fused_kernel<N>), dim3(g), dim3(kBlock), 0, st, a); break;
      NSS_LZ(1) NSS_LZ(2) NSS_LZ(3) NSS_LZ(4) NSS_LZ(5) NSS_LZ(6) NSS_LZ(7) NSS_LZ(8)
#undef NSS_LZ
      default: throw Error("lanczos: unsupported block size");
    }
    NSS_CHECK_LAUNCH();
    if (last) {        // the books of this step for the host; the next batch repeats them bit for bit
      hipLaunchKernelGGL(lanczos_books_kernel, dim3(1), dim3(kBlock), 0, st, s.ctrl, s.scal, s.hist, j + 1, lz_half(s, j + 1), g);
      NSS_CHECK_LAUNCH();
    }
    return;
  }
  hipLaunchKernelGGL((lanczos_sum_kernel<0>), dim3(1), dim3(kLzSum), 0, st, s.ctrl, s.A->nblk, s.partials_a, s.scal,
                     int(L_DH), j, s.hist);
  NSS_CHECK_LAUNCH();
  const bool fused_pre = s.pre_diag && !s.pre_amg;
  LzArgs a{s.ctrl, s.scal, s.hist, s.n, j, s.p, v, v_old, s.pre_diag, v_new, z_new, s.pre_scale, s.partials_b};
  hipLaunchKernelGGL(lanczos_combine_kernel, dim3(lz_grid(s)), dim3(kBlock), 0, st, a, fused_pre ? 1 : 0);
  NSS_CHECK_LAUNCH();
  const int nb = fused_pre ? lz_grid(s) : lz_precondition(s, v_new, z_new, s.p, st);    // (p is free from here on)
  hipLaunchKernelGGL((lanczos_sum_kernel<1>), dim3(1), dim3(kLzSum), 0, st, s.ctrl, nb, s.partials_b, s.scal,
                     int(L_G2), j, s.hist);
  NSS_CHECK_LAUNCH();
}

}  // namespace nss


// ---- host side: the two extreme eigenvalues of the Lanczos tridiagonal ---------------------------------------------
// What the convergence check of a batch needs (hipla/eigen.py).  LAPACK's bisection (dstebz) costs ~0.6 us per row
// and check -- with a check every 5 steps that was MORE host time than the device needs for the steps of a small
// system (6.7e4 rows: 6 ms of checks vs 4 ms of steps).  Here: Laguerre's iteration on the characteristic polynomial
// from outside the spectrum -- for a polynomial with only real roots it moves monotonically to the nearest root and
// never passes it, with cubic convergence -- over the division-free three-term recurrences of p, p', p'' (rescaled
// together when they grow; p'/p and p''/p do not see the scale); the result is then checked by two Sturm counts and
// plain bisection takes over if they disagree.
namespace {

struct Tridiag {
  const double* d;
  const double* e;      // e[i] couples rows i and i + 1
  int n;
};

// number of eigenvalues < x (negative pivots of T - x I)
int sturm_count(const Tridiag& t, double x, double tiny) {
  int count = 0;
  double q = t.d[0] - x;
  if (q < 0.0) ++count;
  for (int i = 1; i < t.n; ++i) {
    if (fabs(q) < tiny) q = q < 0.0 ? -tiny : tiny;
    q = (t.d[i] - x) - t.e[i - 1] * t.e[i - 1] / q;
    if (q < 0.0) ++count;
  }
  return count;
}

// k-th smallest eigenvalue (k = 0 or n - 1 here) by bisection inside [lo, hi]
double bisect(const Tridiag& t, int k, double lo, double hi, double tiny) {
  for (int it = 0; it < 200; ++it) {
    const double mid = 0.5 * (lo + hi);
    if (mid <= lo || mid >= hi) break;
    if (sturm_count(t, mid, tiny) > k) hi = mid; else lo = mid;
  }
  return 0.5 * (lo + hi);
}

// p'/p and p''/p of the characteristic polynomial at x
void log_derivatives(const Tridiag& t, double x, double* g, double* h2) {
  double p0 = 1.0, p1 = t.d[0] - x;        // p_{i-1}, p_i
  double q0 = 0.0, q1 = -1.0;              // first derivatives
  double r0 = 0.0, r1 = 0.0;               // second derivatives
  for (int i = 1; i < t.n; ++i) {
    const double a = t.d[i] - x, b = t.e[i - 1] * t.e[i - 1];
    const double p2 = a * p1 - b * p0;
    const double q2 = a * q1 - p1 - b * q0;
    const double r2 = a * r1 - 2.0 * q1 - b * r0;
    p0 = p1; p1 = p2;
    q0 = q1; q1 = q2;
    r0 = r1; r1 = r2;
    const double big = fmax(fabs(p1), fmax(fabs(q1), fabs(r1)));
    if (big > 1e150 || (big < 1e-150 && big > 0.0)) {
      const double sc = 1.0 / big;
      p0 *= sc; p1 *= sc; q0 *= sc; q1 *= sc; r0 *= sc; r1 *= sc;
    }
  }
  *g = q1 / p1;
  *h2 = r1 / p1;
}

// the smallest (side < 0, from `x` left of the spectrum) or largest (side > 0, from the right) eigenvalue; whatever
// comes back is checked by the caller's Sturm counts
double laguerre_extreme(const Tridiag& t, int side, double x, double span, double eps_abs) {
  const double n = double(t.n);
  for (int it = 0; it < 100; ++it) {
    double g, h2;
    log_derivatives(t, x, &g, &h2);
    if (!std::isfinite(g) || !std::isfinite(h2)) return x;      // on a root of p (or overflow)
    const double hh = g * g - h2;
    const double disc = (n - 1.0) * (n * hh - g * g);
    const double root = disc > 0.0 ? sqrt(disc) : 0.0;
    const double den = g < 0.0 ? g - root : g + root;            // the larger magnitude: the nearest root
    if (den == 0.0) return x;
    const double step = n / den;                                 // x_new = x - step
    if ((side < 0 && step > 0.0) || (side > 0 && step < 0.0)) return x;    // rounding noise at the root (or not outside)
    if (fabs(step) > 4.0 * span) return x;
    const double x_new = x - step;
    if (fabs(step) <= eps_abs || x_new == x) return x_new;
    x = x_new;
  }
  return x;
}

}  // namespace

using namespace nss;

extern "C" {

int nss_tridiag_extremes(const double* diag, const double* off, int32_t n, double* lo_out, double* hi_out) {
  return guarded([&] {
    NSS_REQUIRE(diag && n >= 1 && (off || n == 1) && lo_out && hi_out, "tridiag_extremes: bad arguments");
    if (n == 1) {
      *lo_out = *hi_out = diag[0];
      return;
    }
    const Tridiag t{diag, off, n};      // (the hints in *lo_out / *hi_out are read below)
    double gl = diag[0], gu = diag[0], norm = 0.0;               // Gershgorin interval
    for (int i = 0; i < n; ++i) {
      const double r = (i > 0 ? fabs(off[i - 1]) : 0.0) + (i + 1 < n ? fabs(off[i]) : 0.0);
      gl = fmin(gl, diag[i] - r);
      gu = fmax(gu, diag[i] + r);
      norm = fmax(norm, fabs(diag[i]) + r);
    }
    NSS_REQUIRE(std::isfinite(gl) && std::isfinite(gu), "tridiag_extremes: non-finite entries");
    const double span = fmax(gu - gl, 1e-300), eps = 2.3e-16 * fmax(norm, 1e-300), tiny = 1e-300 + 1e-30 * norm;
    gl -= 1e-3 * span + eps;
    gu += 1e-3 * span + eps;
    // accepted when two counts put the extreme eigenvalue within 16 ulps of the matrix norm of the iterate
    const double w = 16.0 * eps;
    // a hint is used when a count confirms that it lies outside the spectrum (a few iterations from there instead of
    // the ~10-40 from the Gershgorin bound: the Ritz values of a Lanczos run hardly move between two checks)
    double start_lo = gl, start_hi = gu;
    if (std::isfinite(*lo_out) && *lo_out > gl && *lo_out < gu && sturm_count(t, *lo_out, tiny) == 0) start_lo = *lo_out;
    if (std::isfinite(*hi_out) && *hi_out > gl && *hi_out < gu && sturm_count(t, *hi_out, tiny) == n) start_hi = *hi_out;
    double lo = laguerre_extreme(t, -1, start_lo, span, 2.0 * eps);
    if (!(sturm_count(t, lo - w, tiny) == 0 && sturm_count(t, lo + w, tiny) >= 1)) lo = bisect(t, 0, gl, gu, tiny);
    double hi = laguerre_extreme(t, +1, start_hi, span, 2.0 * eps);
    if (!(sturm_count(t, hi + w, tiny) == n && sturm_count(t, hi - w, tiny) <= n - 1)) hi = bisect(t, n - 1, gl, gu, tiny);
    *lo_out = lo;
    *hi_out = hi;
  });
}

int nss_lanczos_workspace(const nss_lanczos_t* s, int64_t* partials_a, int64_t* partials_b) {
  return guarded([&] {
    NSS_REQUIRE(s && s->A, "lanczos_workspace: NULL state / matrix");
    if (partials_a) *partials_a = s->A->nblk;
    if (partials_b) *partials_b = 2 * lz_partials_b(*s);        // two halves: see LzFusedArgs::pout
  });
}

int nss_lanczos_start(const nss_lanczos_t* s, nss_stream_t stream) {
  return guarded([&] {
    lz_check(s);
    hipStream_t st = as_stream(stream);
    // zh_0 = pre rh_0 (rh_0 = v[0], given), gamma_0 = sqrt|<zh_0, rh_0>| -> scal; control words cleared
    NSS_HIP(hipMemsetAsync(s->ctrl, 0, sizeof(int32_t) * 4, st));
    NSS_HIP(hipMemsetAsync(s->scal, 0, sizeof(double) * 8, st));
    int nb;
    if (s->pre_diag && !s->pre_amg) {
      diag_apply(s->n, s->pre_diag, s->pre_scale, s->v[0], 0.0, s->z[0], nullptr, st);
      nb = lz_grid(*s);
      hipLaunchKernelGGL(lanczos_dot_kernel, dim3(nb), dim3(kBlock), 0, st, s->ctrl, s->n, s->z[0], s->v[0], s->partials_b);
      NSS_CHECK_LAUNCH();
    } else {
      nb = lz_precondition(*s, s->v[0], s->z[0], s->p, st);
    }
    hipLaunchKernelGGL((lanczos_sum_kernel<2>), dim3(1), dim3(kLzSum), 0, st, s->ctrl, nb, s->partials_b, s->scal,
                       int(L_G2), 0, s->hist);
    NSS_CHECK_LAUNCH();
  });
}

int nss_lanczos_iterate(const nss_lanczos_t* s, int32_t j_begin, int32_t j_end, nss_stream_t stream) {
  return guarded([&] {
    lz_check(s);
    NSS_REQUIRE(j_begin >= 0 && j_end >= j_begin, "lanczos_iterate: bad step range");
    for (int j = j_begin; j < j_end; ++j) lz_step(*s, j, j + 1 == j_end, as_stream(stream));
  });
}

int nss_lanczos_start_values(int64_t offset, int64_t n, double* out, nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(offset >= 0 && n >= 0 && (out || n == 0), "lanczos_start_values: bad arguments");
    if (n == 0) return;
    hipLaunchKernelGGL(lanczos_start_values_kernel, dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, as_stream(stream), offset, n, out);
    NSS_CHECK_LAUNCH();
  });
}

int nss_lanczos_fold_mode(int32_t mode) {
  return guarded([&] {
    NSS_REQUIRE(mode >= -1 && mode <= 1, "lanczos_fold_mode: -1 (by size), 0 (never) or 1 (whenever the operands allow)");
    g_lanczos_fold_mode = mode;
  });
}

int nss_lanczos_poll(const nss_lanczos_t* s, int32_t* stop, int32_t* j_stop, int32_t* last_j, nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(s && s->ctrl, "lanczos_poll: NULL state");
    int32_t h[4] = {0, 0, 0, 0};
    NSS_HIP(hipMemcpyAsync(h, s->ctrl, sizeof h, hipMemcpyDeviceToHost, as_stream(stream)));
    NSS_HIP(hipStreamSynchronize(as_stream(stream)));
    if (stop) *stop = h[LC_STOP];
    if (j_stop) *j_stop = h[LC_JSTOP];
    if (last_j) *last_j = h[LC_LAST];
  });
}

}  // extern "C"
