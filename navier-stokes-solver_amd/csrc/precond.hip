// Additive block-Jacobi preconditioner: set-up (gather A_bb from CSR, invert on the
// device) and apply (batched dense bs x bs mat-vec, one lane per block).
//
// Storage is block-interleaved -- inv[(r*bs + c)*nblocks + b], idx[c*nblocks + b] -- so
// consecutive lanes (consecutive blocks) read consecutive addresses: the bs*bs inverse
// entries stream from HBM fully coalesced (8*bs^2 bytes per block, the dominant
// traffic), the bs gathers of x and scatters of y hit neighbouring lines when the
// blocks are laid out along the grid.  No MFMA: the op is 2 flop per 8 bytes.
#include "precond.h"

#include <string>
#include <vector>

#ifndef NSS_BJAC_SYM
#define NSS_BJAC_SYM 1
#endif

namespace nss {

// One lane per block: gather the dense block from CSR rows, Gauss-Jordan with partial
// pivoting in private memory, store the inverse interleaved.  Set-up only.
__global__ __launch_bounds__(kBlock) void bjac_setup_kernel(int32_t bs, int32_t nb, const int32_t* __restrict__ idx,
                                                             const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ col,
                                                             const double* __restrict__ val,
                                                             double* __restrict__ inv, int32_t* __restrict__ singular) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= nb) return;
  double m[kMaxBs][2 * kMaxBs];  // [A_bb | I]
  int32_t dof[kMaxBs];
  for (int r = 0; r < bs; ++r) dof[r] = idx[r * nb + b];
  for (int r = 0; r < bs; ++r) {
    for (int c = 0; c < 2 * bs; ++c) m[r][c] = 0.0;
    m[r][bs + r] = 1.0;
    if (dof[r] < 0) {
      m[r][r] = 1.0;  // padding row: identity
      continue;
    }
    for (int p = rowptr[dof[r]]; p < rowptr[dof[r] + 1]; ++p) {
      const int cc = col[p];
      for (int c = 0; c < bs; ++c)
        if (dof[c] == cc) m[r][c] = val[p];
    }
  }
  bool bad = false;
  for (int k = 0; k < bs; ++k) {
    int piv = k;
    double best = fabs(m[k][k]);
    for (int r = k + 1; r < bs; ++r)
      if (fabs(m[r][k]) > best) {
        best = fabs(m[r][k]);
        piv = r;
      }
    if (best == 0.0) {
      bad = true;
      break;
    }
    if (piv != k)
      for (int c = 0; c < 2 * bs; ++c) {
        const double t = m[k][c];
        m[k][c] = m[piv][c];
        m[piv][c] = t;
      }
    const double d = 1.0 / m[k][k];
    for (int c = 0; c < 2 * bs; ++c) m[k][c] *= d;
    for (int r = 0; r < bs; ++r) {
      if (r == k) continue;
      const double f = m[r][k];
      if (f != 0.0)
        for (int c = 0; c < 2 * bs; ++c) m[r][c] = fma(-f, m[k][c], m[r][c]);
    }
  }
  if (bad) atomicAdd(singular, 1);
  for (int r = 0; r < bs; ++r)
    for (int c = 0; c < bs; ++c) inv[(size_t(r) * bs + c) * nb + b] = bad ? 0.0 : m[r][bs + c];
}

// upper triangles of the inverse blocks (averaged with the mirrored entry); *asym counts blocks
// whose inverse is not symmetric to 1e-12
__global__ __launch_bounds__(kBlock) void bjac_pack_sym_kernel(int bs, int32_t nb, const double* __restrict__ inv,
                                                                double* __restrict__ packed,
                                                                int32_t* __restrict__ asym) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= nb) return;
  bool bad = false;
  int t = 0;
  for (int r = 0; r < bs; ++r)
    for (int c = r; c < bs; ++c, ++t) {
      const double u = inv[(size_t(r) * bs + c) * nb + b], l = inv[(size_t(c) * bs + r) * nb + b];
      if (fabs(u - l) > 1e-12 * fmax(fabs(u), fabs(l))) bad = true;
      if (packed) packed[size_t(t) * nb + b] = 0.5 * (u + l);
    }
  if (bad) atomicAdd(asym, 1);
}

// symmetric inverse blocks: every stored entry is read once and used for both triangles (NT: with a streaming
// load -- stream_vector_loads, nss_common.h)
template <int BS, bool NT>
__global__ __launch_bounds__(kBlock) void bjac_apply_sym_kernel(int32_t nb, const int32_t* __restrict__ idx,
                                                                 const int32_t* __restrict__ run,
                                                                 const double* __restrict__ packed, double alpha,
                                                                 const double* __restrict__ x, double beta,
                                                                 double* __restrict__ y,
                                                                 const int32_t* __restrict__ done,
                                                                 double* __restrict__ partials) {
  __shared__ double red[kBlock / kWave];
  if (done && done[0] != 0) return;
  const int stride = gridDim.x * kBlock;
  double acc = 0.0;                              // partial <y, x> of this lane (partials != NULL)
  for (int b = blockIdx.x * kBlock + threadIdx.x; b < nb; b += stride) {
    int32_t dof[BS];
    double xv[BS], s[BS];
    if (run) {                                   // consecutive dofs: one word per block
      const int32_t w = run[b], first = w >> 5, len = w & 31;
#pragma unroll
      for (int c = 0; c < BS; ++c) dof[c] = c < len ? first + c : -1;
    } else {
#pragma unroll
      for (int c = 0; c < BS; ++c) dof[c] = idx[size_t(c) * nb + b];
    }
#pragma unroll
    for (int c = 0; c < BS; ++c) {
      xv[c] = dof[c] >= 0 ? x[dof[c]] : 0.0;
      s[c] = 0.0;
    }
    int t = 0;
#pragma unroll
    for (int r = 0; r < BS; ++r) {
#pragma unroll
      for (int c = r; c < BS; ++c, ++t) {
        const double m = ld1s<NT>(&packed[size_t(t) * nb + b]);
        s[r] = fma(m, xv[c], s[r]);
        if (c > r) s[c] = fma(m, xv[r], s[c]);
      }
    }
#pragma unroll
    for (int r = 0; r < BS; ++r) {
      if (dof[r] >= 0) {
        double v = alpha * s[r];
        if (beta != 0.0) v = fma(beta, y[dof[r]], v);
        y[dof[r]] = v;
        acc = fma(v, xv[r], acc);
      }
    }
  }
  if (partials) {
    const double t = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
  }
}

template <int BS>
__global__ __launch_bounds__(kBlock) void bjac_apply_kernel(int32_t nb, const int32_t* __restrict__ idx,
                                                             const double* __restrict__ inv, double alpha,
                                                             const double* __restrict__ x, double beta,
                                                             double* __restrict__ y, const int32_t* __restrict__ done,
                                                             double* __restrict__ partials) {
  __shared__ double red[kBlock / kWave];
  if (done && done[0] != 0) return;
  const int stride = gridDim.x * kBlock;
  double acc = 0.0;
  for (int b = blockIdx.x * kBlock + threadIdx.x; b < nb; b += stride) {
    int32_t dof[BS];
    double xv[BS];
#pragma unroll
    for (int c = 0; c < BS; ++c) dof[c] = idx[size_t(c) * nb + b];
#pragma unroll
    for (int c = 0; c < BS; ++c) xv[c] = dof[c] >= 0 ? x[dof[c]] : 0.0;
#pragma unroll
    for (int r = 0; r < BS; ++r) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < BS; ++c) s = fma(inv[(size_t(r) * BS + c) * nb + b], xv[c], s);
      if (dof[r] >= 0) {
        double t = alpha * s;
        if (beta != 0.0) t = fma(beta, y[dof[r]], t);
        y[dof[r]] = t;
        acc = fma(t, xv[r], acc);
      }
    }
  }
  if (partials) {
    const double t = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
  }
}

// dofs that belong to no block: y = beta * y
__global__ __launch_bounds__(kBlock) void bjac_uncovered_kernel(int32_t count, const int32_t* __restrict__ dofs,
                                                                 double beta, double* __restrict__ y,
                                                                 const int32_t* __restrict__ done) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= count || (done && done[0] != 0)) return;
  const int d = dofs[i];
  y[d] = beta == 0.0 ? 0.0 : beta * y[d];
}

// ---- multicolour block Gauss-Seidel ---------------------------------------------------------
// A colour is swept in two streaming launches:
//   (1) residual: CSR-stream SpMV over the colour's contiguous rows of the row-permuted copy of
//       A, epilogue res[r] = xscale * x[dof(r)] - (A y)[r]  (blocks of a colour are uncoupled, so
//       the old y of the colour itself is all the launch reads of it);
//   (2) block solve: one lane per block, y[dofs] += A_bb^-1 res_b, inverse blocks and residual
//       rows contiguous in colour-major order (coalesced), y scattered by original dof.
struct EpiGsResidual {
  const int32_t* __restrict__ done;
  const int32_t* __restrict__ rowdof;
  const double* __restrict__ x;
  double* __restrict__ res;
  double xscale;
  __device__ bool skip() const { return done && done[0] != 0; }
  struct Pre { double x = 0.0; };
  __device__ Pre fetch(int r) const { return Pre{x[rowdof[r]]}; }
  __device__ void row(int r, double ay, const Pre& p) const { res[r] = fma(xscale, p.x, -ay); }
  __device__ void finish(int, double*) const {}
};

template <int BS>
__global__ __launch_bounds__(kBlock) void bgs_solve_kernel(int32_t b0, int32_t b1, int32_t nb,
                                                            const int32_t* __restrict__ idx,
                                                            const int32_t* __restrict__ ridx,
                                                            const double* __restrict__ inv,
                                                            const double* __restrict__ res, double* __restrict__ y,
                                                            const int32_t* __restrict__ done) {
  if (done && done[0] != 0) return;
  const int b = b0 + blockIdx.x * kBlock + threadIdx.x;
  if (b >= b1) return;
  double rv[BS];
#pragma unroll
  for (int c = 0; c < BS; ++c) {
    const int rr = ridx[size_t(c) * nb + b];
    rv[c] = rr >= 0 ? res[rr] : 0.0;
  }
#pragma unroll
  for (int r = 0; r < BS; ++r) {
    const int dof = idx[size_t(r) * nb + b];
    if (dof < 0) continue;
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < BS; ++c) s = fma(inv[(size_t(r) * BS + c) * nb + b], rv[c], s);
    y[dof] += s;
  }
}

// ---- colour-major layout inside the sweep (nss_bjac_s::gs_permuted) ------------------------------------------
// One launch per colour: rows of the colour of P A P^T times the permuted iterate, residual of every row into LDS
// (a row block holds at most kGsRows consecutive rows: slot r mod kGsRows), then -- all rows of the row block done --
// every lane applies its row of the inverse block to the residuals of its block: yt[r] += sum_k ginv[k][r] res[first + k].
// Same products in the same order as the two-launch form (EpiGsResidual + bgs_solve_kernel): same bits.
struct EpiGsFused {
  const int32_t* __restrict__ done;
  const int32_t* __restrict__ rowblk;
  const double* __restrict__ xt;
  double* __restrict__ yt;
  const double* __restrict__ ginv;
  const uint8_t* __restrict__ gpos;
  const uint8_t* __restrict__ glen;
  int32_t n_perm;
  double xscale;
  __device__ bool skip() const { return done && done[0] != 0; }
  struct Pre { double x = 0.0; };
  __device__ Pre fetch(int r) const { return Pre{xt[r]}; }
  __device__ void row(int r, double ay, const Pre& p) const {
    extern __shared__ double gs_res[];
    gs_res[r & (kGsRows - 1)] = fma(xscale, p.x, -ay);
  }
  __device__ void finish(int b, double*) const {
    extern __shared__ double gs_res[];
    if (b < 0) return;                                    // (uniform over the workgroup)
    __syncthreads();
    const int r0 = rowblk[b], r1 = rowblk[b + 1];
    for (int r = r0 + int(threadIdx.x); r < r1; r += kBlock) {
      const int first = r - int(gpos[r]), len = int(glen[r]);
      double s = 0.0;
      for (int k = 0; k < len; ++k) s = fma(ginv[size_t(k) * n_perm + r], gs_res[(first + k) & (kGsRows - 1)], s);
      yt[r] += s;
    }
  }
};

// entry: xt = x[rowdof], yt = y[rowdof] (or 0), the extra slot behind them 0
__global__ __launch_bounds__(kBlock) void gs_enter_kernel(int32_t n_perm, const int32_t* __restrict__ rowdof,
                                                           const double* __restrict__ x, const double* __restrict__ y,
                                                           double* __restrict__ xt, double* __restrict__ yt,
                                                           const int32_t* __restrict__ done) {
  if (done && done[0] != 0) return;
  const int stride = gridDim.x * kBlock;
  for (int r = blockIdx.x * kBlock + threadIdx.x; r <= n_perm; r += stride) {
    const bool live = r < n_perm;
    const int d = live ? rowdof[r] : 0;
    if (x) xt[r] = live ? x[d] : 0.0;                    // (x == NULL: the permuted copy of the previous call stays)
    yt[r] = (live && y) ? y[d] : 0.0;
  }
}

// exit: y[rowdof] = yt
__global__ __launch_bounds__(kBlock) void gs_leave_kernel(int32_t n_perm, const int32_t* __restrict__ rowdof,
                                                           const double* __restrict__ yt, double* __restrict__ y,
                                                           const int32_t* __restrict__ done) {
  if (done && done[0] != 0) return;
  const int stride = gridDim.x * kBlock;
  for (int r = blockIdx.x * kBlock + threadIdx.x; r < n_perm; r += stride) y[rowdof[r]] = yt[r];
}

// ginv[k][r] for the rows of one block: (A_bb^-1)(position of r, k-th live position of the block)
__global__ __launch_bounds__(kBlock) void gs_pack_inverse_kernel(int32_t bs, int32_t nb, int32_t n_perm,
                                                                  const int32_t* __restrict__ ridx,
                                                                  const double* __restrict__ inv,
                                                                  double* __restrict__ ginv, uint8_t* __restrict__ gpos,
                                                                  uint8_t* __restrict__ glen) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= nb) return;
  int live[kMaxBs], len = 0;
  for (int c = 0; c < bs; ++c)
    if (ridx[size_t(c) * nb + b] >= 0) live[len++] = c;
  for (int i = 0; i < len; ++i) {
    const int r = ridx[size_t(live[i]) * nb + b];
    gpos[r] = uint8_t(i);
    glen[r] = uint8_t(len);
    for (int k = 0; k < bs; ++k)
      ginv[size_t(k) * n_perm + r] = k < len ? inv[(size_t(live[i]) * bs + live[k]) * nb + b] : 0.0;
  }
}

// first colour of a sweep that starts from y = 0: its rows see A y = 0, so the residual is xscale x and
// y_c = D_c^-1 (xscale x_c) -- the numbers the colour launch would produce (fma(xscale, x, -0) = xscale x; 0 + s = s)
// without its pass over half of P A P^T
__global__ __launch_bounds__(kBlock) void gs_first_color_kernel(int32_t r0, int32_t r1, const double* __restrict__ xt,
                                                                 double* __restrict__ yt, const double* __restrict__ ginv,
                                                                 const uint8_t* __restrict__ gpos,
                                                                 const uint8_t* __restrict__ glen, int32_t n_perm,
                                                                 double xscale, const int32_t* __restrict__ done) {
  if (done && done[0] != 0) return;
  const int r = r0 + blockIdx.x * kBlock + threadIdx.x;
  if (r >= r1) return;
  const int first = r - int(gpos[r]), len = int(glen[r]);
  double s = 0.0;
  for (int k = 0; k < len; ++k) s = fma(ginv[size_t(k) * n_perm + r], xscale * xt[first + k], s);
  yt[r] = s;
}

static void gs_sweep_permuted(const nss_bjac_s& j, double xscale, bool backward, const int32_t* done, hipStream_t st,
                              bool from_zero = false) {
  const int nc = int(j.color_ptr.size()) - 1;
  const EpiGsFused epi{done, j.gs_mat->rowblk, j.xt, j.yt, j.ginv, j.gpos, j.glen, j.n_perm, xscale};
  for (int k = 0; k < nc; ++k) {
    const int c = backward ? nc - 1 - k : k;
    if (k == 0 && from_zero && !j.color_row.empty()) {
      const int r0 = j.color_row[c], r1 = j.color_row[c + 1];
      if (r1 > r0) {
        hipLaunchKernelGGL(gs_first_color_kernel, dim3((r1 - r0 + kBlock - 1) / kBlock), dim3(kBlock), 0, st, r0, r1, j.xt,
                           j.yt, j.ginv, j.gpos, j.glen, j.n_perm, xscale, done);
        NSS_CHECK_LAUNCH();
      }
      continue;
    }
    launch_csr_stream(*j.gs_mat, j.yt, epi, st, j.color_rowblk[c], j.color_rowblk[c + 1], sizeof(double) * kGsRows);
  }
}

static void gs_enter(const nss_bjac_s& j, const double* x, const double* y, const int32_t* done, hipStream_t st) {
  hipLaunchKernelGGL(gs_enter_kernel, dim3(stream_grid(int64_t(j.n_perm) + 1, kBlock)), dim3(kBlock), 0, st, j.n_perm,
                     j.rowdof, x, y, j.xt, j.yt, done);
  NSS_CHECK_LAUNCH();
}

static void gs_leave(const nss_bjac_s& j, double* y, const int32_t* done, hipStream_t st) {
  hipLaunchKernelGGL(gs_leave_kernel, dim3(stream_grid(j.n_perm, kBlock)), dim3(kBlock), 0, st, j.n_perm, j.rowdof, j.yt,
                     y, done);
  NSS_CHECK_LAUNCH();
}

template <int BS>
static void launch_bgs_solve(const nss_bjac_s& j, int c, double* y, const int32_t* done, hipStream_t st) {
  const int b0 = j.color_ptr[c], b1 = j.color_ptr[c + 1];
  if (b1 <= b0) return;
  hipLaunchKernelGGL((bgs_solve_kernel<BS>), dim3((b1 - b0 + kBlock - 1) / kBlock), dim3(kBlock), 0, st, b0, b1,
                     j.nblocks, j.idx, j.ridx, j.inv, j.res, y, done);
}

void bjac_smooth(const nss_bjac_s& j, double xscale, const double* x, double* y, bool backward, const int32_t* done,
                 hipStream_t st, int flags) {
  if (!j.gs_mat) throw Error("bjac_smooth: colours not set (nss_bjac_set_colors)");
  if (j.gs_permuted) {             // gather x and y into the colour-major numbering, sweep, scatter y back
    const bool from_zero = (flags & kGsFromZero) != 0;
    if (from_zero && j.n_uncovered > 0) {      // dofs in no block: 0 (guarded: a frozen solver keeps its y)
      hipLaunchKernelGGL(bjac_uncovered_kernel, dim3((j.n_uncovered + kBlock - 1) / kBlock), dim3(kBlock), 0, st,
                         j.n_uncovered, j.covered, 0.0, y, done);
      NSS_CHECK_LAUNCH();
    }
    gs_enter(j, (flags & kGsKeepX) ? nullptr : x, from_zero ? nullptr : y, done, st);
    gs_sweep_permuted(j, xscale, backward, done, st, from_zero);
    gs_leave(j, y, done, st);
    return;
  }
  if (flags & kGsFromZero) throw Error("bjac_smooth: the from-zero form exists for the colour-major layout only");
  const int nc = int(j.color_ptr.size()) - 1;
  for (int k = 0; k < nc; ++k) {
    const int c = backward ? nc - 1 - k : k;
    launch_csr_stream(*j.gs_mat, y, EpiGsResidual{done, j.rowdof, x, j.res, xscale}, st, j.color_rowblk[c],
                      j.color_rowblk[c + 1]);
    switch (j.bs) {
#define NSS_GS(N) case N: launch_bgs_solve<N>(j, c, y, done, st); break;
      NSS_GS(1) NSS_GS(2) NSS_GS(3) NSS_GS(4) NSS_GS(5) NSS_GS(6) NSS_GS(7) NSS_GS(8)
      NSS_GS(9) NSS_GS(10) NSS_GS(11) NSS_GS(12) NSS_GS(13) NSS_GS(14) NSS_GS(15) NSS_GS(16)
#undef NSS_GS
      default: throw Error("bjac_smooth: unsupported block size");
    }
    NSS_CHECK_LAUNCH();
  }
}

void bjac_symgs_apply(const nss_bjac_s& j, double xscale, const double* x, double* y, const int32_t* done,
                      hipStream_t st) {
  if (j.gs_permuted) {             // both sweeps inside the colour-major numbering: one gather, one scatter
    if (j.n_uncovered > 0) {       // dofs in no block stay 0 (guarded: a frozen solver keeps its y)
      hipLaunchKernelGGL(bjac_uncovered_kernel, dim3((j.n_uncovered + kBlock - 1) / kBlock), dim3(kBlock), 0, st,
                         j.n_uncovered, j.covered, 0.0, y, done);
      NSS_CHECK_LAUNCH();
    }
    gs_enter(j, x, nullptr, done, st);                               // y[:] = 0 (:377)
    gs_sweep_permuted(j, xscale, false, done, st, true);             // jacobi.Smooth(y, x)      (:378)
    gs_sweep_permuted(j, xscale, true, done, st);                    // jacobi.SmoothBack(y, x)  (:381)
    gs_leave(j, y, done, st);
    return;
  }
  NSS_HIP(hipMemsetAsync(y, 0, sizeof(double) * size_t(j.n), st));   // y[:] = 0 (:377)
  bjac_smooth(j, xscale, x, y, false, done, st);                     // jacobi.Smooth(y, x)      (:378)
  bjac_smooth(j, xscale, x, y, true, done, st);                      // jacobi.SmoothBack(y, x)  (:381)
}

template <int BS>
static void launch_bjac(const nss_bjac_s& j, double alpha, const double* x, double beta, double* y,
                        const int32_t* done, double* partials, hipStream_t st) {
  const int grid = bjac_dot_grid(j);
  if (j.inv_sym && stream_vector_loads(j.n))
    hipLaunchKernelGGL((bjac_apply_sym_kernel<BS, true>), dim3(grid), dim3(kBlock), 0, st, j.nblocks, j.idx, j.run,
                       j.inv_sym, alpha, x, beta, y, done, partials);
  else if (j.inv_sym)
    hipLaunchKernelGGL((bjac_apply_sym_kernel<BS, false>), dim3(grid), dim3(kBlock), 0, st, j.nblocks, j.idx, j.run,
                       j.inv_sym, alpha, x, beta, y, done, partials);
  else
    hipLaunchKernelGGL((bjac_apply_kernel<BS>), dim3(grid), dim3(kBlock), 0, st, j.nblocks, j.idx, j.inv, alpha, x,
                       beta, y, done, partials);
}

int bjac_dot_grid(const nss_bjac_s& j) { return stream_grid(j.nblocks, kBlock); }

static void bjac_apply_impl(const nss_bjac_s& j, double alpha, const double* x, double beta, double* y,
                            const int32_t* done, double* partials, hipStream_t st);

void bjac_apply(const nss_bjac_s& j, double alpha, const double* x, double beta, double* y, const int32_t* done,
                hipStream_t st) {
  bjac_apply_impl(j, alpha, x, beta, y, done, nullptr, st);
}

int bjac_apply_dot(const nss_bjac_s& j, double alpha, const double* x, double* y, double* partials, const int32_t* done,
                   hipStream_t st) {
  if (j.gs_mat) throw Error("bjac_apply_dot: not available in Gauss-Seidel mode");
  if (partials == nullptr) throw Error("bjac_apply_dot: NULL partials");
  bjac_apply_impl(j, alpha, x, 0.0, y, done, partials, st);   // uncovered dofs get y = 0: nothing to add
  return bjac_dot_grid(j);
}

static void bjac_apply_impl(const nss_bjac_s& j, double alpha, const double* x, double beta, double* y,
                            const int32_t* done, double* partials, hipStream_t st) {
  if (j.gs_mat) {  // a handle in Gauss-Seidel mode is the symmetric sweep operator
    if (beta != 0.0) throw Error("bjac_apply: Gauss-Seidel mode supports beta == 0 only");
    bjac_symgs_apply(j, alpha, x, y, done, st);
    return;
  }
  switch (j.bs) {
#define NSS_BJ(N) case N: launch_bjac<N>(j, alpha, x, beta, y, done, partials, st); break;
    NSS_BJ(1) NSS_BJ(2) NSS_BJ(3) NSS_BJ(4) NSS_BJ(5) NSS_BJ(6) NSS_BJ(7) NSS_BJ(8)
    NSS_BJ(9) NSS_BJ(10) NSS_BJ(11) NSS_BJ(12) NSS_BJ(13) NSS_BJ(14) NSS_BJ(15) NSS_BJ(16)
#undef NSS_BJ
    default: throw Error("bjac_apply: unsupported block size");
  }
  NSS_CHECK_LAUNCH();
  if (j.n_uncovered > 0) {
    hipLaunchKernelGGL(bjac_uncovered_kernel, dim3((j.n_uncovered + kBlock - 1) / kBlock), dim3(kBlock), 0, st,
                       j.n_uncovered, j.covered, beta, y, done);
    NSS_CHECK_LAUNCH();
  }
}

}  // namespace nss

using namespace nss;

extern "C" {

int nss_bjac_create(nss_csr_t a, int32_t bs, int32_t nblocks, const int32_t* h_idx, nss_bjac_t* out) {
  return guarded([&] {
    NSS_REQUIRE(a != nullptr && out != nullptr && h_idx != nullptr, "bjac_create: NULL argument");
    NSS_REQUIRE(a->m == a->n, "bjac_create: matrix must be square");
    NSS_REQUIRE(bs >= 1 && bs <= kMaxBs, "bjac_create: 1 <= bs <= 16");
    NSS_REQUIRE(nblocks >= 1, "bjac_create: need at least one block");
    // host-side shape check: indices in range and disjoint (the kernels assume it)
    std::vector<uint8_t> seen(size_t(a->m), 0);
    for (int64_t i = 0; i < int64_t(bs) * nblocks; ++i) {
      const int32_t d = h_idx[i];
      if (d < 0) continue;
      NSS_REQUIRE(d < a->m, "bjac_create: dof index out of range");
      NSS_REQUIRE(!seen[d], "bjac_create: blocks must be disjoint");
      seen[d] = 1;
    }
    std::vector<int32_t> uncovered;
    for (int32_t d = 0; d < a->m; ++d)
      if (!seen[d]) uncovered.push_back(d);
    nss_bjac_s* j = new nss_bjac_s;
    int32_t* singular = nullptr;
    try {
      static uint64_t next_serial = 0;
      j->serial = ++next_serial;
      j->bs = bs;
      j->nblocks = nblocks;
      j->n = a->m;
      j->n_uncovered = int32_t(uncovered.size());
      NSS_HIP(hipMalloc(&j->idx, sizeof(int32_t) * size_t(bs) * nblocks));
      NSS_HIP(hipMalloc(&j->inv, sizeof(double) * size_t(bs) * bs * nblocks));
      NSS_HIP(hipMemcpy(j->idx, h_idx, sizeof(int32_t) * size_t(bs) * nblocks, hipMemcpyHostToDevice));
      if (!uncovered.empty()) {
        NSS_HIP(hipMalloc(&j->covered, sizeof(int32_t) * uncovered.size()));
        NSS_HIP(hipMemcpy(j->covered, uncovered.data(), sizeof(int32_t) * uncovered.size(), hipMemcpyHostToDevice));
      }
      {   // blocks that are runs of consecutive dofs (padding last): one packed word per block
        std::vector<int32_t> runs{};
        runs.resize(size_t(nblocks));
        bool all_runs = a->m < (1 << 26);
        for (int32_t b = 0; b < nblocks && all_runs; ++b) {
          const int32_t first = h_idx[b];
          int len = 0;
          while (len < bs && h_idx[size_t(len) * nblocks + b] >= 0) ++len;
          all_runs = first >= 0 && len > 0;
          for (int c = 0; c < bs && all_runs; ++c) {
            const int32_t d = h_idx[size_t(c) * nblocks + b];
            all_runs = c < len ? d == first + c : d < 0;
          }
          runs[b] = first * 32 + len;
        }
        if (all_runs) {
          NSS_HIP(hipMalloc(&j->run, sizeof(int32_t) * size_t(nblocks)));
          NSS_HIP(hipMemcpy(j->run, runs.data(), sizeof(int32_t) * size_t(nblocks), hipMemcpyHostToDevice));
        }
      }
      NSS_HIP(hipMalloc(&singular, sizeof(int32_t)));
      NSS_HIP(hipMemset(singular, 0, sizeof(int32_t)));
      hipLaunchKernelGGL(bjac_setup_kernel, dim3((nblocks + kBlock - 1) / kBlock), dim3(kBlock), 0, nullptr, bs,
                         nblocks, j->idx, a->rowptr, a->col, a->val, j->inv, singular);
      NSS_CHECK_LAUNCH();
      int32_t nsing = 0;
      NSS_HIP(hipMemcpy(&nsing, singular, sizeof(int32_t), hipMemcpyDeviceToHost));
      (void)hipFree(singular);
      singular = nullptr;
      if (nsing > 0) throw Error("bjac_create: " + std::to_string(nsing) + " singular diagonal block(s)");
#if NSS_BJAC_SYM
      if (bs > 1) {   // symmetric inverses (A symmetric): keep the packed upper triangles for the apply kernel
        NSS_HIP(hipMalloc(&singular, sizeof(int32_t)));
        NSS_HIP(hipMemset(singular, 0, sizeof(int32_t)));
        NSS_HIP(hipMalloc(&j->inv_sym, sizeof(double) * size_t(bs) * (bs + 1) / 2 * nblocks));
        hipLaunchKernelGGL(bjac_pack_sym_kernel, dim3((nblocks + kBlock - 1) / kBlock), dim3(kBlock), 0, nullptr, bs,
                           nblocks, j->inv, j->inv_sym, singular);
        NSS_CHECK_LAUNCH();
        int32_t nasym = 0;
        NSS_HIP(hipMemcpy(&nasym, singular, sizeof(int32_t), hipMemcpyDeviceToHost));
        (void)hipFree(singular);
        singular = nullptr;
        if (nasym > 0) {
          (void)hipFree(j->inv_sym);
          j->inv_sym = nullptr;
        }
      }
#endif
    } catch (...) {
      (void)hipFree(singular);
      nss_bjac_destroy(j);
      throw;
    }
    *out = j;
  });
}

int nss_bjac_destroy(nss_bjac_t j) {
  return guarded([&] {
    if (!j) return;
    (void)hipFree(j->idx);
    (void)hipFree(j->inv);
    (void)hipFree(j->inv_sym);
    (void)hipFree(j->run);
    (void)hipFree(j->covered);
    (void)hipFree(j->rowdof);
    (void)hipFree(j->ridx);
    (void)hipFree(j->res);
    (void)hipFree(j->gpos);
    (void)hipFree(j->glen);
    (void)hipFree(j->ginv);
    (void)hipFree(j->xt);
    (void)hipFree(j->yt);
    delete j;
  });
}

int nss_bjac_apply_f64(nss_bjac_t j, double alpha, const double* x, double beta, double* y, nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(j != nullptr, "bjac_apply: NULL handle");
    NSS_REQUIRE(x != y, "bjac_apply: x must not alias y");
    bjac_apply(*j, alpha, x, beta, y, nullptr, as_stream(stream));
  });
}

static void set_colors_common(nss_bjac_t j, nss_csr_t a_perm, int32_t ncolors, const int32_t* h_color_ptr,
                              const int32_t* h_color_rowptr, const int32_t* h_rowdof, const int32_t* h_ridx,
                              bool permuted_columns) {
  {
    NSS_REQUIRE(j && a_perm && h_color_ptr && h_color_rowptr && h_rowdof && h_ridx, "bjac_set_colors: NULL argument");
    NSS_REQUIRE(permuted_columns || a_perm->n == j->n, "bjac_set_colors: permuted matrix has the wrong column count");
    NSS_REQUIRE(ncolors >= 1, "bjac_set_colors: need at least one colour");
    NSS_REQUIRE(h_color_ptr[0] == 0 && h_color_ptr[ncolors] == j->nblocks, "bjac_set_colors: colour offsets must span the blocks");
    NSS_REQUIRE(h_color_rowptr[0] == 0 && h_color_rowptr[ncolors] == a_perm->m, "bjac_set_colors: colour row offsets must span the permuted rows");
    for (int c = 0; c < ncolors; ++c)
      NSS_REQUIRE(h_color_ptr[c + 1] >= h_color_ptr[c] && h_color_rowptr[c + 1] >= h_color_rowptr[c],
                  "bjac_set_colors: colour offsets not monotone");
    for (int32_t r = 0; r < a_perm->m; ++r)
      NSS_REQUIRE(h_rowdof[r] >= 0 && h_rowdof[r] < j->n, "bjac_set_colors: rowdof out of range");
    for (int64_t i = 0; i < int64_t(j->bs) * j->nblocks; ++i)
      NSS_REQUIRE(h_ridx[i] >= -1 && h_ridx[i] < a_perm->m, "bjac_set_colors: ridx out of range");
    // colour row offsets -> row-block offsets of the permuted matrix's launch plan
    std::vector<int32_t> rb(size_t(a_perm->nblk) + 1);
    NSS_HIP(hipMemcpy(rb.data(), a_perm->rowblk, sizeof(int32_t) * rb.size(), hipMemcpyDeviceToHost));
    std::vector<int32_t> crb(size_t(ncolors) + 1);
    size_t pos = 0;
    for (int c = 0; c <= ncolors; ++c) {
      while (pos < rb.size() && rb[pos] < h_color_rowptr[c]) ++pos;
      NSS_REQUIRE(pos < rb.size() && rb[pos] == h_color_rowptr[c],
                  "bjac_set_colors: a row block of the permuted matrix spans two colours (create it with cuts)");
      crb[c] = int32_t(pos);
    }
    (void)hipFree(j->rowdof);
    (void)hipFree(j->ridx);
    (void)hipFree(j->res);
    j->rowdof = j->ridx = nullptr;
    j->res = nullptr;
    NSS_HIP(hipMalloc(&j->rowdof, sizeof(int32_t) * std::max<size_t>(1, a_perm->m)));
    NSS_HIP(hipMalloc(&j->ridx, sizeof(int32_t) * size_t(j->bs) * j->nblocks));
    NSS_HIP(hipMalloc(&j->res, sizeof(double) * std::max<size_t>(1, a_perm->m)));
    NSS_HIP(hipMemcpy(j->rowdof, h_rowdof, sizeof(int32_t) * a_perm->m, hipMemcpyHostToDevice));
    NSS_HIP(hipMemcpy(j->ridx, h_ridx, sizeof(int32_t) * size_t(j->bs) * j->nblocks, hipMemcpyHostToDevice));
    j->gs_mat = a_perm;
    j->gs_permuted = false;
    j->color_ptr.assign(h_color_ptr, h_color_ptr + ncolors + 1);
    j->color_rowblk = crb;
    j->color_row.assign(h_color_rowptr, h_color_rowptr + ncolors + 1);
  }
}

int nss_bjac_set_colors(nss_bjac_t j, nss_csr_t a_perm, int32_t ncolors, const int32_t* h_color_ptr,
                        const int32_t* h_color_rowptr, const int32_t* h_rowdof, const int32_t* h_ridx) {
  return guarded([&] { set_colors_common(j, a_perm, ncolors, h_color_ptr, h_color_rowptr, h_rowdof, h_ridx, false); });
}

int nss_bjac_set_colors_permuted(nss_bjac_t j, nss_csr_t a_perm, int32_t ncolors, const int32_t* h_color_ptr,
                                 const int32_t* h_color_rowptr, const int32_t* h_rowdof, const int32_t* h_ridx) {
  return guarded([&] {
    NSS_REQUIRE(j && a_perm && h_color_ptr && h_color_rowptr && h_rowdof && h_ridx, "bjac_set_colors_permuted: NULL argument");
    const int32_t n_perm = a_perm->m;
    NSS_REQUIRE(a_perm->n == n_perm + 1, "bjac_set_colors_permuted: the permuted matrix must have n_perm + 1 columns (the "
                                         "last one stands for the dofs outside every block)");
    NSS_REQUIRE(int64_t(n_perm) + j->n_uncovered == j->n, "bjac_set_colors_permuted: the permuted rows must be exactly the dofs of the blocks");
    // the common part (offsets, rowdof, ridx, the residual buffer of the two-launch form)
    set_colors_common(j, a_perm, ncolors, h_color_ptr, h_color_rowptr, h_rowdof, h_ridx, true);
    // the launch plan must keep the blocks whole and the row blocks short
    std::vector<int32_t> rb(size_t(a_perm->nblk) + 1);
    NSS_HIP(hipMemcpy(rb.data(), a_perm->rowblk, sizeof(int32_t) * rb.size(), hipMemcpyDeviceToHost));
    std::vector<uint8_t> starts(size_t(n_perm) + 1, 0);
    {   // first row of every block
      std::vector<int32_t> first(size_t(j->nblocks), INT32_MAX);
      for (int c = 0; c < j->bs; ++c)
        for (int32_t b = 0; b < j->nblocks; ++b) {
          const int32_t r = h_ridx[size_t(c) * j->nblocks + b];
          if (r >= 0) first[size_t(b)] = std::min(first[size_t(b)], r);
        }
      for (int32_t b = 0; b < j->nblocks; ++b)
        if (first[size_t(b)] != INT32_MAX) starts[size_t(first[size_t(b)])] = 1;
      starts[size_t(n_perm)] = 1;
    }
    for (size_t k = 0; k + 1 < rb.size(); ++k) {
      NSS_REQUIRE(rb[k + 1] - rb[k] <= kGsRows, "bjac_set_colors_permuted: a row block of the permuted matrix has more than 256 rows");
      NSS_REQUIRE(starts[size_t(rb[k])] == 1, "bjac_set_colors_permuted: a row block of the permuted matrix splits a Gauss-Seidel block");
    }
    for (void* p : {(void*)j->gpos, (void*)j->glen, (void*)j->ginv, (void*)j->xt, (void*)j->yt}) (void)hipFree(p);
    j->gpos = j->glen = nullptr;
    j->ginv = j->xt = j->yt = nullptr;
    j->gs_permuted = false;
    NSS_HIP(hipMalloc(&j->gpos, std::max<size_t>(1, n_perm)));
    NSS_HIP(hipMalloc(&j->glen, std::max<size_t>(1, n_perm)));
    NSS_HIP(hipMalloc(&j->ginv, sizeof(double) * size_t(j->bs) * std::max<size_t>(1, n_perm)));
    NSS_HIP(hipMalloc(&j->xt, sizeof(double) * (size_t(n_perm) + 1)));
    NSS_HIP(hipMalloc(&j->yt, sizeof(double) * (size_t(n_perm) + 1)));
    NSS_HIP(hipMemset(j->xt, 0, sizeof(double) * (size_t(n_perm) + 1)));
    NSS_HIP(hipMemset(j->yt, 0, sizeof(double) * (size_t(n_perm) + 1)));
    hipLaunchKernelGGL(gs_pack_inverse_kernel, dim3((j->nblocks + kBlock - 1) / kBlock), dim3(kBlock), 0, nullptr, j->bs,
                       j->nblocks, n_perm, j->ridx, j->inv, j->ginv, j->gpos, j->glen);
    NSS_CHECK_LAUNCH();
    NSS_HIP(hipDeviceSynchronize());
    j->n_perm = n_perm;
    j->gs_permuted = true;
  });
}

int nss_bjac_smooth_f64(nss_bjac_t j, double xscale, const double* x, double* y, int32_t backward,
                        nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(j != nullptr, "bjac_smooth: NULL handle");
    NSS_REQUIRE(x != y, "bjac_smooth: x must not alias y");
    bjac_smooth(*j, xscale, x, y, backward != 0, nullptr, as_stream(stream));
  });
}

int nss_bjac_symgs_apply_f64(nss_bjac_t j, double xscale, const double* x, double* y, nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(j != nullptr, "bjac_symgs_apply: NULL handle");
    NSS_REQUIRE(x != y, "bjac_symgs_apply: x must not alias y");
    bjac_symgs_apply(*j, xscale, x, y, nullptr, as_stream(stream));
  });
}

int nss_bjac_info(nss_bjac_t j, int32_t* bs, int32_t* nblocks, int64_t* n, int64_t* algorithmic_bytes) {
  return guarded([&] {
    NSS_REQUIRE(j != nullptr, "bjac_info: NULL handle");
    if (bs) *bs = j->bs;
    if (nblocks) *nblocks = j->nblocks;
    if (n) *n = j->n;
    if (algorithmic_bytes) *algorithmic_bytes = 8 * int64_t(j->nblocks) * j->bs * j->bs + 16 * j->n;
  });
}

}  // extern "C"
