// Block-Jacobi handle shared by the preconditioner entry points and the fused loops.
#pragma once

#include "csr_stream.h"

#include <vector>

struct nss_bjac_s {
  uint64_t serial = 0;         // unique per handle (matrices planned around its blocks remember it: nss_csr_s::jb_serial)
  int32_t bs = 0, nblocks = 0;
  int64_t n = 0;
  int32_t* idx = nullptr;      // [bs][nblocks], -1 = padding
  int32_t* run = nullptr;      // [nblocks]: first dof * 32 + length, when every block is a run of
                               // consecutive dofs (4 bytes per block instead of 4 per dof)
  double* inv = nullptr;       // [bs*bs][nblocks]
  double* inv_sym = nullptr;   // [bs*(bs+1)/2][nblocks]: upper triangles, when every inverse block is
                               // symmetric (A symmetric): the apply kernel then reads ~half the bytes
  int32_t* covered = nullptr;  // dofs that belong to no block (count: n_uncovered)
  int32_t n_uncovered = 0;
  // multicolour Gauss-Seidel mode (nss_bjac_set_colors): blocks are stored colour-major
  const nss_csr_s* gs_mat = nullptr;      // rows of A permuted block by block, colour-major
  std::vector<int32_t> color_ptr;         // ncolors + 1 block offsets (host)
  std::vector<int32_t> color_rowblk;      // ncolors + 1 row-block offsets into gs_mat's launch plan (host)
  std::vector<int32_t> color_row;         // ncolors + 1 row offsets of the colours in the permuted numbering (host)
  int32_t* rowdof = nullptr;              // device: original dof of permuted row r
  int32_t* ridx = nullptr;                // device [bs][nblocks]: permuted row of a block entry, -1 = padding
  double* res = nullptr;                  // device: residual of the colour being swept (permuted rows)
  // Colour-major layout INSIDE the sweep (nss_bjac_set_colors_permuted): gs_mat is P A P^T -- rows and columns in the
  // colour-major block order, dofs that belong to no block map to the extra column n_perm (always 0) -- the iterate
  // and the right-hand side are gathered into that numbering once on entry (yt, xt) and the iterate is scattered back
  // once on exit; a colour is then ONE launch: the rows of the colour with the block solve in the epilogue (every
  // row block of gs_mat holds whole Gauss-Seidel blocks and at most kGsRows rows; the residuals of a row block pass
  // through LDS).  Everything a colour touches of its own is contiguous.
  bool gs_permuted = false;
  int32_t n_perm = 0;
  uint8_t* gpos = nullptr;                // [n_perm] position of the row inside its block
  uint8_t* glen = nullptr;                // [n_perm] rows of its block
  double* ginv = nullptr;                 // [bs][n_perm]: ginv[k][r] = (A_bb^-1)(row r, k-th row of the block)
  double *xt = nullptr, *yt = nullptr;    // [n_perm + 1]
};

namespace nss {

constexpr int kMaxBs = 16;
constexpr int kGsRows = 256;     // rows per row block of the permuted Gauss-Seidel matrix (their residuals: 2 KiB of LDS)

// y[dofs] = alpha * J x + beta * y[dofs]; returns immediately on the device when
// `done` (device int, may be NULL) is non-zero.
void bjac_apply(const nss_bjac_s& j, double alpha, const double* x, double beta, double* y, const int32_t* done,
                hipStream_t st);
// y = alpha * J x (dofs outside every block: 0) and, from the same registers, the per-workgroup
// partial sums of <y, x> into partials[0 .. bjac_dot_grid(j)); returns that count.  Block-Jacobi
// mode only.  Saves the separate dot pass (two vector reads and a launch) after the apply.
int bjac_dot_grid(const nss_bjac_s& j);
int bjac_apply_dot(const nss_bjac_s& j, double alpha, const double* x, double* y, double* partials, const int32_t* done,
                   hipStream_t st);


// one multicolour block Gauss-Seidel sweep / the symmetric pair as an operator (y = 0 first)
// flags (colour-major layout; ignored by the row-permuted one):
//   kGsFromZero  y is taken to be 0 on entry and need not hold zeros: it is not gathered, and the first colour of the
//                sweep -- whose rows see A y = 0 -- is y_c = D_c^-1 (xscale x_c) without a pass over its rows of A
//   kGsKeepX     x is the vector of the previous call on this handle (its permuted copy is still there): not gathered again
enum { kGsFromZero = 1, kGsKeepX = 2 };
void bjac_smooth(const nss_bjac_s& j, double xscale, const double* x, double* y, bool backward, const int32_t* done,
                 hipStream_t st, int flags = 0);
void bjac_symgs_apply(const nss_bjac_s& j, double xscale, const double* x, double* y, const int32_t* done,
                      hipStream_t st);

}  // namespace nss
