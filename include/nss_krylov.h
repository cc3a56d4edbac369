/* nss_krylov.h -- C ABI of libnsskrylov.so: the MI355X (gfx950) engine behind the
 * Stokes / SIMPLE saddle-point Krylov path of matschiner/navier-stokes-solver.
 *
 * The reference has no FFI of its own: its hot path sits behind the Python-level
 * NGSolve linear-algebra protocol.  Each entry point below
 * names the reference operation(s) it replaces (file:line under /root/reference).
 * The host side (navier-stokes-solver_amd/hipla, Python) binds these with ctypes;
 * INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - every function returns int: 0 = ok, non-zero = error; nss_last_error() gives
 *    the message of the last failure on the calling thread.  No C++ exception
 *    crosses the boundary.
 *  - `double*` / `int32_t*` vector arguments are DEVICE pointers owned by the caller
 *    (the Python host allocates them as torch tensors so torch.distributed / RCCL can
 *    exchange them); `h_`-prefixed arguments are HOST pointers, copied during the
 *    call.  Matrix / preconditioner handles are owned by the library (hipMalloc).
 *  - all arithmetic is fp64, indices int32.
 *  - calls are asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *    legacy default stream) unless the name ends in `_host` or says otherwise.
 *  - one process drives one GPU; handles are not thread-safe.
 */
#ifndef NSS_KRYLOV_H
#define NSS_KRYLOV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSS_API __attribute__((visibility("default")))
#define NSS_ABI_VERSION 1

typedef void* nss_stream_t;            /* hipStream_t */
typedef struct nss_csr_s* nss_csr_t;   /* CSR matrix resident in HBM */
typedef struct nss_bjac_s* nss_bjac_t; /* block-Jacobi inverse blocks resident in HBM */

/* ---- library / device ---------------------------------------------------- */
NSS_API int nss_abi_version(void);
NSS_API const char* nss_last_error(void);
/* name_cap bytes at `name` receive the gcnArchName; any out pointer may be NULL */
NSS_API int nss_device_info(int32_t* cu_count, int64_t* hbm_bytes, int32_t* wavefront_size,
                            char* name, int32_t name_cap);
NSS_API int nss_stream_synchronize(nss_stream_t stream);

/* ---- BLAS-1 (reference: every `x.data = ...`, `x *= s`, `v[:] = c` statement of
 * minres.py:62-118, bramble_pasciak_cg.py:88-141, solvers/bramblepasciak_new.py:130-241) */
NSS_API int nss_fill_f64(int64_t n, double value, double* x, nss_stream_t stream);
NSS_API int nss_copy_f64(int64_t n, const double* x, double* y, nss_stream_t stream);
NSS_API int nss_scal_f64(int64_t n, double a, double* x, nss_stream_t stream);
/* y_i = 1 / x_i (inverse diagonals of the AMG levels; IEEE division) */
NSS_API int nss_reciprocal_f64(int64_t n, const double* x, double* y, nss_stream_t stream);
/* y = sum_{i<nterms} h_coeff[i] * x_i, 1 <= nterms <= 4, evaluated left to right;
 * y may alias any x_i (element-wise).  h_x is a HOST array of device pointers. */
NSS_API int nss_lincomb_f64(int64_t n, int32_t nterms, const double* h_coeff,
                            const double* const* h_x, double* y, nss_stream_t stream);
/* InnerProduct (minres.py:71,98,103; bramble_pasciak_cg.py:105,130,137;
 * bramblepasciak_new.py:185,222,235).  Deterministic two-stage reduction (fixed grid,
 * fixed tree, no atomics).  npairs <= 4 pairs are summed pair 0 first (block vectors).
 * `result_dev` (device, 1 double) is written asynchronously. */
NSS_API int nss_dot_f64(int32_t npairs, const int64_t* h_n, const double* const* h_x,
                        const double* const* h_y, double* result_dev, nss_stream_t stream);
/* same, then waits and returns the value on the host */
NSS_API int nss_dot_host_f64(int32_t npairs, const int64_t* h_n, const double* const* h_x,
                             const double* const* h_y, double* h_result, nss_stream_t stream);
/* dst[i] = src[idx[i]], i < n: packs the entries a neighbouring slab needs into a contiguous
 * send buffer (multi-GPU halo exchange; no reference counterpart -- the
 * reference is single-process).  idx is a DEVICE int32 array. */
NSS_API int nss_gather_f64(int64_t n, const int32_t* idx, const double* src, double* dst,
                           nss_stream_t stream);
/* flux[i] = adv[i] * avg[i] - 1/2 |adv[i]| * diff[i]: the donor-cell (upwind) numerical flux of the explicit
 * convection term of the IMEX step (templates/NavierStokesSIMPLE_iterative.py:106-113,427-431: `conv_operator *
 * gfu`), between the SpMVs that form adv / avg / diff and the one that takes the divergence. */
NSS_API int nss_upwind_flux_f64(int64_t n, const double* adv, const double* avg, const double* diff, double* flux,
                                nss_stream_t stream);
/* z = x + a*y : STREAM-triad, the roofline denominator measured in the same run
 *  */
NSS_API int nss_stream_triad_f64(int64_t n, double a, const double* x, const double* y,
                                 double* z, nss_stream_t stream);

/* ---- CSR SpMV (reference: every `mat * vec` on blfA.mat / blfB.mat / B^T:
 * bramblepasciak_new.py:130,133,160,163,166,202,206,210,213; bramble_pasciak_cg.py:46-47,
 * 98-101,125,127; minres.py:66,97) ---------------------------------------------------- */
NSS_API int nss_csr_create(int32_t nrows, int32_t ncols, int64_t nnz, const int32_t* h_rowptr,
                           const int32_t* h_col, const double* h_val, nss_csr_t* out);
/* same, with row positions (ascending, h_cuts[i] in [0, nrows]) that no row block of the launch
 * plan may span: used where a launch covers only a sub-range of rows (colour by colour) */
NSS_API int nss_csr_create_cuts(int32_t nrows, int32_t ncols, int64_t nnz, const int32_t* h_rowptr,
                                const int32_t* h_col, const double* h_val, int32_t ncuts,
                                const int32_t* h_cuts, nss_csr_t* out);
/* explicit transpose as a new CSR handle (`matB.CreateTranspose()`, solvers/bramblepasciak_new.py:198;
 * `b.T`, run.py:45).  Stable radix sort of the entries by column on the device, entries of a row of
 * the result ordered by column; set-up, not on the iteration path. */
NSS_API int nss_csr_transpose(nss_csr_t a, nss_csr_t* out);
/* C = X Y as a new CSR handle (Galerkin products of the AMG set-up).  Expand / stable sort / compress
 * on the device: no atomics, every c_ij is the sum of x_ik * y_kj in ascending k, each product and
 * each addition rounded once (no FMA) -- bit-identical to a row-wise CPU product.  Rows are processed
 * in passes of at most max_products_per_pass products (<= 0: library default, 2^27). */
NSS_API int nss_csr_spgemm(nss_csr_t x, nss_csr_t y, int64_t max_products_per_pass, nss_csr_t* out,
                           nss_stream_t stream);
/* bytes per column index the SpMV kernel streams: 2 when the columns of every row block of the launch
 * plan fall into at most 16 aligned windows of 4096 columns (stored as 4 bits of window number + 12
 * bits of offset, decoded through 16 per-block window bases), else 4.  The values and the order of
 * the products are the same either way. */
NSS_API int nss_csr_index_width(nss_csr_t a, int32_t* bytes);
/* ---- multicolour ordering of the block Gauss-Seidel sweep, on the device (scope row N1) -------------
 * nss_csr_ones_like: the pattern of `a` with every value 1 (products of patterns cannot cancel).
 * nss_graph_color: colours (DEVICE int32[n] out) such that no two neighbours share one; every
 *   colour is a maximal independent set of the still uncoloured nodes (Luby rounds with the
 *   caller's distinct positive priorities, DEVICE int64[n]).  Neighbours = off-diagonal entries of
 *   `g` and, when given, of `g_transposed` (structurally non-symmetric graphs).  Integer state only:
 *   identical to hipla/coloring.py::color_blocks with the same priorities.
 * nss_csr_select_rows: new matrix whose row r is row d_rows[r] of `a` (DEVICE int32[nrows]); the
 *   launch plan does not span the ascending HOST row positions h_cuts (colour boundaries). */
NSS_API int nss_csr_ones_like(nss_csr_t a, nss_csr_t* out, nss_stream_t stream);
NSS_API int nss_graph_color(nss_csr_t g, nss_csr_t g_transposed, const int64_t* d_priority, int32_t* d_colors,
                            int32_t* ncolors_out, nss_stream_t stream);
NSS_API int nss_csr_select_rows(nss_csr_t a, int32_t nrows, const int32_t* d_rows, int32_t ncuts,
                                const int32_t* h_cuts, nss_csr_t* out, nss_stream_t stream);
/* nss_csr_permute: as nss_csr_select_rows, and every column c becomes d_colmap[c] (DEVICE int32[columns of a]; the
 *   result has ncols_out columns; the entries of a row keep their order); the launch plan additionally keeps at most
 *   `max_rows` rows per row block (0: no limit) and starts row blocks only at rows with h_row_pos[r] == 0 (HOST
 *   uint8[nrows] or NULL): rows of one Gauss-Seidel block stay in one workgroup.
 * nss_graph_color_greedy: first-fit colouring in node order on the HOST (sequential; h_colors: HOST int32[n] out).
 *   On grid-like block graphs it finds the parity colouring (2 - 4 balanced colours) where the Luby rounds of
 *   nss_graph_color give 5 - 6 with a tail of tiny ones -- each colour is a launch of the sweep. */
NSS_API int nss_csr_permute(nss_csr_t a, int32_t nrows, const int32_t* d_rows, const int32_t* d_colmap, int32_t ncols_out,
                            int32_t ncuts, const int32_t* h_cuts, int32_t max_rows, const uint8_t* h_row_pos,
                            nss_csr_t* out, nss_stream_t stream);
NSS_API int nss_graph_color_greedy(nss_csr_t g, nss_csr_t g_transposed, int32_t* h_colors, int32_t* ncolors_out);
/* the set-up entry points (nss_csr_spgemm, nss_csr_transpose, nss_amg_*) keep their multi-GB
 * temporaries in a pool between calls; this returns the unused ones to the driver */
NSS_API int nss_scratch_trim(void);
/* Streaming (non-temporal) loads of the vector operands the fused loops do not read again before they are
 * rewritten: -1 = automatic (from 20 MB per vector on: they pay when the vectors do not fit the caches and cost
 * when they do), 0 = never, 1 = always.  Same bits either way. */
NSS_API int nss_stream_loads_mode(int32_t mode);
/* copy the CSR arrays back to HOST buffers (rows+1 / nnz / nnz entries; sizes from nss_csr_info) */
NSS_API int nss_csr_download(nss_csr_t a, int32_t* h_rowptr, int32_t* h_col, double* h_val);
NSS_API int nss_csr_destroy(nss_csr_t a);
/* y = alpha * A x + beta * y   (beta == 0: y is not read).  x must not alias y. */
NSS_API int nss_csr_spmv_f64(nss_csr_t a, double alpha, const double* x, double beta, double* y,
                             nss_stream_t stream);
/* entries that share one stored column index: 1 for plain CSR streams; g > 1 when the matrix consists of
 * aligned runs of g consecutive columns (block-structured operators, e.g. facet blocks of 5 / 12 dofs) and
 * the SpMV kernel streams one 16-bit index per run -- 8 + 2/g bytes per non-zero. */
NSS_API int nss_csr_index_group(nss_csr_t a, int32_t* entries_per_index);
/* how the SpMV kernels reach the operand x: 0 = 4-byte columns, gather; 1 = 16-bit window-relative columns,
 * gather; 2 = staged -- per row block the runs of consecutive columns it touches are copied to LDS with
 * LDS-DMA issued ahead of the matrix stream, and the 16-bit index is a position in that copy (taken when every
 * row block touches at most 13 runs / `chunk` columns; kernels whose operand is an expression of two vectors
 * use form 1 of the same matrix); 3 = fixed-width copy of a large matrix with at most two entries per row
 * (B^T of the staggered grid), multiplied by the row-per-lane kernel. */
NSS_API int nss_csr_operand_form(nss_csr_t a, int32_t* form);
/* Kernels whose SpMV operand is an expression of TWO stored vectors (the rows of B multiply t1 - s0, the rows of B^T
 * beta s1 + w1: solvers/bramblepasciak_new.py:212-213, :206 with :240-241 folded in) can take both from LDS copies
 * ("pair-staged") when the operand runs of every row block hold at most half the LDS buffer.  This re-plans `a` in
 * place with shorter row blocks (1024 products) when that gets it there -- per-row sums keep their bits, the
 * dot-product partials are regrouped -- and reports whether the matrix is pair-stageable now.  Set-up only
 * (synchronises the device); a no-op for matrices that are not staged at all or already qualify. */
/* Re-plan `a` (in place, set-up only) so that every row block holds whole blocks of `j` and at most 512 rows, and record
 * the first Jacobi block of every row block: a kernel over the rows of `a` can then apply `j` to its own result in the
 * epilogue (nss_bpcg2_fuse_block_jacobi).  Needs symmetric inverse blocks over runs of consecutive dofs that tile the
 * rows of `a` (in any numbering of the blocks), and a system the size rule of nss_bpcg2_fuse_block_jacobi wants fused;
 * *planned = 0 (and nothing changes) otherwise.  Per-row sums keep their
 * bits: only the partition of the rows into workgroups changes. */
NSS_API int nss_csr_plan_for_blocks(nss_csr_t a, nss_bjac_t j, int32_t* planned);
NSS_API int nss_csr_plan_for_pairs(nss_csr_t a, int32_t* pair_staged);
/* the same question without re-planning */
NSS_API int nss_csr_pair_staged(nss_csr_t a, int32_t* pair_staged);
/* process-wide override (tests, measurements): 0 = kernels never take the pair-staged form (they gather through the
 * window form of the same matrix: same bits), -1 / 1 = wherever a matrix admits it */
NSS_API int nss_csr_pair_mode(int32_t mode);
/* Reuse-aware dispatch order of the staged kernels (grid operators beyond ~1e7 rows; csrc/csr_stream.h: blkdisp): a
 * matrix whose row blocks re-read operand runs `period` blocks further down the natural order (the next grid plane)
 * gets a second descriptor table that visits the blocks b, b + P, ..., b + (T - 1) P of T planes back to back.
 * Only the workgroup -> row block map changes; all bits stay.  nss_csr_dispatch_mode applies to matrices created
 * afterwards: planes = -2 library default (OFF: measured at 5e7 DoF the tiled walk cuts the bytes fetched through the
 * L2s from 1.18 x to 1.05 x of the algorithmic reads and is not faster, profiles/r03_ab_dispatch_cfg5.txt), -1
 * automatic (T = 8 from a period of `min_period` row blocks on; min_period <= 0: 256), 0 never, 1 .. 64 = always with
 * T = planes; `run` (<= 0: library default) = consecutive row blocks
 * of one plane before the walk moves on to the next plane.  nss_csr_dispatch_info reports what a matrix got
 * (period 0: natural order). */
NSS_API int nss_csr_dispatch_mode(int32_t planes, int32_t min_period, int32_t run);
NSS_API int nss_csr_dispatch_info(nss_csr_t a, double* period, int32_t* planes);
/* matrices created from now on take form 3 from `min_rows` rows on (default 2^21: below, the iteration is
 * launch-bound and the launch a pair of matrices shares is worth more); -1 restores the default.  Same bits
 * either way. */
NSS_API int nss_csr_direct_rows_threshold(int64_t min_rows);
/* shape, nnz, launch plan (row blocks, lanes per row) and algorithmic bytes of one SpMV AS THIS MATRIX IS
 * STORED: value stream 8*nnz + column stream (2*nnz/g with a 16-bit form and one index per g entries, 4*nnz
 * without; + the per-row-block table: 128 B staged / 64 B windows) + 4*(rows+1) + 8*cols + 8*rows; the
 * fixed-width copy (form 3): 24*rows + 8*cols + 8*rows.  (SURVEY.md section 8d states the fp64 / int32 CSR
 * figure, 12*nnz + ...; pricing a launch that streams fewer bytes at that figure would overstate its rate.) */
NSS_API int nss_csr_info(nss_csr_t a, int32_t* nrows, int32_t* ncols, int64_t* nnz,
                         int32_t* nblocks, int32_t* lanes_per_row, int64_t* algorithmic_bytes);
NSS_API int nss_csr_diagonal(nss_csr_t a, double* diag_dev, nss_stream_t stream);
/* copy the launch plan to the host: h_out receives nblocks+1 first-row indices (cap = capacity) */
NSS_API int nss_csr_row_blocks(nss_csr_t a, int32_t* h_out, int64_t cap);

/* ---- preconditioner applies -------------------------------------------------------------
 * point Jacobi / lumped-mass inverse `Preconditioner(m,'local')`
 * (templates/NavierStokesSIMPLE_iterative.py:197-200, run.py:62): y = alpha*d.*x + beta*y */
NSS_API int nss_diag_apply_f64(int64_t n, const double* d, double alpha, const double* x,
                               double beta, double* y, nss_stream_t stream);
/* additive block Jacobi J = sum_b E_b A_bb^-1 E_b^T, the operator form of
 * `a.mat.CreateBlockSmoother(blocks)` (templates/NavierStokesSIMPLE_iterative.py:360-373,383).
 * h_idx: int32[bs][nblocks] (block-interleaved), -1 = padding; blocks must be disjoint.
 * The bs x bs blocks are gathered from `a` and inverted on the device (Gauss-Jordan with
 * partial pivoting), stored interleaved: inv[(r*bs+c)*nblocks + b].  1 <= bs <= 16. */
NSS_API int nss_bjac_create(nss_csr_t a, int32_t bs, int32_t nblocks, const int32_t* h_idx,
                            nss_bjac_t* out);
NSS_API int nss_bjac_destroy(nss_bjac_t j);
/* y[dofs] = alpha * J x + beta * y[dofs]; dofs in no block: y = beta*y (0 if beta == 0) */
NSS_API int nss_bjac_apply_f64(nss_bjac_t j, double alpha, const double* x, double beta, double* y,
                               nss_stream_t stream);
NSS_API int nss_bjac_info(nss_bjac_t j, int32_t* bs, int32_t* nblocks, int64_t* n, int64_t* algorithmic_bytes);
/* multiplicative block Gauss-Seidel sweeps `jacobi.Smooth(y, x)` / `jacobi.SmoothBack(y, x)`
 * (templates/NavierStokesSIMPLE_iterative.py:376-381) over a
 * MULTICOLOUR block ordering: the handle's blocks [h_color_ptr[c], h_color_ptr[c+1]) carry colour
 * c and must not be coupled to each other (the host colours the block graph and creates the
 * handle with its blocks in colour-major order).  All blocks of a colour update in parallel:
 * y_b += A_bb^-1 (xscale * x_b - (A y)_b).
 * To keep the sweep streaming, the host passes `a_perm` = the rows of A re-ordered block by
 * block in that colour-major order (columns unchanged; created with nss_csr_create_cuts so
 * that no row block spans two colours): the residual of a colour is then one CSR-stream SpMV
 * over a contiguous row range.  h_rowdof[r] = original dof of permuted row r;
 * h_ridx: int32[bs][nblocks] = permuted row of each block entry (-1 = padding). */
NSS_API int nss_bjac_set_colors(nss_bjac_t j, nss_csr_t a_perm, int32_t ncolors, const int32_t* h_color_ptr,
                                const int32_t* h_color_rowptr, const int32_t* h_rowdof, const int32_t* h_ridx);
/* The same sweeps with the colour-major numbering used INSIDE the sweep as well: `a_perm` = P A P^T from
 * nss_csr_permute (rows AND columns in the colour-major block order; n_perm rows = the dofs of the blocks, n_perm + 1
 * columns, the last standing for the dofs outside every block; every row block of its launch plan holds whole
 * Gauss-Seidel blocks and at most 256 rows).  A sweep call gathers x and y into that numbering once, runs ONE launch
 * per colour (rows of the colour with the block solve in the epilogue: the residuals of a row block pass through LDS)
 * and scatters y back once; the symmetric operator gathers / scatters once for both sweeps.  Same products in the
 * same order as the two-launch form: same bits. */
NSS_API int nss_bjac_set_colors_permuted(nss_bjac_t j, nss_csr_t a_perm, int32_t ncolors, const int32_t* h_color_ptr,
                                         const int32_t* h_color_rowptr, const int32_t* h_rowdof, const int32_t* h_ridx);
/* one sweep: colours ascending (backward == 0) or descending */
NSS_API int nss_bjac_smooth_f64(nss_bjac_t j, double xscale, const double* x, double* y, int32_t backward,
                                nss_stream_t stream);
/* symmetric sweep as an operator: y = 0; forward sweep; backward sweep */
NSS_API int nss_bjac_symgs_apply_f64(nss_bjac_t j, double xscale, const double* x, double* y,
                                     nss_stream_t stream);

/* ---- smoothed-aggregation AMG V(1,1)-cycle -------------------------------------------------
 * Counterpart of the 'h1amg' correction inside the reference's MypreA
 * (templates/NavierStokesSIMPLE_iterative.py:320-357,380,383).  The
 * hierarchy (A_l, P_l, R_l = P_l^T, inverse diagonals, dense inverse of the coarsest operator as
 * a CSR matrix) is built with the set-up entry points below; the cycle runs on the device with CSR-stream SpMVs:
 *   x = w D^-1 b;  r = b - A x;  x += P V(R r);  x += w D^-1 (b - A x).
 * The library allocates the per-level work vectors. */
/* Set-up on the device.
 * nss_amg_aggregate: aggregates of smoothed aggregation over the strength graph of A
 *   (j strong for i  <=>  j != i and |a_ij| >= theta * sqrt(|a_ii a_jj|); theta <= 0: every
 *   off-diagonal entry).  Roots = maximal independent set of the distance-2 graph (Luby rounds with
 *   the caller's distinct positive priorities, DEVICE int64[n]); roots are numbered in index order,
 *   unaggregated nodes join their first aggregated strong neighbour in four sweeps, the rest become
 *   singletons.  d_agg: DEVICE int64[n] out; *nagg_out: number of aggregates.  Integer state only:
 *   the result is identical to oracle/krylov_ref.py::sa_aggregate.
 * nss_amg_prolongator: P = (I - omega D^-1 A) T, T = piecewise-constant prolongator of d_agg. */
NSS_API int nss_amg_aggregate(nss_csr_t a, double theta, const int64_t* d_priority, int64_t* d_agg,
                              int64_t* nagg_out, nss_stream_t stream);
NSS_API int nss_amg_prolongator(nss_csr_t a, const int64_t* d_agg, int64_t nagg, double omega, nss_csr_t* out,
                                nss_stream_t stream);
typedef struct nss_amg_level_s {
  nss_csr_t A;          /* level operator (n x n)                                  */
  nss_csr_t P, R;       /* prolongation (n x n_coarse) and restriction; NULL on the coarsest level */
  const double* dinv;   /* DEVICE: inverse diagonal of A (n)                        */
} nss_amg_level_t;
typedef struct nss_amg_s* nss_amg_t;
NSS_API int nss_amg_create(int32_t nlevels, const nss_amg_level_t* h_levels, nss_csr_t coarse_inverse,
                           double omega, nss_amg_t* out);
/* Auxiliary-space preconditioner term  x -> T (sum_c E_c V_c E_c^T) T^T x  as an nss_amg_t: the
 * `transform @ preAh1 @ transform.T` of the reference's MypreA
 * (templates/NavierStokesSIMPLE_iterative.py:291,320-357,380,383).  T (rows: velocity dofs, columns: the
 * stacked per-component auxiliary spaces) and its explicit transpose TT; comps[c] = V-cycle handle of
 * component c's auxiliary operator (nss_amg_create), sizes adding up to the columns of T.  Accepted
 * wherever a V-cycle handle is (nss_amg_apply_f64, pre_amg of the fused loops). */
NSS_API int nss_amg_create_auxiliary(nss_csr_t T, nss_csr_t TT, int32_t ncomp, const nss_amg_t* h_comps,
                                     nss_amg_t* out);
/* Components that share ONE hierarchy handle (the same Laplacian and boundary conditions for every velocity component)
 * are cycled together: every level operator is read once for all right-hand sides (csrc/amg.hip: csr_multi_kernel,
 * 2 or 3 components).  on = 0 cycles them one after the other as round 2 did (tests, A/B runs); default 1. */
NSS_API int nss_amg_batch_components(int32_t on);
NSS_API int nss_amg_destroy(nss_amg_t a);
/* x = V(bscale * b);  b and x have the finest level's size and must not alias */
NSS_API int nss_amg_apply_f64(nss_amg_t a, double bscale, const double* b, double* x, nss_stream_t stream);

/* ---- fused Bramble-Pasciak CG, recurrence-optimised form ---------------------------------
 * Replaces the loop body of solvers/bramblepasciak_new.py:200-249 (the solver the SIMPLE
 * drivers call, templates/NavierStokesSIMPLE_iterative.py:397).  The host fills this struct
 * with plain device pointers after the set-up phase (:124-189) and enqueues iterations;
 * alpha / beta / wd / the stop test live in `scal` / `ctrl` on the device.
 *
 * Vector lengths: n_u for u0,d0,w0,s0,z0,q,t0,t2 ; t1 and t4 are SpMV operands of A and B and
 * have A's / B's column count (n_u plus halo entries in a row-partitioned run); n_p for
 * u1,d1,w1,t3 ; s1 is the operand of B^T (B^T's column count).
 * scal: double[16] = { wd (even it), as_s, wdn, alpha, beta, err0, tol, rel_err(0/1), wd (odd it),
 *                     local as_s, local wdn (row-partitioned runs) };
 * ctrl: int32[8]  = { done, it_final, last_it, breakdown, pending, -, -, - }; `pending` = it + 1 while
 * the velocity part of `u += alpha s` of iteration it waits for K1 of the next iteration (it reads s0
 * anyway); nss_bpcg2_poll applies and clears it, so read the solution after a poll.
 * hist: double[maxsteps]. */
typedef struct nss_bpcg2_s {
  nss_csr_t A, B, BT;          /* A: n_u rows; B: n_p rows; BT: n_u rows (explicit transpose, :198) */
  const double* pre_diag;      /* point-Jacobi preA (inverse diagonal, n_u)  -- or NULL          */
  nss_bjac_t pre_bjac;         /* block-Jacobi / block-GS preA               -- or NULL          */
  nss_amg_t pre_amg;           /* AMG V-cycle, added to the above (either may be NULL; at least
                                  one of the three is set): preA_unscaled = AMG + Jacobi part,
                                  the additive form of MypreA (:383)                             */
  const double* minv;          /* preM = Preconditioner(mass,'local'): inverse mass diagonal, n_p */
  double *u0, *u1, *d0, *d1, *w0, *w1, *s0, *s1, *z0, *q, *t0, *t1, *t2, *t3, *t4;
  double* scal;
  int32_t* ctrl;
  double* hist;
  double *partials_a, *partials_b, *partials_c; /* sizes: nss_bpcg2_workspace() */
  double k;                    /* scale factor: preA = k * preA_unscaled (:118-122) */
  int32_t n_u, n_p;
  /* statically condensed form (`blfA.condense`, :11-18, :84-103) -- all four NULL otherwise.  A is
   * then the explicit product (I - H^T)(S + A_ii)(I - H) and the preconditioner apply of K1 becomes
   * harmonic_extension(): f = t0 + H^T t0;  t1 = k preA_unscaled f;  t1 += H t1;  t1 += A_ii^-1 f.
   * The rows of H that hold entries (interior dofs) must be disjoint from its columns (coupling
   * dofs): `t1 += H t1` runs in place. */
  nss_csr_t cond_HT, cond_H, cond_inner;   /* harmonic_extension_trans, harmonic_extension, inner_solve */
  double* cond_f;                          /* n_u work vector */
  /* row-partitioned runs: ghost copies of s0 / w0 on the ghost columns of B's operand, updated
   * redundantly with the same recurrences as the owned entries (w0_g -= alpha t1_g in K4, s0_g =
   * beta s0_g + w0_g in K5), so that t4_g = t1_g - s0_g (end of K2) needs no halo exchange of its own:
   * two exchanges per iteration instead of three.  ghost_map[i] = position of ghost i in the t1 operand
   * buffer (whose ghost list must contain B's); the ghost tail of t4 follows its n_u owned entries.
   * ghost_mode == 0: off (t4 is exchanged). */
  int32_t ghost_mode, ghost_n;
  const int32_t* ghost_map;
  double *ghost_s0, *ghost_w0;
  /* the same for the pressure part, so that s1 needs no exchange either (one halo exchange per
   * iteration: t1): ghost_b = the rows of B of the ghost pressure cells of B^T's operand, with columns
   * in the layout of the t4 operand; every iteration t3_g = ghost_b t4 (one small SpMV before K4),
   * w1_g -= alpha minv_g t3_g (K4), s1_g = beta s1_g + w1_g (K5), where s1_g is the ghost tail of the
   * s1 operand buffer itself (s1 + n_p).  ghost_p_mode == 0: off (s1 is exchanged). */
  int32_t ghost_p_mode, ghost_p_n;
  nss_csr_t ghost_b;
  double *ghost_t3, *ghost_w1;
  const double* ghost_minv;
  /* row-partitioned runs: SUM1 / SUM2 leave the LOCAL sums in scal[9] / scal[10] and the caller
   * all-reduces them OUT OF PLACE into scal[1] (as_s) / scal[2] (wdn), so that the scalars stay frozen
   * once the stop flag is set (the sum kernels then return early and every further all-reduce
   * reproduces the same value).  0: the sums go straight to scal[1] / scal[2] (single GPU). */
  int32_t local_sums;
  /* row-partitioned runs: V-cycle with replicated coarse levels as (part of) preA, applied natively with its
   * two halo exchanges and one coarse all-reduce (nss_dist_amg_create); NULL otherwise */
  struct nss_dist_amg_s* pre_dist_amg;
  /* row-partitioned runs on the COMPACT plan (C1, preA, exchange of t1, C23, sum, all-reduce, C4, sum,
   * all-reduce: six launches and three collectives per iteration instead of nine launches).  The state is laid
   * out so that every ghost entry sits behind the owned entries of its vector:
   *   s0, w0, t1 : buffers [n_u owned | ghost_n ghosts] in the layout of A's operand (ghost_s0 = s0 + n_u,
   *                ghost_w0 = w0 + n_u; ghost_map = NULL: the ghost of t1 that belongs to ghost i is t1[n_u + i]);
   *   s1, w1, t3 : buffers [n_p owned | ghost_p_n ghosts] in the layout of B^T's operand (ghost_w1 = w1 + n_p,
   *                ghost_t3 = t3 + n_p, ghost_minv as above);
   *   B          : n_p + ghost_p_n rows -- the slab's pressure rows followed by the rows of the ghost pressure
   *                cells -- with columns in the layout of A's operand (it multiplies t1 - s0 formed on the fly);
   *                no row block of its launch plan spans row n_p; ghost_b and t4 are not used.
   * The ghost copies follow the recurrences of their owners: s0_g = beta s0_g + w0_g in C1, s1_g = beta s1_g +
   * w1_g and t3_g with the rows of B in C23 (ghost rows add nothing to the dot partials), w0_g and w1_g in C4.
   * 0: the eight-phase layout described above. */
  int32_t dist_compact;
  /* row-partitioned runs: the auxiliary-space term of MypreA on slabs (nss_dist_aux_create), alone, added to a (block)
   * Jacobi part (additive MypreA, :383) or -- pre_bjac in Gauss-Seidel mode -- inside the multiplicative MypreA
   * (:376-381) whose sweeps run inside the slab (additive across slabs) and whose residual uses the partitioned A;
   * NULL otherwise */
  struct nss_dist_aux_s* pre_dist_aux;
  /* row-partitioned runs on the compact plan: the mailbox transport (nss_p2p_create) -- the two all-reduces of an
   * iteration happen INSIDE the sum kernels (remote stores into the peers' mailboxes, a bounded spin on the own one,
   * the nranks values added in rank order) and the halo of t1 travels by a put kernel into the neighbours' landing
   * zones; no collective library on the critical path.  NULL: RCCL (nss_dist_t). */
  struct nss_p2p_s* p2p;
} nss_bpcg2_t;

enum {
  NSS_BPCG2_K1 = 1,    /* t0 = (q recurrence) + B^T s1 [, t1 = k dinv t0]; block-Jacobi: t1 = k J t0 */
  NSS_BPCG2_K2 = 2,    /* t2 = A t1, t4 = t1 - s0, partials <s0, t2 - t0>                             */
  NSS_BPCG2_K3 = 3,    /* t3 = B t4, partials <s1, t3>                                                */
  NSS_BPCG2_SUM1 = 4,  /* scal[as_s] = local sum of the K2/K3 partials                               */
  NSS_BPCG2_ALPHA = 5, /* no-op: alpha = wd / as_s is evaluated inside K4                              */
  NSS_BPCG2_K4 = 6,    /* alpha; u1 += a s1 (u0: deferred to K1 / poll), d -= a v, w -= a C^-1 v, <w, d> */
  NSS_BPCG2_SUM2 = 7,  /* scal[wdn] = local sum of the K4 partials                                   */
  NSS_BPCG2_BETA = 8,  /* no-op: folded into K5                                                      */
  NSS_BPCG2_K5 = 9     /* beta = wdn / wd, s1 = beta s1 + w1, hist[it] = sqrt|wd|, stop test          */
};

/* The single-GPU loop issues the same arithmetic in three dependent launches (+ the block-Jacobi apply):
 * the "compact plan" (csrc/bpcg2.hip).  Bit-identical to the eight-phase form above. */
enum {
  NSS_BPCG2C_C1 = 1,   /* books of iteration it-1 (K5) in every workgroup, then K1 with beta*s1 + w1 on the fly; preA */
  NSS_BPCG2C_C23 = 2,  /* rows of A (K2 without t4) and rows of B (K3 on t1 - s0, storing s1) in one launch  */
  NSS_BPCG2C_SUMA = 3, /* stand-alone sum of the C23 partials -- only when they are too many to fold into C4 */
  NSS_BPCG2C_C4 = 4,   /* K4 (one-shot launch), as_s summed in every workgroup when folded                   */
  NSS_BPCG2C_SUMW = 5  /* stand-alone sum of the C4 partials -- only when not folded into C1                 */
};

NSS_API int nss_bpcg2_workspace(const nss_bpcg2_t* s, int64_t* partials_a, int64_t* partials_b,
                                int64_t* partials_c);
/* one phase of iteration `it` (row-partitioned runs all-reduce scal[as_s] / scal[wdn] and exchange
 * halos between phases) */
NSS_API int nss_bpcg2_phase(const nss_bpcg2_t* s, int32_t which, int32_t it, nss_stream_t stream);
/* phases first..last (inclusive, in enum order) of iteration `it` in one call: the stretch
 * between two communication points of the row-partitioned loop */
NSS_API int nss_bpcg2_phases(const nss_bpcg2_t* s, int32_t first, int32_t last, int32_t it,
                             nss_stream_t stream);
/* enqueue iterations [it_begin, it_end) back to back (single GPU, compact plan): no host
 * synchronisation.  The books of iteration it_end - 1 (history entry, stop test) are done by the next
 * call's first kernel or by nss_bpcg2_poll, whichever comes first. */
NSS_API int nss_bpcg2_iterate(const nss_bpcg2_t* s, int32_t it_begin, int32_t it_end, nss_stream_t stream);
/* the same iterations in the eight-phase form (what the row-partitioned loop issues between its
 * collectives; kept callable on one GPU for cross-checks and measurements) */
NSS_API int nss_bpcg2_iterate_classic(const nss_bpcg2_t* s, int32_t it_begin, int32_t it_end,
                                      nss_stream_t stream);
/* phases first..last (NSS_BPCG2C_*, inclusive) of iteration `it` of the compact plan */
NSS_API int nss_bpcg2_cphases(const nss_bpcg2_t* s, int32_t first, int32_t last, int32_t it,
                              nss_stream_t stream);
/* 1 if the dot-product sums of this state are folded into the consuming kernels, 0 if the stand-alone
 * sum kernels run (more than 4096 partials).  nss_bpcg2_fold_mode: process-wide override for tests and
 * measurements: -1 automatic (default), 0 never fold, 1 always fold. */
NSS_API int nss_bpcg2_folds_sums(const nss_bpcg2_t* s, int32_t* folds);
NSS_API int nss_bpcg2_fold_mode(int32_t mode);
/* Block Jacobi alone as preA (templates/NavierStokesSIMPLE_iterative.py:360-373,383 with GS=False and no auxiliary
 * term): when B^T's row blocks are planned around the Jacobi blocks (nss_csr_plan_for_blocks) C1 applies
 * t1 = k J t0 in its epilogue -- the row block's t0 passes through LDS, one lane per Jacobi block, the arithmetic of
 * the stand-alone apply (identical bits) -- instead of a launch of its own that reads t0 back: 3 dependent launches
 * per iteration.  By size (mode -1, the default: up to 2^18 velocity rows -- the launch-bound regime, where it is
 * worth 8-25 % of an iteration; above 1e6 DoF the stand-alone apply is 2-3 % faster); 0 never, 1 whenever B^T is
 * planned for it (tests, A/B runs).  nss_bpcg2_c1_applies_preA: what the next C1 of this state will do. */
NSS_API int nss_bpcg2_fuse_block_jacobi(int32_t mode);
NSS_API int nss_bpcg2_c1_applies_preA(const nss_bpcg2_t* s, int32_t* yes);
/* wait for the stream and read ctrl: done (0 running, 1 stop test fired, 2 breakdown
 * <s, K s> == 0 where the reference raises ZeroDivisionError, :226), iteration at which it
 * happened, last iteration whose history entry was written */
NSS_API int nss_bpcg2_poll(const nss_bpcg2_t* s, int32_t* done, int32_t* it_final, int32_t* last_it,
                           nss_stream_t stream);

/* ---- row-partitioned BPCG iteration with RCCL issued natively --------------------------------
 * (multi-GPU; the reference is single-process, so no counterpart.)
 * One nss_halo_t per SpMV operand of the loop (s1 for B^T, t1 for A, t4 for B): the owned
 * entries other ranks need are packed (send_idx -> sendbuf), exchanged with grouped
 * ncclSend / ncclRecv between slab neighbours, and land behind the owned entries of `ext`.
 * [int_begin, int_end) are the row blocks of the consuming matrix that touch no ghost column:
 * with overlap != 0 they run while the exchange is in flight on a second stream, the remaining
 * (boundary) row blocks after it.  Inner products are all-reduced in place in `scal`. */
typedef struct nss_halo_s {
  const int32_t* send_idx;     /* DEVICE int32[n_pack]: owned indices to pack                    */
  double* sendbuf;             /* DEVICE double[n_pack]                                          */
  double* ext;                 /* DEVICE operand buffer [owned | ghosts]                         */
  const int32_t* h_send_peer;  /* HOST  int32[n_send]                                            */
  const int64_t* h_send_off;   /* HOST  int64[n_send]: offsets into sendbuf (direct: into ext)   */
  const int64_t* h_send_cnt;   /* HOST  int64[n_send]                                            */
  const int32_t* h_recv_peer;  /* HOST  int32[n_recv]                                            */
  const int64_t* h_recv_off;   /* HOST  int64[n_recv]: offsets into ext (>= n_owned)             */
  const int64_t* h_recv_cnt;   /* HOST  int64[n_recv]                                            */
  int32_t n_pack, n_send, n_recv;
  int32_t int_begin, int_end;  /* interior row blocks of the consuming matrix                   */
  int32_t direct;              /* 1: every destination gets one contiguous run of the owned entries,
                                  sent straight out of ext (n_pack = 0, no pack kernel)          */
} nss_halo_t;

typedef struct nss_dist_s* nss_dist_t;
/* `nccl_comm` is an initialised ncclComm_t (created by the host, e.g. through ctypes on librccl);
 * the library resolves ncclAllReduce / ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd from
 * the librccl already loaded in the process and creates its communication stream and events. */
NSS_API int nss_dist_create(void* nccl_comm, int32_t nranks, int32_t rank, nss_dist_t* out);
NSS_API int nss_dist_destroy(nss_dist_t d);
/* iterations [it_begin, it_end) of the partitioned loop: per iteration 3 halo exchanges, the
 * phases of nss_bpcg2_phase split into interior / boundary row blocks, 2 all-reduces.
 * overlap: 0 = exchange then multiply on `stream`; 1 = interior rows overlap the exchange;
 * 2 = as 1 but also when this rank has no neighbour (exercises the split path in tests).
 * A state with dist_compact != 0 runs the COMPACT plan instead -- per iteration C1 (+ preA), the exchange of t1
 * (halo_t1; halo_s1 / halo_t4 may be NULL, `overlap` is ignored), C23, sum, all-reduce, C4, sum, all-reduce: six
 * launches and three collectives; the books of an iteration are done by C1 of the next one from the all-reduced
 * <w, d> (or by nss_bpcg2_poll).  Same arithmetic per lane as the eight-phase form. */
NSS_API int nss_bpcg2_iterate_dist(const nss_bpcg2_t* s, nss_dist_t d, const nss_halo_t* halo_s1,
                                   const nss_halo_t* halo_t1, const nss_halo_t* halo_t4, int32_t overlap,
                                   int32_t it_begin, int32_t it_end, nss_stream_t stream);
/* Smoothed-aggregation V(1,1)-cycle on a row-partitioned operator with replicated coarse levels, applied
 * natively (the reference has no counterpart: it is single-process; this is what keeps the auxiliary / AMG
 * term of MypreA, templates/NavierStokesSIMPLE_iterative.py:380,383, inside the native multi-GPU loop):
 * a_loc = the slab's rows of the finest operator (columns [owned | ghosts]) with the halo plan of the iterate,
 * r_loc = R[:, owned columns], p_loc = P[owned rows, :], wdinv = omega / diag on the slab, coarse = V-cycle
 * handle of levels 1.. (nss_amg_create).  y = scale * V(b). */
typedef struct nss_dist_amg_s* nss_dist_amg_t;
NSS_API int nss_dist_amg_create(nss_dist_t d, nss_csr_t a_loc, const nss_halo_t* halo_x, nss_csr_t r_loc,
                                nss_csr_t p_loc, const double* wdinv, nss_amg_t coarse, nss_dist_amg_t* out);
NSS_API int nss_dist_amg_destroy(nss_dist_amg_t h);
NSS_API int nss_dist_amg_apply_f64(nss_dist_amg_t h, double scale, const double* b, double* y, nss_stream_t stream);

/* The auxiliary-space term `transform @ preAh1 @ transform.T` of MypreA (templates/NavierStokesSIMPLE_iterative.py:336-337,
 * 357,380,383) on slabs, applied natively inside the partitioned loops (the reference is single-process): tt_loc = the
 * nodal slab's rows of transform.T (columns: velocity dofs [owned | ghosts], halo_x), t_loc = the velocity slab's rows of
 * transform (columns: stacked nodal dofs [owned | ghosts], halo_e), amg = the V-cycle on the stacked block-diagonal nodal
 * Laplacian with replicated coarse levels.  halo_y (may be NULL) = the halo of the loop's t1 as A's operand: needed by
 * the multiplicative form, whose residual x - A y runs between the two sweeps.  y = scale * T V(L) T^T b. */
typedef struct nss_dist_aux_s* nss_dist_aux_t;
NSS_API int nss_dist_aux_create(nss_dist_t d, nss_csr_t tt_loc, const nss_halo_t* halo_x, nss_csr_t t_loc,
                                const nss_halo_t* halo_e, nss_dist_amg_t amg, const nss_halo_t* halo_y, nss_dist_aux_t* out);
NSS_API int nss_dist_aux_destroy(nss_dist_aux_t h);
NSS_API int nss_dist_aux_apply_f64(nss_dist_aux_t h, double scale, const double* b, double* y, nss_stream_t stream);

/* ---- mailbox transport over xGMI (peer-mapped memory, no collective library) -----------------------------------
 * Every rank owns a small fine-grained region -- a mailbox of 2 x nranks word pairs, one arrival flag per source rank
 * and a landing zone for the ghost entries of ONE operand layout (`halo`, direct sends only) -- that its peers map
 * through HIP IPC and write into with plain stores (csrc/p2p.h).  nss_p2p_create writes this rank's blob
 * (nss_p2p_blob_bytes: the 64-byte IPC handle + where it wants each peer's segment); the host gathers the blobs of all
 * ranks in rank order (any side channel) and hands them to nss_p2p_connect.  At most 16 ranks.  Every device-side
 * spin is bounded (3 s): a peer that does not arrive stops the loop, nss_bpcg2_poll then reports done = 3.
 * nss_p2p_allreduce_f64 (one double, rank-ordered sum, the same bits on every rank) and nss_p2p_exchange are the
 * stand-alone forms of what the partitioned loop does with `nss_bpcg2_t.p2p` set (tests). */
typedef struct nss_p2p_s* nss_p2p_t;
NSS_API int nss_p2p_blob_bytes(int32_t nranks, int32_t nhalo, int64_t* bytes);
/* `halos`: HOST array of nhalo (1 .. 4) descriptors, one per operand LAYOUT the loops exchange (BPCG v2: t1; MINRES /
 * BPCG v1: A's operand and B^T's operand), n_owned[c] = owned entries of layout c.  A later exchange finds its layout by
 * the identity of the descriptor's host tables (h_send_off / h_recv_off): pass copies of these descriptors with only
 * `ext` changed. */
NSS_API int nss_p2p_create(int32_t nranks, int32_t rank, int32_t nhalo, const nss_halo_t* const* halos,
                           const int32_t* n_owned, nss_p2p_t* out, void* h_blob);
NSS_API int nss_p2p_connect(nss_p2p_t p, const void* h_blobs);
NSS_API int nss_p2p_destroy(nss_p2p_t p);
NSS_API int nss_p2p_allreduce_f64(nss_p2p_t p, const double* src, double* dst, nss_stream_t stream);
NSS_API int nss_p2p_exchange(nss_p2p_t p, const nss_halo_t* halo, nss_stream_t stream);
NSS_API int nss_p2p_error(nss_p2p_t p, int32_t* timed_out, nss_stream_t stream);
/* every exchange and every one-double all-reduce of the native loops that take this dist handle (MINRES, BPCG v1, the
 * eight-phase BPCG v2 plan) goes through the mailbox transport instead of RCCL (NULL detaches); BPCG v2 on the compact
 * plan fuses its all-reduces into the sum kernels through nss_bpcg2_t.p2p instead */
NSS_API int nss_dist_attach_p2p(nss_dist_t d, nss_p2p_t p);

/* Per-phase device times of the native partitioned loop: between _begin and _end every iteration issued by
 * nss_bpcg2_iterate_dist (up to max_iterations) records 9 HIP events on the compute stream; _end waits for them
 * and returns the average duration (ms) of the 8 segments between them:
 *   0 K1 (B^T rows; + the s1 exchange unless that operand is kept by recurrence) + preA,  1 t1 halo exchange,
 *   2 K2 (A rows),  3 K3 (B rows; + the t4 exchange if any) + local sum,  4 all-reduce <s, K s>,
 *   5 K4 + local sum,  6 all-reduce <w, d>,  7 K5.   (Non-overlapped mode; what tells WHICH collective costs
 * what on a real node.)  Compact plan (dist_compact): 0 C1 + preA,  1 t1 halo exchange,  2 C23 (rows of A and B),
 * 3 local sum,  4 all-reduce <s, K s>,  5 C4 + local sum,  6 all-reduce <w, d>,  7 empty (no K5). */
NSS_API int nss_dist_profile_begin(nss_dist_t d, int32_t max_iterations);
NSS_API int nss_dist_profile_end(nss_dist_t d, double* h_segment_ms /* 8 */, int32_t* iterations);


/* ---- fused preconditioned conjugate gradients -------------------------------------------------
 * The inner solver of the reference's time stepping (`CGSolver(mstar.mat, pre=..)`,
 * templates/NavierStokesSIMPLE_iterative.py:92,130) and the "CG" of BASELINE config 1.
 * x (start value given), r = b - A x, z = pre r, p = z prepared by the host; per iteration:
 * q = A p with partial <p,q>;  alpha = rz / <p,q>;  x += alpha p, r -= alpha q;  z = pre r with
 * partial <r,z>;  beta = rz_new / rz;  p = z + beta p.  Stops when sqrt|<r,z>| < tol * err0.
 * scal: double[8] = { rz, <p,q>, rz_new, err0, tol, -, -, - };  ctrl: int32[4] = { done, it_final,
 * last_it, - };  hist[it] = sqrt|<r,z>| after iteration it. */
typedef struct nss_cg_s {
  nss_csr_t A;
  const double* pre_diag;     /* z = dinv r            -- or NULL */
  nss_bjac_t pre_bjac;        /* block Jacobi / GS     -- or NULL */
  nss_amg_t pre_amg;          /* AMG V-cycle           -- or NULL; none of the three: z = r */
  double *x, *r, *z, *p, *q;
  double* scal;
  int32_t* ctrl;
  double* hist;
  double *partials_a, *partials_b;  /* A's row-block count / element-wise grid: nss_cg_workspace() */
  int32_t n;
} nss_cg_t;
NSS_API int nss_cg_workspace(const nss_cg_t* s, int64_t* partials_a, int64_t* partials_b);
NSS_API int nss_cg_iterate(const nss_cg_t* s, int32_t it_begin, int32_t it_end, nss_stream_t stream);
NSS_API int nss_cg_poll(const nss_cg_t* s, int32_t* done, int32_t* it_final, int32_t* last_it,
                        nss_stream_t stream);

/* ---- device-resident preconditioned Lanczos: the scale factor k -------------------------------
 * Replaces the n-sized work AND the scalar recurrences of `EigenValues_Preconditioner(mat=A, pre=preA, tol=1e-3)`
 * (call sites bramble_pasciak_cg.py:68-74, solvers/bramblepasciak_new.py:111-122; the reference times it inside
 * its solver time, run.py:34-38; NGSolve's implementation is upstream: parity unpinned against it, pinned against
 * the identical recurrence of oracle/krylov_ref.py::lanczos_ritz).  v[0] holds the start vector on entry; the
 * library keeps the Lanczos vectors un-normalised and carries the factors in `scal` (csrc/lanczos.hip).  Step j
 * appends hist[2j] = delta_j, hist[2j+1] = gamma_{j+1}: T = tridiag(gamma_1.., delta_0.., gamma_1..); the host
 * reads them once per batch of steps and solves the small eigenproblem.
 * preA = pre_scale * (pre_amg + (pre_diag | pre_bjac)) as in nss_bpcg2_t (additive MypreA, :383), or -- pre_bjac in
 * Gauss-Seidel mode together with pre_amg -- the multiplicative MypreA (:376-381) with this A in its residual.
 * scal: double[8]; ctrl: int32[4] = { stop, j_stop (-1: gamma_0 == 0), last_j, - }: stop is set by the breakdown
 * test gamma_{j+1} <= 1e-14 max(|delta_0|, |delta_j|), after which every kernel returns at once. */
typedef struct nss_lanczos_s {
  nss_csr_t A;
  const double* pre_diag;
  nss_bjac_t pre_bjac;
  nss_amg_t pre_amg;
  double pre_scale;
  double* v[3];                /* ring: step j reads v[j % 3] (and v[(j + 2) % 3]), writes v[(j + 1) % 3] */
  double* z[2];                /* ring: step j reads z[j % 2], writes z[(j + 1) % 2]                      */
  double* p;
  double* scal;
  int32_t* ctrl;
  double* hist;                /* double[2 * maxsteps]                                                     */
  double *partials_a, *partials_b;  /* sizes: nss_lanczos_workspace()                                      */
  int32_t n;
} nss_lanczos_t;
NSS_API int nss_lanczos_workspace(const nss_lanczos_t* s, int64_t* partials_a, int64_t* partials_b);
/* out[i] = the start vector's entry of global index offset + i (hipla/eigen.py::lanczos_start_values: a hash of the
 * index, so a row-partitioned run fills its slice of the same vector) -- formed on the device instead of uploaded */
NSS_API int nss_lanczos_start_values(int64_t offset, int64_t n, double* out, nss_stream_t stream);
/* z[0] = preA v[0], gamma_0 = sqrt|<z[0], v[0]>|; clears scal / ctrl */
NSS_API int nss_lanczos_start(const nss_lanczos_t* s, nss_stream_t stream);
/* enqueue steps j_begin .. j_end - 1: no host synchronisation */
NSS_API int nss_lanczos_iterate(const nss_lanczos_t* s, int32_t j_begin, int32_t j_end, nss_stream_t stream);
NSS_API int nss_lanczos_poll(const nss_lanczos_t* s, int32_t* stop, int32_t* j_stop, int32_t* last_j, nss_stream_t stream);
/* Small systems run a step in two launches (the rows of A; one kernel that sums both sets of dot partials in every
 * workgroup, keeps the books and applies the update + block Jacobi + dot) when preA is a block Jacobi over runs of
 * consecutive dofs and a sum has <= 1024 partials -- a step is launch-bound there.  Process-wide override for tests
 * and A/B runs: -1 by size (default), 0 never, 1 whenever the operands allow. */
NSS_API int nss_lanczos_fold_mode(int32_t mode);
/* HOST function (no device work): smallest and largest eigenvalue of the symmetric tridiagonal matrix with diagonal
 * diag[0..n) and off-diagonal off[0..n-1) -- the convergence check of the Ritz values every `check_every` steps
 * (Laguerre's iteration from outside the spectrum, verified by Sturm counts, bisection as the fallback; absolute
 * accuracy ~16 ulps of the matrix norm).  On entry *lo / *hi may hold starting points believed to lie below the
 * smallest / above the largest eigenvalue (e.g. the previous check's values pushed outwards; NaN: none) -- used only
 * when a Sturm count confirms them. */
NSS_API int nss_tridiag_extremes(const double* diag, const double* off, int32_t n, double* lo, double* hi);

/* ---- fused preconditioned MINRES ----------------------------------------------------------
 * Replaces the loop body of minres.py:96-144 for K = [[A, B^T], [B, 0]], C = diag(preA, preS)
 * (the operands run.py:45-46 builds).  Vectors are given per block component ([0] velocity, n_u;
 * [1] pressure, n_p).  The three-vector rotations of minres.py:131-133 are index arithmetic on
 * the rings: at iteration k (1-based)  v_old = v[(k-1)%3], v = v[k%3], v_new = v[(k+1)%3]
 * (same for w);  z = z[k%2], z_new = z[(k+1)%2].
 * scal: double[64] = two sets of 32 scalars, double-buffered by the parity of k (iteration k = 1
 * reads the first set; see csrc/minres.hip); ctrl: int32[4] = { stop, k_stop, reason, last_k } with
 * reason 1 = relative break (:126), 2 = absolute guard `ResNorm > tol` failed (:96, warns);
 * hist[k] = ResNorm_k / err0 (:125). */
typedef struct nss_minres_s {
  nss_csr_t A, B, BT;
  const double* pre_diag;      /* point-Jacobi preA (n_u) -- or NULL */
  nss_bjac_t pre_bjac;         /* block-Jacobi preA       -- or NULL */
  nss_amg_t pre_amg;           /* AMG V-cycle             -- or NULL; with pre_diag / pre_bjac: preA = AMG + J */
  const double* minv;          /* preS diagonal (n_p)                */
  double* u[2];
  double* v[3][2];
  double* w[3][2];
  double* z[2][2];
  double* kz[2];
  double* scal;
  int32_t* ctrl;
  double* hist;
  double *partials_a, *partials_b, *partials_c; /* sizes: nss_minres_workspace() */
  int32_t n_u, n_p;
  /* row-partitioned runs: A, B, BT are the slab's local blocks (ghost columns behind the owned ones; B's
   * columns numbered in the layout of A's operand), the z ring vectors are the owned views of
   * halo-extended buffers, and the two sums of an iteration stay local in scal slots 19 / 20 of the
   * iteration's set until the caller (or nss_minres_iterate_dist) all-reduces them out of place into
   * slots 0 (delta) / 2 (gamma_new^2). */
  int32_t local_sums;
} nss_minres_t;

NSS_API int nss_minres_workspace(const nss_minres_t* s, int64_t* partials_a, int64_t* partials_b,
                                 int64_t* partials_c);
/* enqueue iterations k = k_begin .. k_end-1 (1-based, as the reference counts) */
NSS_API int nss_minres_iterate(const nss_minres_t* s, int32_t k_begin, int32_t k_end, nss_stream_t stream);
NSS_API int nss_minres_poll(const nss_minres_t* s, int32_t* stop, int32_t* k_stop, int32_t* reason,
                            int32_t* last_k, nss_stream_t stream);
/* phases first..last of iteration k: 1 the three SpMVs (M1 + M2), 2 local sum of delta, 3 M3 (+ a
 * preconditioner apply that is not fused, and its dot), 4 local sum of gamma_new^2, 5 M4 -- the stretches
 * between the collectives of a row-partitioned schedule driven by the host */
NSS_API int nss_minres_phases(const nss_minres_t* s, int32_t first, int32_t last, int32_t k, nss_stream_t stream);
/* row-partitioned MINRES iterations issued natively (as nss_bpcg2_iterate_dist): per iteration ONE grouped
 * halo exchange (z0 for A and B, z1 for B^T: halo_z0 / halo_z1 describe ring slot 0, slot 1 has the same
 * layout) and two all-reduces of one double.  minres.py:96-144; the reference is single-process. */
NSS_API int nss_minres_iterate_dist(const nss_minres_t* s, nss_dist_t d, const nss_halo_t* halo_z0,
                                    const nss_halo_t* halo_z1, int32_t k_begin, int32_t k_end, nss_stream_t stream);
/* process-wide override of where the two dot-product sums of an iteration are evaluated (tests,
 * measurements): -1 automatic (inside the consuming kernels up to 4096 partials), 0 always by the
 * stand-alone sum kernel, 1 always inside the consumers.  Same bits either way. */
NSS_API int nss_minres_fold_mode(int32_t mode);
/* the same kind of override for applying a block-Jacobi preA (runs of consecutive dofs) inside the
 * element-wise kernel M3 instead of as its own launch: automatic = in the launch-bound regime only.
 * The dot partials are grouped differently in the two forms: results agree to rounding, not bitwise.
 * Round 3: the same regime also runs the rows of B^T INSIDE the launch of A's rows when B^T has at most two entries
 * per row (a fixed-width copy, built on first use; the rows of B share the launch): three dependent launches per
 * iteration.  That part keeps every bit (mode 2 = merged rows only, block Jacobi apart: for tests). */
NSS_API int nss_minres_fuse_mode(int32_t mode);

/* ---- fused Bramble-Pasciak CG, textbook form ------------------------------------------------
 * Replaces the loop body of bramble_pasciak_cg.py:110-143 (6 SpMV per iteration) for
 * C = None, pre_a Jacobi / block-Jacobi / AMG / AMG + Jacobi, pre_schur diagonal.  Block vectors per component
 * ([0] velocity n_u, [1] pressure n_p): x = solution, r = residuum, d =
 * full_preconditioned_residuum, a = a_preconditioned_residuum, t1 / t2 = temp_1 / temp_2.
 * scal: double[16] = { rho, <d,t1>, rho_new, alpha, beta, err0, tolerance, -, local <d,t1>, local rho_new, .. };
 * ctrl: int32[4] = { stop, it_stop, last_it, - };  hist[it] = err_it / err_0 (:118), written
 * before the stop test of iteration `it` (:119).
 * local_sums != 0: row-partitioned run (this rank's slab of the rows; the SpMV operands d[0], t2[0], a[0]
 * are buffers [owned | ghosts] in the layout of A's operand -- B's local columns are numbered in it too --
 * and d[1] in the layout of B^T's operand): the sum kernels leave their LOCAL totals in scal[8] / scal[9] and
 * the caller all-reduces them into scal[1] / scal[2] before the phase that consumes them. */
typedef struct nss_bpcg1_s {
  nss_csr_t A, B, BT;
  const double* pre_diag;
  nss_bjac_t pre_bjac;
  nss_amg_t pre_amg;           /* AMG V-cycle -- or NULL; with pre_diag / pre_bjac: pre_a = AMG + J */
  const double* minv;
  double *x[2], *r[2], *d[2], *a[2], *t1[2], *t2[2];
  double* scal;
  int32_t* ctrl;
  double* hist;
  double *partials_a, *partials_b, *partials_c;
  double k;
  int32_t n_u, n_p;
  int32_t local_sums;
} nss_bpcg1_t;

NSS_API int nss_bpcg1_workspace(const nss_bpcg1_t* s, int64_t* partials_a, int64_t* partials_b,
                                int64_t* partials_c);
NSS_API int nss_bpcg1_iterate(const nss_bpcg1_t* s, int32_t it_begin, int32_t it_end, nss_stream_t stream);
/* One GPU, B^T with at most two entries per row: the rows of B^T ride in the epilogue of A's rows (V1a + V1b in one
 * pass, up to 2^21 velocity rows); small systems (every sum of an iteration <= 1024 partials) also evaluate the
 * loop-top bookkeeping (:115-119), alpha (:129) and rho_new / beta (:137-138) inside the consuming kernels by every
 * workgroup -- 6 dependent launches per iteration instead of 10.  Identical bits in every form.  Process-wide override
 * for tests and A/B runs: -1 by size (default), 0 neither, 1 both whenever B^T allows.  (rho of iteration it is kept in
 * scal[0] for even it and scal[7] for odd it.) */
NSS_API int nss_bpcg1_fold_mode(int32_t mode);
NSS_API int nss_bpcg1_poll(const nss_bpcg1_t* s, int32_t* stop, int32_t* it_stop, int32_t* last_it,
                           nss_stream_t stream);
/* The device phases first .. last of iteration `it` (row-partitioned schedules with a host-side
 * communicator exchange / all-reduce between them):
 *   1  loop top (:115-119); t1 = -K d, t2 = A~ t1 (:125-126)     -- needs the ghosts of d[0], d[1]
 *   2  t1 += [A; B] t2u (:127), local <d, t1>                    -- needs the ghosts of t2[0]
 *   3  alpha (:129) from scal[1]; x, r, a updates (:131-133), local <a_u, r_u>
 *   4  t1p = minv (B a_u - a_p) (:135), local <t1p, r_p>         -- needs the ghosts of a[0]
 *   5  rho_new, beta (:137-138) from scal[2]; d = beta d + ... (:140-141) */
NSS_API int nss_bpcg1_phases(const nss_bpcg1_t* s, int32_t first, int32_t last, int32_t it, nss_stream_t stream);
/* iterations [it_begin, it_end) of the row-partitioned loop with RCCL issued natively: per iteration the
 * grouped exchange of d[0] and d[1], the exchanges of t2[0] and a[0], and two all-reduces of one double.
 * halo_u describes the velocity operands (its `ext` is replaced by d[0] / t2[0] / a[0]), halo_p the operand of
 * B^T (d[1]).  bramble_pasciak_cg.py:110-143; the reference is single-process. */
NSS_API int nss_bpcg1_iterate_dist(const nss_bpcg1_t* s, nss_dist_t d, const nss_halo_t* halo_u,
                                   const nss_halo_t* halo_p, int32_t it_begin, int32_t it_end, nss_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NSS_KRYLOV_H */
