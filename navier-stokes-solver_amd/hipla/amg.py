"""Smoothed-aggregation algebraic multigrid for the velocity block -- the build's counterpart of
the ``'h1amg'`` correction inside the reference's ``MypreA``
(templates/NavierStokesSIMPLE_iterative.py:320-357,380,383).

The reference applies NGSolve's AMG to per-component P1 Laplacians and maps it to the HDG facet
space with a basis transformation ``T`` (upstream FE machinery, not available here).  What that
term buys is mesh-independent iteration counts; here the same is obtained by running a
smoothed-aggregation V(1,1)-cycle directly on ``A`` (``T`` = identity):

* set-up on the GPU, level by level (``csrc/amg_setup.hip``): distance-2 maximal-independent-set
  aggregation over the strength graph (``nss_amg_aggregate``; the finest level aggregates over the
  full matrix graph, coarse levels only over couplings ``|a_ij| >= theta sqrt(a_ii a_jj)`` -- the
  Galerkin operators carry many weak entries and aggregating over all of them over-coarsens:
  63x per level without the filter, 8x with it, condition number 16 -> 4.3 at 1e7 dofs), smoothed
  prolongator ``P = (I - w D^-1 A) P_tent`` (``nss_amg_prolongator``), ``R = P^T``
  (``nss_csr_transpose``), Galerkin operator ``R (A P)`` (``nss_csr_spgemm``, deterministic
  expand / sort / compress).  Only the coarsest operator (<= `coarse_size` rows) is inverted on
  the host;
* cycle on the GPU (``nss_amg_apply_f64``): every level is CSR SpMVs of the CSR-stream kernel with
  fused Jacobi / residual epilogues -- symmetric (one pre- and one post-smoothing step with the
  same damping), hence an SPD preconditioner for the Bramble-Pasciak CG.

NGSolve's own hierarchy is not visible: parity is pinned against the build's CPU oracle only
(``oracle/krylov_ref.py::sa_*`` restates the set-up with numpy/scipy -- aggregates identical,
operators bit-identical; ``oracle/numpy_engine.py`` runs the identical cycle)."""

import os

import numpy as np
import scipy.sparse as sp

from .matrix import BaseMatrix, SparseMatrix


def _priorities(n, seed):
    """Distinct positive pseudo-random priorities for the Luby rounds: (i + 1) * odd constant modulo
    2^63 -- a bijection, so no two nodes tie -- in 0.1 s for 4e7 nodes (a true random permutation
    costs a second there and gives the same iteration counts)."""
    if os.environ.get("NSS_AMG_PRIORITY") == "permutation":
        return np.random.default_rng(seed).permutation(n).astype(np.int64) + 1
    mult = np.uint64(0x9E3779B97F4A7C15) + np.uint64(2) * np.uint64(0x632BE5AB * (seed + 1) % (1 << 31))
    pri = np.arange(1, n + 1, dtype=np.uint64)
    pri *= mult                                           # wraps modulo 2^64 (in place: one allocation)
    pri &= np.uint64((1 << 63) - 1)
    return pri.view(np.int64)


def _coarsest(A, omega, dense_limit):
    """Coarsest-level solve as a matrix: the dense inverse while it is small, otherwise (coarsening
    stalled on a large level) one damped-Jacobi step -- both symmetric positive definite."""
    host = A.to_scipy()
    if A.height <= dense_limit:
        dense = np.linalg.inv(host.toarray())
        n = dense.shape[0]
        if np.count_nonzero(dense) == dense.size:     # (the usual case: the CSR arrays written directly -- scipy's
            # dense -> CSR conversion goes through nonzero() and a COO sort: 0.15 s for the 1487-row coarse level of cfg4)
            inv = sp.csr_matrix((dense.ravel(), np.tile(np.arange(n, dtype=np.int32), n),
                                 np.arange(0, n * n + 1, n, dtype=np.int32)), shape=(n, n))
        else:
            inv = sp.csr_matrix(dense)
    else:
        inv = sp.diags(omega / host.diagonal()).tocsr()
    return SparseMatrix.from_scipy(inv, engine=A.engine)


def build_hierarchy(mat, max_levels=10, coarse_size=2000, omega=2.0 / 3.0, seed=0, theta=0.04,
                    dense_limit=6000):
    """Levels, finest first: dict(n, A, dinv, P, R) and on the coarsest dict(n, A, dinv, inv); the
    operators are `SparseMatrix` objects resident in the engine, `dinv` an engine buffer.  `theta`
    is the strength-of-connection threshold of the aggregation on the coarse levels."""
    eng = mat.engine
    levels = []
    A = mat
    while True:
        n = A.height
        entry = dict(n=n, A=A, dinv=eng.csr_inverse_diagonal(A.handle))
        last = n <= coarse_size or len(levels) == max_levels - 1
        if not last:
            priority = _priorities(n, seed + len(levels))
            agg, nagg = eng.amg_aggregate(A.handle, theta if levels else 0.0, priority)
            last = nagg > 0.7 * n                         # stalled: stop here rather than stack levels
        if last:
            entry["inv"] = _coarsest(A, omega, dense_limit)
            levels.append(entry)
            if hasattr(eng, "scratch_trim"):
                eng.scratch_trim()                        # the pooled temporaries of the sparse products
            return levels
        P = SparseMatrix.from_handle(eng.amg_prolongator(A.handle, agg, nagg, omega), eng)
        R = P.CreateTranspose()
        AP = eng.csr_spgemm(A.handle, P.handle)
        entry["P"], entry["R"] = P, R
        levels.append(entry)
        A = SparseMatrix.from_handle(eng.csr_spgemm(R.handle, AP), eng)


class SmoothedAggregationAMG(BaseMatrix):
    """``y = V(x)``: one symmetric V(1,1)-cycle of the smoothed-aggregation hierarchy of `mat`
    (a `SparseMatrix`), set up and applied on the engine."""

    def __init__(self, mat, max_levels=10, coarse_size=2000, omega=2.0 / 3.0, seed=0, theta=0.04):
        super().__init__()
        if not isinstance(mat, SparseMatrix):
            raise TypeError("SmoothedAggregationAMG needs a SparseMatrix")
        self.engine = mat.engine
        self.mat = mat
        self.n = mat.height
        self.omega = float(omega)
        self.levels = build_hierarchy(mat, max_levels, coarse_size, omega, seed, theta)
        self.level_sizes = [lv["n"] for lv in self.levels]
        self.operator_complexity = sum(lv["A"].nnz for lv in self.levels) / mat.nnz
        self.handle = self.engine.amg_create(self.levels, self.omega)

    def Height(self):
        return self.n

    def Width(self):
        return self.n

    def Mult(self, x, y):
        self.engine.amg_apply(self.handle, 1.0, x.buf, y.buf)

    MultTrans = Mult

    @property
    def T(self):
        return self


class AuxiliarySpaceAMG(SmoothedAggregationAMG):
    """``y = T (sum_c E_c V_c E_c^T) T^T x`` -- the auxiliary-space term of the reference's ``MypreA``,
    ``transform @ preAh1 @ transform.T`` with ``preAh1 = sum_c emb_c @ Preconditioner(aH1_c, 'h1amg') @
    emb_c.T`` (templates/NavierStokesSIMPLE_iterative.py:291,320-357,380,383).

    `transform`: `SparseMatrix` (velocity dofs x stacked auxiliary dofs); `components`: one
    `SmoothedAggregationAMG` per velocity component, built on that component's auxiliary Laplacian, in
    the order in which their spaces are stacked.  One native handle (``nss_amg_create_auxiliary``): the
    fused Krylov loops apply it wherever they accept a V-cycle, alone or added to a (block) Jacobi
    (the additive ``MypreA``, :383)."""

    def __init__(self, transform, components):
        BaseMatrix.__init__(self)
        if not isinstance(transform, SparseMatrix) or not all(type(c) is SmoothedAggregationAMG for c in components):
            raise TypeError("AuxiliarySpaceAMG needs a SparseMatrix transform and plain V-cycles")
        if sum(c.n for c in components) != transform.width:
            raise ValueError("component sizes do not add up to the columns of the transform")
        self.engine = transform.engine
        self.transform, self.components = transform, list(components)
        self.transform_t = transform.CreateTranspose()
        self.n = transform.height
        self.mat = None
        self.levels = []
        self.level_sizes = [c.level_sizes for c in components]
        self.handle = self.engine.amg_create_auxiliary(transform.handle, self.transform_t.handle,
                                                       [c.handle for c in components])
