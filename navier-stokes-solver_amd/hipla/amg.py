"""Smoothed-aggregation algebraic multigrid for the velocity block -- the build's counterpart of
the ``'h1amg'`` correction inside the reference's ``MypreA``
(templates/NavierStokesSIMPLE_iterative.py:320-357,380,383).

The reference applies NGSolve's AMG to per-component P1 Laplacians and maps it to the HDG facet
space with a basis transformation ``T`` (upstream FE machinery, not available here).  What that
term buys is mesh-independent iteration counts; here the same is obtained by running a
smoothed-aggregation V(1,1)-cycle directly on ``A`` (``T`` = identity):

* set-up on the host (scipy, once): distance-2 maximal-independent-set aggregation of the
  matrix graph (Luby rounds, vectorised), tentative piecewise-constant prolongator smoothed by one
  damped-Jacobi step ``P = (I - w D^-1 A) P_tent``, Galerkin coarse operators ``P^T A P``, dense
  inverse on the coarsest level;
* cycle on the GPU (``nss_amg_apply_f64``): every level is CSR SpMVs of the same CSR-stream kernel
  with fused Jacobi / residual epilogues -- symmetric (one pre- and one post-smoothing step with the
  same damping), hence an SPD preconditioner for the Bramble-Pasciak CG.

NGSolve's own hierarchy is not visible: iteration counts are pinned against the build's CPU
oracle only (``oracle/numpy_engine.py`` runs the identical cycle on the identical hierarchy)."""

import numpy as np
import scipy.sparse as sp

from .matrix import BaseMatrix, SparseMatrix
from .coloring import _neighbour_max


def _mis2(g, seed=0):
    """Maximal independent set of the distance-2 graph of `g` (roots of the aggregates), without
    forming g @ g: priorities are propagated over two hops."""
    n = g.shape[0]
    rng = np.random.default_rng(seed)
    priority = rng.permutation(n).astype(np.int64) + 1
    cand = np.ones(n, dtype=bool)
    roots = np.zeros(n, dtype=bool)
    while cand.any():
        pri = np.where(cand, priority, 0)
        one = np.maximum(_neighbour_max(g, pri, 0), 0)
        two = _neighbour_max(g, np.maximum(pri, one), 0)          # max over distance <= 2 (incl. self via hop back)
        far = np.maximum(one, two)
        # a node wins if no *other* candidate within distance 2 has a larger priority
        winners = cand & (pri >= far) & (pri > one)
        roots |= winners
        w = winners.astype(np.int64)
        near1 = _neighbour_max(g, w, 0)
        near2 = _neighbour_max(g, np.maximum(w, near1), 0)
        cand &= ~(winners | (near1 > 0) | (near2 > 0))
    return roots


def strength_graph(A, theta):
    """Pattern of the strong couplings ``|a_ij| >= theta * sqrt(a_ii a_jj)`` (off-diagonal).  The
    Galerkin operators of smoothed aggregation carry many weak entries; aggregating over all of
    them over-coarsens the second level (measured: 63x per level without the filter, 8x with)."""
    coo = A.tocoo()
    d = np.abs(A.diagonal())
    keep = coo.row != coo.col
    if theta > 0.0:
        keep &= np.abs(coo.data) >= theta * np.sqrt(d[coo.row] * d[coo.col])
    g = sp.csr_matrix((np.ones(int(keep.sum()), dtype=np.int8), (coo.row[keep], coo.col[keep])), shape=A.shape)
    g.sort_indices()
    return g


def aggregate(A, seed=0, theta=0.0):
    """Aggregates for smoothed aggregation: distance-2 MIS roots (over the strength graph) absorb
    their neighbours, leftovers join an adjacent aggregate.  Returns (aggregate id per node, number
    of aggregates)."""
    n = A.shape[0]
    g = strength_graph(A, theta)
    roots = np.nonzero(_mis2(g, seed))[0]
    agg = -np.ones(n, dtype=np.int64)
    agg[roots] = np.arange(roots.size)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(g.indptr))
    cols = g.indices
    for _ in range(4):                                    # neighbours join, then neighbours of neighbours
        left = agg < 0
        if not left.any():
            break
        m = left[rows] & (agg[cols] >= 0)
        r, first = np.unique(rows[m], return_index=True)
        agg[r] = agg[cols[m][first]]
    left = np.nonzero(agg < 0)[0]
    agg[left] = roots.size + np.arange(left.size)         # isolated nodes: singletons
    return agg, int(agg.max()) + 1


def _coarsest(A, d, omega, dense_limit):
    """Coarsest-level solve as a matrix: the dense inverse while it is small, otherwise (coarsening
    stalled on a large level) one damped-Jacobi step -- both symmetric positive definite."""
    if A.shape[0] <= dense_limit:
        return np.linalg.inv(A.toarray())
    return sp.diags(omega / d).tocsr()


def build_hierarchy(A, max_levels=10, coarse_size=2000, omega=2.0 / 3.0, seed=0, theta=0.04,
                    dense_limit=6000):
    """List of levels, finest first: dict(A, dinv, P, R) and on the coarsest dict(A, dinv, inv).
    `theta` is the strength-of-connection threshold of the aggregation on the coarse levels (the
    finest level aggregates over the full graph)."""
    A = sp.csr_matrix(A)
    A.sort_indices()
    levels = []
    while True:
        d = A.diagonal()
        n = A.shape[0]
        if n <= coarse_size or len(levels) == max_levels - 1:
            levels.append(dict(A=A, dinv=1.0 / d, inv=_coarsest(A, d, omega, dense_limit)))
            break
        agg, nagg = aggregate(A, seed + len(levels), theta if levels else 0.0)
        if nagg > 0.7 * n:                                # stalled: stop here rather than stack levels
            levels.append(dict(A=A, dinv=1.0 / d, inv=_coarsest(A, d, omega, dense_limit)))
            break
        tent = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, nagg))
        P = (tent - omega * (sp.diags(1.0 / d) @ (A @ tent))).tocsr()
        P.sort_indices()
        R = P.T.tocsr()
        R.sort_indices()
        Ac = (R @ A @ P).tocsr()
        Ac.sort_indices()
        levels.append(dict(A=A, dinv=1.0 / d, P=P, R=R))
        A = Ac
    return levels


class SmoothedAggregationAMG(BaseMatrix):
    """``y = V(x)``: one symmetric V(1,1)-cycle of the smoothed-aggregation hierarchy of `mat`
    (a `SparseMatrix`), applied on the engine."""

    def __init__(self, mat, max_levels=10, coarse_size=2000, omega=2.0 / 3.0, seed=0, theta=0.04):
        super().__init__()
        if not isinstance(mat, SparseMatrix):
            raise TypeError("SmoothedAggregationAMG needs a SparseMatrix")
        self.engine = mat.engine
        self.mat = mat
        self.n = mat.height
        self.omega = float(omega)
        host = build_hierarchy(mat.to_scipy(), max_levels, coarse_size, omega, seed, theta)
        self.level_sizes = [lv["A"].shape[0] for lv in host]
        self.operator_complexity = sum(lv["A"].nnz for lv in host) / host[0]["A"].nnz
        eng = self.engine
        self.levels = []
        for i, lv in enumerate(host):
            entry = {"n": lv["A"].shape[0], "dinv": eng.from_host(lv["dinv"])}
            entry["A"] = mat if i == 0 else SparseMatrix.from_scipy(lv["A"], engine=eng)
            if "P" in lv:
                entry["P"] = SparseMatrix.from_scipy(lv["P"], engine=eng)
                entry["R"] = SparseMatrix.from_scipy(lv["R"], engine=eng)
            else:
                inv = sp.csr_matrix(lv["inv"])
                entry["inv"] = SparseMatrix.from_scipy(inv, engine=eng)
            self.levels.append(entry)
        self.handle = eng.amg_create(self.levels, self.omega)

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
