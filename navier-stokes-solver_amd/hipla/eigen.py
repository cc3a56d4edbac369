"""Ritz values of ``pre * mat`` by preconditioned Lanczos -- the build's own
``EigenValues_Preconditioner`` (reference call sites: bramble_pasciak_cg.py:70-74,
solvers/bramblepasciak_new.py:115-122; the NGSolve implementation is upstream and
not visible, SURVEY.md section 8a row A8, so this estimator is *parity unpinned*
against NGSolve and pinned only against the oracle's identical recurrence).

All n-sized work (operator applies, dots, updates) runs through the protocol on
the engine; only the j x j tridiagonal eigenproblem is solved on the host."""

from math import sqrt

import numpy as np

from .vector import BlockVector, InnerProduct


def lanczos_start_values(offset, n, total=None):
    """Deterministic, sliceable start vector: entry i depends only on the global
    index (Knuth multiplicative hash -> [-0.5, 0.5)), so a row-partitioned run
    reproduces the single-GPU vector."""
    i = (np.arange(offset, offset + n, dtype=np.uint64) + np.uint64(1)) * np.uint64(2654435761)
    i = (i ^ (i >> np.uint64(15))) & np.uint64(0xFFFFFFFF)
    return i.astype(np.float64) / 4294967296.0 - 0.5


def _tridiag_eigs(diag, off):
    from scipy.linalg import eigvalsh_tridiagonal
    if len(diag) == 1:
        return np.array(diag, dtype=np.float64)
    return eigvalsh_tridiagonal(np.asarray(diag), np.asarray(off[: len(diag) - 1]))


def lanczos_ritz(mat, pre, start, tol=1e-10, maxsteps=2000, check_every=5, dot=InnerProduct):
    """Preconditioned Lanczos on ``mat`` with SPD (possibly range-restricted) ``pre``.

    r0 = start, z0 = pre r0, gamma0 = sqrt<z0,r0>; then for j = 0,1,..:
    p = mat z_j; delta_j = <p,z_j>; r_{j+1} = p - delta_j v_j - gamma_j v_{j-1};
    z_{j+1} = pre r_{j+1}; gamma_{j+1} = sqrt<z_{j+1}, r_{j+1}>.
    T = tridiag(gamma, delta, gamma); Ritz values = eig(T).  Stops when both
    extreme Ritz values moved by < tol (relative) between two checks."""
    v = start.CreateVector()
    v.data = start
    v_old = start.CreateVector()
    v_new = start.CreateVector()
    z = start.CreateVector()
    z_new = start.CreateVector()
    p = start.CreateVector()
    v_old[:] = 0.0
    z.data = pre * v
    gamma = sqrt(abs(dot(z, v)))
    if gamma == 0.0:
        return np.zeros(0)
    z *= 1.0 / gamma
    v *= 1.0 / gamma
    diag, off = [], []
    lo_prev = hi_prev = None
    ritz = np.zeros(0)
    scale0 = None
    for j in range(maxsteps):
        p.data = mat * z
        delta = dot(p, z)
        v_new.data = p - delta * v - gamma * v_old
        z_new.data = pre * v_new
        g2 = dot(z_new, v_new)
        gamma_new = sqrt(abs(g2))
        diag.append(delta)
        if scale0 is None:
            scale0 = abs(delta)
        breakdown = gamma_new <= 1e-14 * max(scale0, abs(delta))
        if breakdown or (j + 1) % check_every == 0 or j + 1 == maxsteps:
            ritz = _tridiag_eigs(diag, off)
            lo, hi = float(ritz[0]), float(ritz[-1])
            if breakdown:
                break
            if lo_prev is not None and abs(lo - lo_prev) <= tol * abs(lo) and abs(hi - hi_prev) <= tol * abs(hi):
                break
            lo_prev, hi_prev = lo, hi
        off.append(gamma_new)
        z_new *= 1.0 / gamma_new
        v_new *= 1.0 / gamma_new
        v_old, v, v_new = v, v_new, v_old
        z, z_new = z_new, z
        gamma = gamma_new
    return ritz


def EigenValues_Preconditioner(mat, pre, tol=1e-10, inner=InnerProduct):
    """Returns the Ritz values (ascending numpy array) of ``pre * mat``; callers use
    ``min``/``max`` (bramble_pasciak_cg.py:71,74).  ``inner`` is the (global) inner product;
    a row-partitioned operator exposes ``row_offset`` so that every rank fills its slice of
    the same global start vector."""
    start = mat.CreateColVector()
    off = int(getattr(mat, "row_offset", 0))
    if isinstance(start, BlockVector):
        for c in start.components:
            c.set_from(lanczos_start_values(off, len(c)))
            off += len(c)
    else:
        start.set_from(lanczos_start_values(off, len(start)))
    return lanczos_ritz(mat, pre, start, tol=tol, dot=inner)
