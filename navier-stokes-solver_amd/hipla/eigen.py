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
    # values only: the root-free QL/QR variant (dsterf) is 3x cheaper than scipy's default (dstemr) at a few hundred rows
    return eigvalsh_tridiagonal(np.asarray(diag), np.asarray(off[: len(diag) - 1]), lapack_driver="sterf")


class _LanczosState:
    """ctypes mirror of ``nss_lanczos_t`` (include/nss_krylov.h), built lazily (no ctypes at import)."""
    _cls = None

    @classmethod
    def get(cls):
        if cls._cls is None:
            import ctypes as C

            class LanczosState(C.Structure):
                _fields_ = ([(n, C.c_void_p) for n in ("A", "pre_diag", "pre_bjac", "pre_amg")]
                            + [("pre_scale", C.c_double), ("v", C.c_void_p * 3), ("z", C.c_void_p * 2), ("p", C.c_void_p),
                               ("scal", C.c_void_p), ("ctrl", C.c_void_p), ("hist", C.c_void_p),
                               ("partials_a", C.c_void_p), ("partials_b", C.c_void_p), ("n", C.c_int32)])
            cls._cls = LanczosState
        return cls._cls


NATIVE = True          # tests flip this to force the protocol recurrence on native operands
TRACE = None           # tools/lanczos_time.py: a list that receives (label, perf_counter) marks of the native run


def _mark(label):
    if TRACE is not None:
        import time
        TRACE.append((label, time.perf_counter()))


def _native_lanczos(mat, pre, start, tol, maxsteps, check_every):
    """The same recurrence resident on the device (csrc/lanczos.hip: nss_lanczos_*): per step an SpMV with the dot in
    its epilogue, one element-wise kernel, the preconditioner (+ dot) and two single-workgroup sums that also advance
    the scalars; the host reads the new (delta, gamma) pairs once per `check_every` steps.  Returns None when an
    operand is not native to the HIP engine (the caller then runs the protocol recurrence)."""
    import ctypes as C
    from . import fused
    from .matrix import SparseMatrix
    from .vector import Vector
    if not (NATIVE and fused.ENABLED and isinstance(mat, SparseMatrix) and isinstance(start, Vector)):
        return None
    eng = mat.engine
    if getattr(eng, "name", "") != "hip-gfx950" or not hasattr(eng.lib, "nss_lanczos_iterate") or mat.height != mat.width:
        return None
    pa = fused.native_velocity_pre(pre)
    if pa is None or (pa["multiplicative"] and pa["bjac"].mat is not mat):
        return None
    n = mat.height
    _mark("enter")
    st = _LanczosState.get()()
    st.A = mat.handle.ptr
    st.pre_diag = pa["diag"].d.data_ptr() if pa["diag"] is not None else None
    st.pre_bjac = pa["bjac"].handle.ptr if pa["bjac"] is not None else None
    st.pre_amg = pa["amg"].handle.ptr if pa["amg"] is not None else None
    st.pre_scale = float(pa["scale"])
    vecs = [eng.zeros(n) for _ in range(6)]
    eng.copy(start.buf, vecs[0])
    for i in range(3):
        st.v[i] = vecs[i].data_ptr()
    st.z[0], st.z[1], st.p = vecs[3].data_ptr(), vecs[4].data_ptr(), vecs[5].data_ptr()
    st.n = n
    na, nb = C.c_int64(), C.c_int64()
    eng._check(eng.lib.nss_lanczos_workspace(C.byref(st), C.byref(na), C.byref(nb)))
    partials = [eng.zeros(max(1, na.value)), eng.zeros(max(1, nb.value))]
    st.partials_a, st.partials_b = partials[0].data_ptr(), partials[1].data_ptr()
    scal = eng.zeros(8)
    ctrl = eng.torch.zeros(4, dtype=eng.torch.int32, device=eng.device)
    hist = eng.zeros(2 * maxsteps)
    st.scal, st.ctrl, st.hist = scal.data_ptr(), ctrl.data_ptr(), hist.data_ptr()
    _mark("workspace")
    eng._check(eng.lib.nss_lanczos_start(C.byref(st), eng.stream))
    torch = eng.torch
    # The host looks at a batch while the device already runs the next one: after every batch the control words and
    # the new (delta, gamma) pairs go to pinned host memory (stream-ordered copies, an event behind them); the host
    # waits for the event of batch b only after batch b + 1 is enqueued.  When batch b ends the run, the steps of
    # batch b + 1 were wasted work -- the result does not see them.
    h_ctrl = [torch.empty(4, dtype=torch.int32).pin_memory() for _ in range(2)]
    h_hist = torch.empty(2 * maxsteps, dtype=torch.float64).pin_memory()
    pending = []                                          # (end, event, slot)

    # A batch is a multiple of `check_every` steps sized to ~0.25 ms of device work: long enough that the host's
    # per-batch work (two copies, an event, the checks: ~50 us) hides behind it, short enough that the steps enqueued
    # beyond the one that ends the run -- the rest of its batch and the speculative next one -- stay a small part
    # (40-step batches wasted 70 of 360 steps at 6.7e4 rows).  The checks of the steps check_every - 1,
    # 2 check_every - 1, ... inside a batch are made in order afterwards, exactly as if the host had looked after each.
    est_step = 12e-6 + (12.0 * mat.nnz + 80.0 * n) / 5e12          # (two launches + the bytes of a step)
    per_batch = check_every * max(1, min(8, int(round(0.25e-3 / est_step / check_every))))
    nbatch = [0]

    def enqueue(j0):
        end = min(maxsteps, j0 + per_batch)
        eng._check(eng.lib.nss_lanczos_iterate(C.byref(st), j0, end, eng.stream))
        slot = nbatch[0] & 1
        nbatch[0] += 1
        h_ctrl[slot].copy_(ctrl, non_blocking=True)
        h_hist[2 * j0: 2 * end].copy_(hist[2 * j0: 2 * end], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        pending.append((j0, end, ev, slot))
        return end

    # The check itself: the library's host routine (Laguerre from just outside the previous check's values, verified
    # by Sturm counts: ~20 us at 300 rows where LAPACK's bisection through scipy took ~200 -- with a check every 5
    # steps that was more host time than a small system's steps take on the device).
    c_lo, c_hi = C.c_double(), C.c_double()
    moved = [None, None]

    def extremes(diag, off):
        if moved[0] is None:
            c_lo.value = c_hi.value = float("nan")
        else:                                             # the extreme Ritz values only move outwards (interlacing)
            pad = 1e-14 * abs(hi_prev)
            c_lo.value, c_hi.value = lo_prev - 2.0 * moved[0] - pad, hi_prev + 2.0 * moved[1] + pad
        eng._check(eng.lib.nss_tridiag_extremes(diag.ctypes.data, off.ctypes.data if len(off) else None, len(diag),
                                                C.byref(c_lo), C.byref(c_hi)))
        return c_lo.value, c_hi.value

    lo_prev = hi_prev = None
    diag = off = np.zeros(0)
    finished = False
    _mark("pinned buffers")
    j = enqueue(0)
    while pending and not finished:
        if j < maxsteps:
            j = enqueue(j)                                # the next batch runs while this one is looked at
        j0, end, ev, slot = pending.pop(0)
        ev.synchronize()
        stop, j_stop = int(h_ctrl[slot][0]), int(h_ctrl[slot][1])
        if stop and j_stop < 0:
            torch.cuda.synchronize()
            return np.zeros(0)                            # gamma_0 == 0
        j_end = j_stop if stop and j_stop < end else end - 1          # last step of this batch that was computed
        h = h_hist[: 2 * (j_end + 1)].numpy()
        checks = [jj for jj in range(j0, j_end + 1) if (jj + 1) % check_every == 0 or jj + 1 == maxsteps]
        if stop and j_stop <= j_end and j_stop not in checks:
            checks.append(j_stop)                         # the breakdown step is looked at as well
        for jj in sorted(checks):
            diag, off = h[0::2][: jj + 1].copy(), h[1::2][: jj].copy()
            lo, hi = extremes(diag, off)
            if (stop and jj == j_stop) or (lo_prev is not None and abs(lo - lo_prev) <= tol * abs(lo)
                                           and abs(hi - hi_prev) <= tol * abs(hi)):
                finished = True
                break
            if lo_prev is not None:
                moved[0], moved[1] = max(lo_prev - lo, 0.0), max(hi - hi_prev, 0.0)
            lo_prev, hi_prev = lo, hi
    _mark("converged (%d steps, %d enqueued)" % (len(diag), j))
    # The final eigenproblem is solved while the speculative steps still in flight (less than one batch: ~0.25 ms of
    # device work by the batch size) drain.  (Raising the stop flag from a second stream to cut them short was tried:
    # it gains nothing once the two overlap, and the first use of a second torch stream costs 5.5 ms.)
    ritz = _tridiag_eigs(list(diag), list(off))
    _mark("all Ritz values")
    torch.cuda.synchronize()                              # (before the buffers of the speculative steps go away)
    _mark("drained")
    return ritz


def lanczos_ritz(mat, pre, start, tol=1e-10, maxsteps=2000, check_every=5, dot=InnerProduct):
    """Preconditioned Lanczos on ``mat`` with SPD (possibly range-restricted) ``pre``.

    r0 = start, z0 = pre r0, gamma0 = sqrt<z0,r0>; then for j = 0,1,..:
    p = mat z_j; delta_j = <p,z_j>; r_{j+1} = p - delta_j v_j - gamma_j v_{j-1};
    z_{j+1} = pre r_{j+1}; gamma_{j+1} = sqrt<z_{j+1}, r_{j+1}>.
    T = tridiag(gamma, delta, gamma); Ritz values = eig(T).  Stops when both
    extreme Ritz values moved by < tol (relative) between two checks."""
    if dot is InnerProduct and getattr(start, "comm", None) is None:
        native = _native_lanczos(mat, pre, start, tol, maxsteps, check_every)
        if native is not None:
            return native
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

    def fill(c, off):
        eng = getattr(c, "engine", None)
        if hasattr(eng, "lanczos_start_values") and hasattr(c, "buf"):
            eng.lanczos_start_values(c.buf[: len(c)], off)        # on the device: no host pass, no upload
        else:
            c.set_from(lanczos_start_values(off, len(c)))

    if isinstance(start, BlockVector):
        for c in start.components:
            fill(c, off)
            off += len(c)
    else:
        fill(start, off)
    return lanczos_ritz(mat, pre, start, tol=tol, dot=inner)
