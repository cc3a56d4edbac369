"""Host side of the fused, device-resident Krylov loops (C ABI: ``nss_bpcg2_*`` ...).

``*.try_create`` return a loop object when every operand is native to the HIP engine
(CSR ``SparseMatrix`` blocks, Jacobi / block-Jacobi preconditioners, plain vectors) and
``None`` otherwise -- the caller then drives the same algorithm through the operator
protocol (still on the GPU, one kernel per statement), which is what keeps user
``BaseMatrix`` subclasses working.

The loops enqueue ``poll_every`` iterations at a time without any host synchronisation;
alpha / beta / the stop test live on the device and a ``done`` flag freezes the state at
exactly the iteration where the reference would ``break``."""

import ctypes as C
import os

import numpy as np

from .amg import SmoothedAggregationAMG
from .matrix import (BlockGaussSeidel, BlockJacobi, BlockMatrix, DiagonalMatrix, ScaledMatrix, SparseMatrix,
                     SumMatrix)
from .vector import BlockVector, Vector

POLL_EVERY = int(os.environ.get("NSS_POLL_EVERY", "32"))
ENABLED = True      # tests flip this to force the protocol path on native operands


class Bpcg2State(C.Structure):
    """ctypes mirror of ``nss_bpcg2_t`` (include/nss_krylov.h)."""
    _fields_ = ([(n, C.c_void_p) for n in ("A", "B", "BT", "pre_diag", "pre_bjac", "pre_amg", "minv")]
                + [(n, C.c_void_p) for n in ("u0", "u1", "d0", "d1", "w0", "w1", "s0", "s1", "z0", "q",
                                             "t0", "t1", "t2", "t3", "t4")]
                + [("scal", C.c_void_p), ("ctrl", C.c_void_p), ("hist", C.c_void_p),
                   ("partials_a", C.c_void_p), ("partials_b", C.c_void_p), ("partials_c", C.c_void_p),
                   ("k", C.c_double), ("n_u", C.c_int32), ("n_p", C.c_int32)]
                + [(n, C.c_void_p) for n in ("cond_HT", "cond_H", "cond_inner", "cond_f")]
                + [("ghost_mode", C.c_int32), ("ghost_n", C.c_int32), ("ghost_map", C.c_void_p),
                   ("ghost_s0", C.c_void_p), ("ghost_w0", C.c_void_p),
                   ("ghost_p_mode", C.c_int32), ("ghost_p_n", C.c_int32), ("ghost_b", C.c_void_p),
                   ("ghost_t3", C.c_void_p), ("ghost_w1", C.c_void_p), ("ghost_minv", C.c_void_p),
                   ("local_sums", C.c_int32), ("pre_dist_amg", C.c_void_p), ("dist_compact", C.c_int32),
                   ("pre_dist_aux", C.c_void_p), ("p2p", C.c_void_p)])


class HaloStruct(C.Structure):
    """ctypes mirror of ``nss_halo_t`` (include/nss_krylov.h)."""
    _fields_ = [("send_idx", C.c_void_p), ("sendbuf", C.c_void_p), ("ext", C.c_void_p),
                ("h_send_peer", C.c_void_p), ("h_send_off", C.c_void_p), ("h_send_cnt", C.c_void_p),
                ("h_recv_peer", C.c_void_p), ("h_recv_off", C.c_void_p), ("h_recv_cnt", C.c_void_p),
                ("n_pack", C.c_int32), ("n_send", C.c_int32), ("n_recv", C.c_int32),
                ("int_begin", C.c_int32), ("int_end", C.c_int32), ("direct", C.c_int32)]


PHASE = {"K1": 1, "K2": 2, "K3": 3, "SUM1": 4, "ALPHA": 5, "K4": 6, "SUM2": 7, "BETA": 8, "K5": 9}
CPHASE = {"C1": 1, "C23": 2, "SUMA": 3, "C4": 4, "SUMW": 5}      # compact plan (NSS_BPCG2C_*)
S_WD, S_AS, S_WDN, S_ALPHA, S_BETA, S_ERR0, S_TOL, S_REL = range(8)


def _hip(engine):
    return getattr(engine, "name", "") == "hip-gfx950" and hasattr(engine.lib, "nss_bpcg2_iterate")


def native_diag(op):
    """(scale, DiagonalMatrix) if `op` is a (scaled) diagonal preconditioner, else None."""
    scale = 1.0
    if isinstance(op, ScaledMatrix):
        scale, op = op.scale, op.mat
    if isinstance(op, DiagonalMatrix):
        return scale, op
    return None


def native_bjac(op):
    """(scale, op) for a (scaled) block-Jacobi or symmetric block Gauss-Seidel preconditioner: both
    own an ``nss_bjac_t`` handle (the latter in Gauss-Seidel mode), which the fused loops apply."""
    scale = 1.0
    if isinstance(op, ScaledMatrix):
        scale, op = op.scale, op.mat
    if isinstance(op, BlockGaussSeidel) and op.middle is not None:
        return None                      # sweep + middle operator: protocol path
    if isinstance(op, (BlockJacobi, BlockGaussSeidel)):
        return scale, op
    return None


def native_velocity_pre(op):
    """Decompose a velocity-block preconditioner into what the fused BPCG loop applies natively:
    ``scale * (AMG + J)`` with J a point / block Jacobi (or block Gauss-Seidel when there is no AMG
    term) -- the additive form of the reference's MypreA
    (templates/NavierStokesSIMPLE_iterative.py:383).  Returns dict(scale, amg, diag, bjac) or None."""
    scale = 1.0
    if isinstance(op, ScaledMatrix):
        scale, op = op.scale, op.mat
    parts = [op]
    if isinstance(op, SumMatrix):
        if op.sb != 1.0:
            return None
        parts = [op.a, op.b]
    out = {"scale": scale, "amg": None, "diag": None, "bjac": None, "multiplicative": False}
    if (len(parts) == 1 and isinstance(op, BlockGaussSeidel) and isinstance(op.middle, SmoothedAggregationAMG)):
        # the multiplicative MypreA (GS=True, :376-381): sweep, residual, (auxiliary-space) AMG correction,
        # back sweep -- composed natively by the fused BPCG loop
        out.update(bjac=op, amg=op.middle, multiplicative=True)
        return out
    for part in parts:
        if isinstance(part, SmoothedAggregationAMG) and out["amg"] is None:
            out["amg"] = part
        elif isinstance(part, DiagonalMatrix) and out["diag"] is None and out["bjac"] is None:
            out["diag"] = part
        elif (isinstance(part, (BlockJacobi, BlockGaussSeidel)) and out["diag"] is None and out["bjac"] is None
              and getattr(part, "middle", None) is None):
            out["bjac"] = part
        else:
            return None
    if out["amg"] is not None and isinstance(out["bjac"], BlockGaussSeidel):
        return None
    return out


def plan_for_textbook_bpcg(a_matrix, pre_a):
    """Launch plans the fused textbook BPCG loop wants, made BEFORE anything multiplies with the matrices (the scale
    factor's Lanczos runs over A: with the plan changed afterwards the first solve on fresh matrices would differ from
    later ones in the last bits of k).  Block Jacobi alone: A's row blocks around the Jacobi blocks (small systems), so
    that the launch of A's rows applies it in its epilogue.  No-op for anything that is not native."""
    if not ENABLED or not isinstance(a_matrix, SparseMatrix) or not hasattr(a_matrix.handle, "plan_for_blocks"):
        return
    parts = native_velocity_pre(pre_a)
    if parts is None or parts["amg"] is not None or parts["multiplicative"] or not isinstance(parts["bjac"], BlockJacobi):
        return
    a_matrix.handle.plan_for_blocks(parts["bjac"].handle)


def _amg_plus_jacobi(op):
    """(amg, (1.0, diag) | None, (1.0, bjac) | None) when `op` is an unscaled AMG V-cycle, optionally
    plus a point / block Jacobi (the additive MypreA); (None, None, None) otherwise."""
    parts = native_velocity_pre(op)
    if parts is None or parts["amg"] is None or parts["scale"] != 1.0 or parts["multiplicative"]:
        return None, None, None
    diag = (1.0, parts["diag"]) if parts["diag"] is not None else None
    bjac = (1.0, parts["bjac"]) if parts["bjac"] is not None else None
    return parts["amg"], diag, bjac


def _extension_is_in_place_safe(H):
    """`t1 += H t1` runs in place on the device: the rows of H that hold entries (interior dofs) must
    not appear among its columns (coupling dofs)."""
    rowptr, col, _ = H.host_csr()
    rows_with_entries = np.nonzero(np.diff(rowptr))[0]
    return not np.intersect1d(rows_with_entries, np.unique(col), assume_unique=True).size


def _plain(v, n):
    return isinstance(v, Vector) and v.size == n


class Bpcg2Loop:
    """Device-resident iteration of solvers/bramblepasciak_new.py:200-249."""

    @classmethod
    def try_create(cls, matA, matB, matBT, preA_unscaled, k, preM, vecs, distributed=False, condensed=None,
                   dist_amg=None, ghost_rows_b=0, dist_aux=None):
        """`distributed`: the matrices are the local row blocks of a partitioned run -- their
        column spaces carry halo entries behind the owned ones and t1 / t4 / s1 are the owned
        views of halo-extended buffers (same base pointer).  `condensed`: dict(HT, H, inner) of
        `SparseMatrix` for a statically condensed form (matA is then the explicit product).
        `ghost_rows_b` > 0: matB carries that many ghost pressure rows behind the slab's own (the compact
        partitioned plan, nss_bpcg2_t.dist_compact)."""
        if not (isinstance(matA, SparseMatrix) and isinstance(matB, SparseMatrix) and isinstance(matBT, SparseMatrix)):
            return None
        eng = matA.engine
        if not ENABLED or not _hip(eng):
            return None
        n_u, n_p = matA.height, matB.height - int(ghost_rows_b)
        if matBT.height != n_u or (ghost_rows_b and not distributed):
            return None
        if not distributed and (matA.width != n_u or matB.width != n_u or matBT.width != n_p):
            return None
        if distributed and (matA.width < n_u or matB.width < n_u or matBT.width < n_p):
            return None
        pm = native_diag(preM)
        if dist_aux is not None:     # row-partitioned auxiliary-space term [+ Jacobi part | around Gauss-Seidel sweeps]
            pa = native_velocity_pre(preA_unscaled) if preA_unscaled is not None else None
            if pa is None:
                pa = {"scale": 1.0, "amg": None, "diag": None, "bjac": None, "multiplicative": False}
            if pa["amg"] is not None or pa["multiplicative"] or not distributed or dist_amg is not None:
                return None
        elif dist_amg is not None:   # row-partitioned V-cycle (native handle) [+ an additive Jacobi part]
            pa = native_velocity_pre(preA_unscaled) if preA_unscaled is not None else None
            if pa is None:
                pa = {"scale": 1.0, "amg": None, "diag": None, "bjac": None, "multiplicative": False}
            if pa["amg"] is not None or pa["multiplicative"] or not distributed:
                return None
        else:
            pa = native_velocity_pre(preA_unscaled)
        if pm is None or pa is None:
            return None
        if pa["multiplicative"] and (condensed is not None or distributed or pa["bjac"].mat is not matA):
            return None                  # the sweeps' residual is formed with the loop's own A
        sizes = {"u0": n_u, "d0": n_u, "w0": n_u, "s0": n_u, "z0": n_u, "q": n_u, "t0": n_u, "t1": n_u,
                 "t2": n_u, "t4": n_u, "u1": n_p, "d1": n_p, "w1": n_p, "s1": n_p, "t3": n_p}
        if any(not _plain(vecs.get(name), n) for name, n in sizes.items()):
            return None
        if condensed is not None:
            if distributed or any(not isinstance(condensed.get(key), SparseMatrix) or condensed[key].height != n_u
                                  or condensed[key].width != n_u for key in ("HT", "H", "inner")):
                return None
            if not _extension_is_in_place_safe(condensed["H"]):
                return None
        return cls(eng, matA, matB, matBT, pa, k, pm, vecs, condensed, distributed, dist_amg, n_p, dist_aux)

    def __init__(self, eng, matA, matB, matBT, pa, k, pm, vecs, condensed=None, distributed=False, dist_amg=None,
                 n_p=None, dist_aux=None):
        torch = eng.torch
        self.eng, self.lib = eng, eng.lib
        self.keep = [matA, matB, matBT, vecs, pa, pm, condensed]       # keep device memory alive
        st = Bpcg2State()
        if condensed is not None:
            self.cond_f = eng.zeros(matA.height)
            st.cond_HT, st.cond_H = condensed["HT"].handle.ptr, condensed["H"].handle.ptr
            st.cond_inner, st.cond_f = condensed["inner"].handle.ptr, self.cond_f.data_ptr()
        # the rows of B multiply t1 - s0 (:212-213): with row blocks short enough both vectors are read from LDS copies
        self.pair_staged_b = (os.environ.get("NSS_PAIR_STAGE", "1") == "1" and hasattr(matB.handle, "plan_for_pairs")
                              and matB.handle.plan_for_pairs())
        # block Jacobi alone as preA: B^T's row blocks are planned around its blocks and C1 applies it in its epilogue
        self.c1_applies_bjac = (pa["bjac"] is not None and pa["diag"] is None and pa["amg"] is None and condensed is None
                                and hasattr(matBT.handle, "plan_for_blocks") and matBT.handle.plan_for_blocks(pa["bjac"].handle))
        st.A, st.B, st.BT = matA.handle.ptr, matB.handle.ptr, matBT.handle.ptr
        st.pre_diag = pa["diag"].d.data_ptr() if pa["diag"] is not None else None
        st.pre_bjac = pa["bjac"].handle.ptr if pa["bjac"] is not None else None
        st.pre_amg = pa["amg"].handle.ptr if pa["amg"] is not None else None
        st.k = float(k) * pa["scale"]
        mscale, mop = pm
        if mscale != 1.0:
            self.minv = mop.d * mscale
        else:
            self.minv = mop.d
        st.minv = self.minv.data_ptr()
        for name in ("u0", "u1", "d0", "d1", "w0", "w1", "s0", "s1", "z0", "q", "t0", "t1", "t2", "t3", "t4"):
            setattr(st, name, vecs[name].buf.data_ptr())
        st.n_u, st.n_p = matA.height, (matB.height if n_p is None else int(n_p))
        st.local_sums = 1 if distributed else 0     # the caller all-reduces scal[9], scal[10] into scal[1], scal[2]
        st.pre_dist_amg = dist_amg
        st.pre_dist_aux = dist_aux
        self.keep.append(dist_amg)
        na, nb, nc = C.c_int64(), C.c_int64(), C.c_int64()
        eng._check(self.lib.nss_bpcg2_workspace(C.byref(st), C.byref(na), C.byref(nb), C.byref(nc)))
        self.partials = [eng.zeros(max(1, v.value)) for v in (na, nb, nc)]
        st.partials_a, st.partials_b, st.partials_c = (p.data_ptr() for p in self.partials)
        self.scal = eng.zeros(16)
        self.ctrl = torch.zeros(8, dtype=torch.int32, device=eng.device)
        st.scal, st.ctrl = self.scal.data_ptr(), self.ctrl.data_ptr()
        self.hist = None
        self.state = st

    # ---- driving the loop -------------------------------------------------------------------
    def start(self, wdn, err0, tol, rel_err, maxsteps):
        """Upload the scalars of iteration 0 and clear the control words / history."""
        eng, st = self.eng, self.state
        self.maxsteps = int(maxsteps)
        self.hist = eng.zeros(max(1, self.maxsteps))
        st.hist = self.hist.data_ptr()
        scal = np.zeros(16)
        scal[S_WD], scal[S_ERR0], scal[S_TOL], scal[S_REL] = wdn, err0, tol, 1.0 if rel_err else 0.0
        eng.upload(scal, self.scal)
        self.ctrl.zero_()

    def enqueue(self, it_begin, it_end):
        """Enqueue iterations [it_begin, it_end) on the current stream; returns immediately."""
        self.eng._check(self.lib.nss_bpcg2_iterate(C.byref(self.state), int(it_begin), int(it_end), self.eng.stream))

    def phase(self, name, it):
        self.eng._check(self.lib.nss_bpcg2_phase(C.byref(self.state), PHASE[name], int(it), self.eng.stream))

    def phases(self, first, last, it):
        self.eng._check(self.lib.nss_bpcg2_phases(C.byref(self.state), PHASE[first], PHASE[last], int(it),
                                                  self.eng.stream))

    def cphases(self, first, last, it):
        """Phases first..last of iteration `it` of the compact plan (what `enqueue` issues)."""
        self.eng._check(self.lib.nss_bpcg2_cphases(C.byref(self.state), CPHASE[first], CPHASE[last], int(it),
                                                   self.eng.stream))

    def enqueue_classic(self, it_begin, it_end):
        """The same iterations in the eight-phase form (cross-checks, measurements)."""
        self.eng._check(self.lib.nss_bpcg2_iterate_classic(C.byref(self.state), int(it_begin), int(it_end),
                                                           self.eng.stream))

    def folds_sums(self):
        out = C.c_int32()
        self.eng._check(self.lib.nss_bpcg2_folds_sums(C.byref(self.state), C.byref(out)))
        return bool(out.value)

    def c1_applies_preA(self):
        """Whether the next C1 applies the block Jacobi in its epilogue (B^T planned around the blocks, and the size
        rule / override of nss_bpcg2_fuse_block_jacobi)."""
        out = C.c_int32()
        self.eng._check(self.lib.nss_bpcg2_c1_applies_preA(C.byref(self.state), C.byref(out)))
        return bool(out.value)

    def enqueue_dist(self, dist_handle, halos, overlap, it_begin, it_end):
        """Row-partitioned iterations issued natively (nss_bpcg2_iterate_dist)."""
        ref = lambda h: C.byref(h) if h is not None else None     # (compact plan: only the halo of t1 is used)
        self.eng._check(self.lib.nss_bpcg2_iterate_dist(C.byref(self.state), dist_handle, ref(halos[0]),
                                                        ref(halos[1]), ref(halos[2]), int(overlap),
                                                        int(it_begin), int(it_end), self.eng.stream))

    def poll(self):
        """Drain the stream; returns (done, it_final, last_it)."""
        done, it_final, last = C.c_int32(), C.c_int32(), C.c_int32()
        self.eng._check(self.lib.nss_bpcg2_poll(C.byref(self.state), C.byref(done), C.byref(it_final),
                                                C.byref(last), self.eng.stream))
        if done.value == 3:
            raise RuntimeError("mailbox transport: a peer did not arrive within the timeout (iteration %d)" % it_final.value)
        if done.value == 2:      # the reference's `alpha = wd / as_s` with as_s == 0 (:226)
            raise ZeroDivisionError("float division by zero (BPCG breakdown <s, K s> = 0 at iteration %d)"
                                    % it_final.value)
        return bool(done.value), it_final.value, last.value

    def history(self, upto):
        return self.eng.to_host(self.hist)[: upto + 1]

    def run(self, wdn, err0, tol, rel_err, maxsteps, poll_every=None):
        """Returns (it, history, converged) -- `it` as the reference's loop variable after
        the loop (index of the iteration whose stop test fired, or maxsteps-1)."""
        poll_every = poll_every or POLL_EVERY
        self.start(wdn, err0, tol, rel_err, maxsteps)
        it, done, it_final = 0, False, 0
        while it < maxsteps:
            end = min(maxsteps, it + poll_every)
            self.enqueue(it, end)
            it = end
            done, it_final, _ = self.poll()
            if done:
                break
        final = it_final if done else maxsteps - 1
        return final, self.history(final), done


class MinresState(C.Structure):
    """ctypes mirror of ``nss_minres_t`` (include/nss_krylov.h)."""
    _fields_ = ([(n, C.c_void_p) for n in ("A", "B", "BT", "pre_diag", "pre_bjac", "pre_amg", "minv")]
                + [("u", C.c_void_p * 2), ("v", (C.c_void_p * 2) * 3), ("w", (C.c_void_p * 2) * 3),
                   ("z", (C.c_void_p * 2) * 2), ("kz", C.c_void_p * 2),
                   ("scal", C.c_void_p), ("ctrl", C.c_void_p), ("hist", C.c_void_p),
                   ("partials_a", C.c_void_p), ("partials_b", C.c_void_p), ("partials_c", C.c_void_p),
                   ("n_u", C.c_int32), ("n_p", C.c_int32), ("local_sums", C.c_int32)])


(M_DELTA, M_GAMMA, M_G2, M_ETA_OLD, M_C_OLD, M_C, M_S_OLD, M_S, M_RES_OLD, M_ERR0, M_TOL) = range(11)


def _block2(v, n_u, n_p):
    return (isinstance(v, BlockVector) and v.nblocks == 2 and _plain(v[0], n_u) and _plain(v[1], n_p))


class MinresLoop:
    """Device-resident iteration of minres.py:96-144 for K = [[A, B^T], [B, None]] and
    C = [[preA, None], [None, preS]] (the operands run.py:45-46 builds)."""

    @classmethod
    def try_create(cls, mat, pre, u, v_ring, w_ring, z_ring, kz):
        if not (isinstance(mat, BlockMatrix) and isinstance(pre, BlockMatrix)):
            return None
        if (mat.nrows, mat.ncols) != (2, 2) or (pre.nrows, pre.ncols) != (2, 2):
            return None
        A, BT, B, C11 = mat[0, 0], mat[0, 1], mat[1, 0], mat[1, 1]
        if C11 is not None or pre[0, 1] is not None or pre[1, 0] is not None:
            return None
        if not all(isinstance(m, SparseMatrix) for m in (A, BT, B)):
            return None
        eng = A.engine
        if not ENABLED or not _hip(eng) or not hasattr(eng.lib, "nss_minres_iterate"):
            return None
        n_u, n_p = A.height, B.height
        if (A.width, B.width, BT.height, BT.width) != (n_u, n_u, n_u, n_p):
            return None
        pa_d, pa_b, ps = native_diag(pre[0, 0]), native_bjac(pre[0, 0]), native_diag(pre[1, 1])
        pa_amg = None
        if pa_d is None and pa_b is None:
            pa_amg, pa_d, pa_b = _amg_plus_jacobi(pre[0, 0])
        if ps is None or (pa_d is None and pa_b is None and pa_amg is None):
            return None
        if pa_b is not None and pa_b[0] != 1.0:
            return None                  # scaled block Jacobi: the protocol path handles it
        vecs = [u, kz] + list(v_ring) + list(w_ring) + list(z_ring)
        if len(v_ring) != 3 or len(w_ring) != 3 or len(z_ring) != 2 or not all(_block2(x, n_u, n_p) for x in vecs):
            return None
        return cls(eng, A, B, BT, pa_d, pa_b, ps, u, v_ring, w_ring, z_ring, kz, pa_amg)

    def __init__(self, eng, A, B, BT, pa_d, pa_b, ps, u, v_ring, w_ring, z_ring, kz, pa_amg=None):
        torch = eng.torch
        self.eng, self.lib = eng, eng.lib
        self.keep = [A, B, BT, pa_d, pa_b, ps, u, v_ring, w_ring, z_ring, kz, pa_amg]
        st = MinresState()
        st.A, st.B, st.BT = A.handle.ptr, B.handle.ptr, BT.handle.ptr
        st.pre_amg = pa_amg.handle.ptr if pa_amg is not None else None

        def scaled(pair):
            scale, op = pair
            return op.d if scale == 1.0 else op.d * scale

        if pa_d is not None:
            self.dinv = scaled(pa_d)
            st.pre_diag, st.pre_bjac = self.dinv.data_ptr(), None
        elif pa_b is not None:
            scale, op = pa_b             # scale == 1.0 (try_create declines anything else)
            st.pre_diag, st.pre_bjac = None, op.handle.ptr
        else:
            st.pre_diag, st.pre_bjac = None, None
        self.minv = scaled(ps)
        st.minv = self.minv.data_ptr()
        for c in range(2):
            st.u[c] = u[c].buf.data_ptr()
            st.kz[c] = kz[c].buf.data_ptr()
            for j in range(3):
                st.v[j][c] = v_ring[j][c].buf.data_ptr()
                st.w[j][c] = w_ring[j][c].buf.data_ptr()
            for j in range(2):
                st.z[j][c] = z_ring[j][c].buf.data_ptr()
        st.n_u, st.n_p = A.height, B.height
        na, nb, nc = C.c_int64(), C.c_int64(), C.c_int64()
        eng._check(self.lib.nss_minres_workspace(C.byref(st), C.byref(na), C.byref(nb), C.byref(nc)))
        self.partials = [eng.zeros(max(1, x.value)) for x in (na, nb, nc)]
        st.partials_a, st.partials_b, st.partials_c = (p.data_ptr() for p in self.partials)
        self.scal = eng.zeros(64)           # two sets of 32 scalars, double-buffered by the parity of k
        self.ctrl = torch.zeros(4, dtype=torch.int32, device=eng.device)
        st.scal, st.ctrl = self.scal.data_ptr(), self.ctrl.data_ptr()
        self.state = st
        self.hist = None

    def run(self, gamma, tol, maxsteps, poll_every=None):
        """Iterations k = 1.. as the reference's while loop.  Returns (errors, hit_relative_tol)."""
        eng, st = self.eng, self.state
        poll_every = poll_every or POLL_EVERY
        self.hist = eng.zeros(maxsteps + 2)
        st.hist = self.hist.data_ptr()
        scal = np.zeros(64)                 # iteration k = 1 reads the first set
        scal[M_GAMMA], scal[M_ETA_OLD], scal[M_C_OLD], scal[M_C] = gamma, gamma, 1.0, 1.0
        scal[M_RES_OLD], scal[M_ERR0], scal[M_TOL] = gamma, gamma, tol
        scal[16:19] = 1.0                   # factors of the (here: normalised) z, v, v_old -- see csrc/minres.hip
        eng.upload(scal, self.scal)
        self.ctrl.zero_()
        stop, k_stop, reason, last = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        k = 1
        while k < maxsteps + 1:
            end = min(maxsteps + 1, k + poll_every)
            eng._check(self.lib.nss_minres_iterate(C.byref(st), k, end, eng.stream))
            k = end
            eng._check(self.lib.nss_minres_poll(C.byref(st), C.byref(stop), C.byref(k_stop), C.byref(reason),
                                                C.byref(last), eng.stream))
            if stop.value:
                break
        last_k = k_stop.value if stop.value else maxsteps
        errors = [1.0] + [float(x) for x in eng.to_host(self.hist)[1: last_k + 1]]
        return errors, bool(stop.value and reason.value == 1)


class Bpcg1State(C.Structure):
    """ctypes mirror of ``nss_bpcg1_t`` (include/nss_krylov.h)."""
    _fields_ = ([(n, C.c_void_p) for n in ("A", "B", "BT", "pre_diag", "pre_bjac", "pre_amg", "minv")]
                + [(n, C.c_void_p * 2) for n in ("x", "r", "d", "a", "t1", "t2")]
                + [("scal", C.c_void_p), ("ctrl", C.c_void_p), ("hist", C.c_void_p),
                   ("partials_a", C.c_void_p), ("partials_b", C.c_void_p), ("partials_c", C.c_void_p),
                   ("k", C.c_double), ("n_u", C.c_int32), ("n_p", C.c_int32), ("local_sums", C.c_int32)])


class Bpcg1Loop:
    """Device-resident iteration of bramble_pasciak_cg.py:110-143."""

    @classmethod
    def try_create(cls, a_matrix, b_matrix, c_matrix, pre_a, pre_s, k, vecs):
        if c_matrix is not None or not (isinstance(a_matrix, SparseMatrix) and isinstance(b_matrix, SparseMatrix)):
            return None
        eng = a_matrix.engine
        if not ENABLED or not _hip(eng) or not hasattr(eng.lib, "nss_bpcg1_iterate"):
            return None
        n_u, n_p = a_matrix.height, b_matrix.height
        if a_matrix.width != n_u or b_matrix.width != n_u:
            return None
        pa_d, pa_b, ps = native_diag(pre_a), native_bjac(pre_a), native_diag(pre_s)
        pa_amg = None
        if pa_d is None and pa_b is None:
            pa_amg, pa_d, pa_b = _amg_plus_jacobi(pre_a)
        if ps is None or (pa_d is None and pa_b is None and pa_amg is None):
            return None
        if any(not _block2(vecs.get(name), n_u, n_p) for name in ("x", "r", "d", "a", "t1", "t2")):
            return None
        return cls(eng, a_matrix, b_matrix, pa_d, pa_b, ps, k, vecs, pa_amg)

    def __init__(self, eng, A, B, pa_d, pa_b, ps, k, vecs, pa_amg=None, BT=None):
        """`BT`: the rows of B^T this process owns, when they are not the transpose of its B (row-partitioned
        runs: distributed.Bpcg1DistLoop)."""
        torch = eng.torch
        self.eng, self.lib = eng, eng.lib
        if BT is None:
            BT = B.CreateTranspose()
        self.keep = [A, B, BT, pa_d, pa_b, ps, vecs, pa_amg]
        st = Bpcg1State()
        st.A, st.B, st.BT = A.handle.ptr, B.handle.ptr, BT.handle.ptr
        st.pre_amg = pa_amg.handle.ptr if pa_amg is not None else None
        scale = 1.0
        if pa_d is not None:
            scale, op = pa_d
            st.pre_diag, st.pre_bjac = op.d.data_ptr(), None
        elif pa_b is not None:
            scale, op = pa_b
            st.pre_diag, st.pre_bjac = None, op.handle.ptr
            # block Jacobi alone: A's row blocks are planned around its blocks (small systems) and the launch of A's rows
            # applies it in its epilogue (csrc/bpcg1.hip: EpiV1Rows)
            if pa_amg is None and hasattr(A.handle, "plan_for_blocks"):
                A.handle.plan_for_blocks(op.handle)
        else:
            st.pre_diag, st.pre_bjac = None, None
        st.k = float(k) * scale
        mscale, mop = ps
        self.minv = mop.d if mscale == 1.0 else mop.d * mscale
        st.minv = self.minv.data_ptr()
        for name in ("x", "r", "d", "a", "t1", "t2"):
            arr = getattr(st, name)
            for c in range(2):
                arr[c] = vecs[name][c].buf.data_ptr()
        st.n_u, st.n_p = A.height, B.height
        na, nb, nc = C.c_int64(), C.c_int64(), C.c_int64()
        eng._check(self.lib.nss_bpcg1_workspace(C.byref(st), C.byref(na), C.byref(nb), C.byref(nc)))
        self.partials = [eng.zeros(max(1, x.value)) for x in (na, nb, nc)]
        st.partials_a, st.partials_b, st.partials_c = (p.data_ptr() for p in self.partials)
        self.scal = eng.zeros(16)
        self.ctrl = torch.zeros(4, dtype=torch.int32, device=eng.device)
        st.scal, st.ctrl = self.scal.data_ptr(), self.ctrl.data_ptr()
        self.state = st
        self.hist = None

    def enqueue(self, it_begin, it_end):
        self.eng._check(self.lib.nss_bpcg1_iterate(C.byref(self.state), it_begin, it_end, self.eng.stream))

    def run(self, rho, err0, tolerance, max_steps, poll_every=None):
        """Returns (errors, converged): errors[i] = err_i/err_0 as appended at :118."""
        eng, st = self.eng, self.state
        poll_every = poll_every or POLL_EVERY
        self.hist = eng.zeros(max(1, max_steps))
        st.hist = self.hist.data_ptr()
        scal = np.zeros(16)
        scal[0], scal[5], scal[6] = rho, err0, tolerance
        eng.upload(scal, self.scal)
        self.ctrl.zero_()
        stop, it_stop, last = C.c_int32(), C.c_int32(), C.c_int32()
        it = 0
        while it < max_steps:
            end = min(max_steps, it + poll_every)
            self.enqueue(it, end)
            it = end
            eng._check(self.lib.nss_bpcg1_poll(C.byref(st), C.byref(stop), C.byref(it_stop), C.byref(last), eng.stream))
            if stop.value:
                break
        count = it_stop.value + 1 if stop.value else max_steps
        return [float(x) for x in eng.to_host(self.hist)[:count]], bool(stop.value)


class CgState(C.Structure):
    """ctypes mirror of ``nss_cg_t`` (include/nss_krylov.h)."""
    _fields_ = ([(n, C.c_void_p) for n in ("A", "pre_diag", "pre_bjac", "pre_amg", "x", "r", "z", "p", "q",
                                           "scal", "ctrl", "hist", "partials_a", "partials_b")]
                + [("n", C.c_int32)])


class CgLoop:
    """Device-resident preconditioned CG (``nss_cg_*``) behind `hipla.CGSolver`."""

    @classmethod
    def try_create(cls, mat, pre):
        if not isinstance(mat, SparseMatrix) or mat.height != mat.width:
            return None
        eng = mat.engine
        if not ENABLED or not _hip(eng) or not hasattr(eng.lib, "nss_cg_iterate"):
            return None
        kind = None
        if pre is None:
            kind = ("none", None)
        elif isinstance(pre, DiagonalMatrix):
            kind = ("diag", pre)
        elif isinstance(pre, (BlockJacobi, BlockGaussSeidel)) and getattr(pre, "middle", None) is None:
            kind = ("bjac", pre)
        elif isinstance(pre, SmoothedAggregationAMG):
            kind = ("amg", pre)
        if kind is None:
            return None
        return cls(eng, mat, kind)

    def __init__(self, eng, mat, kind):
        torch = eng.torch
        self.eng, self.lib, self.mat, self.kind = eng, eng.lib, mat, kind
        n = mat.height
        self.work = {name: eng.zeros(n) for name in ("r", "z", "p", "q")}
        st = CgState()
        st.A, st.n = mat.handle.ptr, n
        st.pre_diag = kind[1].d.data_ptr() if kind[0] == "diag" else None
        st.pre_bjac = kind[1].handle.ptr if kind[0] == "bjac" else None
        st.pre_amg = kind[1].handle.ptr if kind[0] == "amg" else None
        for name, buf in self.work.items():
            setattr(st, name, buf.data_ptr())
        na, nb = C.c_int64(), C.c_int64()
        eng._check(self.lib.nss_cg_workspace(C.byref(st), C.byref(na), C.byref(nb)))
        self.partials = [eng.zeros(max(1, v.value)) for v in (na, nb)]
        st.partials_a, st.partials_b = (p.data_ptr() for p in self.partials)
        self.scal = eng.zeros(8)
        self.ctrl = torch.zeros(4, dtype=torch.int32, device=eng.device)
        st.scal, st.ctrl = self.scal.data_ptr(), self.ctrl.data_ptr()
        self.state = st
        self.hist = None

    def solve(self, b, x, precision, maxsteps, poll_every=None):
        """x = mat^-1 b from x = 0.  Returns (iterations, errors) with errors[0] = err0."""
        from math import sqrt
        eng, st, w = self.eng, self.state, self.work
        poll_every = poll_every or POLL_EVERY
        eng.fill(x, 0.0)
        eng.copy(b, w["r"])
        kind, pre = self.kind
        if kind == "none":
            eng.copy(w["r"], w["z"])
        elif kind == "diag":
            eng.diag_apply(pre.d, 1.0, w["r"], 0.0, w["z"])
        elif kind == "bjac":
            eng.bjac_apply(pre.handle, 1.0, w["r"], 0.0, w["z"])
        else:
            eng.amg_apply(pre.handle, 1.0, w["r"], w["z"])
        eng.copy(w["z"], w["p"])
        rz = eng.dot(w["r"], w["z"])
        err0 = sqrt(abs(rz))
        if err0 == 0.0:
            return 0, [0.0]
        self.hist = eng.zeros(max(1, maxsteps))
        st.hist, st.x = self.hist.data_ptr(), x.data_ptr()
        scal = np.zeros(8)
        scal[0], scal[3], scal[4] = rz, err0, precision
        eng.upload(scal, self.scal)
        self.ctrl.zero_()
        done, it_final, last = C.c_int32(), C.c_int32(), C.c_int32()
        it = 0
        while it < maxsteps:
            end = min(maxsteps, it + poll_every)
            eng._check(self.lib.nss_cg_iterate(C.byref(st), it, end, eng.stream))
            it = end
            eng._check(self.lib.nss_cg_poll(C.byref(st), C.byref(done), C.byref(it_final), C.byref(last), eng.stream))
            if done.value:
                break
        count = it_final.value + 1 if done.value else maxsteps
        return count, [err0] + [float(v) for v in eng.to_host(self.hist)[:count]]
