"""Row-partitioned multi-GPU execution of the Stokes solve path.

One process per GPU.  Rank r owns a contiguous *slab* of rows: the velocity rows of
``A`` and ``B^T``, the pressure rows of ``B`` and the matching slices of every vector;
block-Jacobi blocks and the lumped mass are partition-local.  Two kinds of
communication, both over ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests):

* inner products: local deterministic reduction on the device, then ``all_reduce(SUM)`` of
  one double (2 per BPCG iteration, sequentially dependent);
* SpMV operands: *neighbour halo exchange*, not an all-gather -- a slab boundary is one
  grid plane (about 3 n^2 doubles for A's operand, n^2 for B's and B^T's), three orders of
  magnitude less than the full vector.  Halo entries are appended behind the owned entries
  of the operand (``[owned | from rank 0 | from rank 1 | ...]``), the local CSR blocks are
  renumbered accordingly once at set-up, and one ``all_to_all_single`` with split sizes
  fills the tail directly (zero-copy receive).

The reference is single-process (no MPI / NCCL anywhere); this layer
is new.  Everything here is host logic over the engine interface, so the same code runs
on the numpy checker engine in the gloo tests."""

import os

import numpy as np
import scipy.sparse as sp

from hipla import BaseMatrix, BlockJacobi, BlockMatrix, DiagonalMatrix, InnerProduct, SparseMatrix, Vector
from hipla.engine import get_engine


# --------------------------------------------------------------------------------------
# partition + halo bookkeeping (pure numpy)
# --------------------------------------------------------------------------------------
def even_offsets(n, nranks):
    """Fallback partition for matrices without a slab hint: equal row counts."""
    return np.round(np.arange(nranks + 1) * n / nranks).astype(np.int64)


def localize_rows(mat, row_range, col_offsets, rank, extra_ghosts=None):
    """Cut rows [r0, r1) out of the global CSR `mat` and renumber its columns for `rank`:
    owned columns -> [0, n_owned), ghost columns -> n_owned + position in the ghost list,
    which is sorted by (owner rank, global index).  `extra_ghosts`: global ids to receive as well
    although no row references them.  Returns (local csr, ghost global ids)."""
    r0, r1 = int(row_range[0]), int(row_range[1])
    c0, c1 = int(col_offsets[rank]), int(col_offsets[rank + 1])
    loc = sp.csr_matrix(mat[r0:r1, :])
    loc.sort_indices()
    cols = loc.indices.astype(np.int64)
    owned = (cols >= c0) & (cols < c1)
    ghosts = np.unique(cols[~owned])            # global ids ascending == sorted by owner, then id
    if extra_ghosts is not None and len(extra_ghosts):
        ghosts = np.union1d(ghosts, np.asarray(extra_ghosts, dtype=np.int64))
    ghosts = densify_ghosts(ghosts, col_offsets)
    new = np.empty_like(cols)
    new[owned] = cols[owned] - c0
    new[~owned] = (c1 - c0) + np.searchsorted(ghosts, cols[~owned])
    out = sp.csr_matrix((loc.data, new.astype(np.int32), loc.indptr), shape=(r1 - r0, (c1 - c0) + ghosts.size))
    out.sort_indices()
    return out, ghosts


def densify_ghosts(ghosts, col_offsets, max_waste=2.0):
    """Per owner, replace the set of wanted ids by the contiguous range that covers it when that
    range is at most `max_waste` times larger (slab neighbours want most of one or two grid planes):
    the owner can then send straight out of its vector -- no pack kernel -- and the ghost tail is
    still sorted by (owner, id).  The extra entries are received and never referenced."""
    ghosts = np.asarray(ghosts, dtype=np.int64)
    if ghosts.size == 0:
        return ghosts
    owner = np.searchsorted(np.asarray(col_offsets, dtype=np.int64), ghosts, side="right") - 1
    out = []
    for q in np.unique(owner):
        ids = ghosts[owner == q]
        span = int(ids[-1] - ids[0] + 1)
        out.append(np.arange(ids[0], ids[-1] + 1, dtype=np.int64) if span <= max_waste * ids.size else ids)
    return np.concatenate(out)


class HaloPlan:
    """Who sends what to whom for one operand layout."""

    def __init__(self, rank, nranks, n_owned, ghosts, col_offsets):
        self.rank, self.nranks, self.n_owned = rank, nranks, int(n_owned)
        self.ghosts = np.asarray(ghosts, dtype=np.int64)
        self.n_ghost = int(self.ghosts.size)
        self.col_offsets = np.asarray(col_offsets, dtype=np.int64)
        owner = np.searchsorted(self.col_offsets, self.ghosts, side="right") - 1
        self.recv_counts = np.bincount(owner, minlength=nranks).astype(np.int64)
        self.send_counts = np.zeros(nranks, dtype=np.int64)
        self.send_idx = np.zeros(0, dtype=np.int32)
        self.send_runs = {}          # destination -> first local index, when its entries are one contiguous run
        self.direct = False          # every destination is served by one contiguous run of the owned entries

    def requests(self):
        """ghost ids wanted from each owner (what travels in the set-up all-gather)."""
        cuts = np.concatenate([[0], np.cumsum(self.recv_counts)])
        return [self.ghosts[cuts[q]:cuts[q + 1]] for q in range(self.nranks)]

    def finalize(self, wanted_from_me):
        """`wanted_from_me[q]` = global ids rank q needs from this rank."""
        c0 = int(self.col_offsets[self.rank])
        c1 = int(self.col_offsets[self.rank + 1])
        idx = []
        for q, ids in enumerate(wanted_from_me):
            ids = np.asarray(ids, dtype=np.int64)
            if ids.size and (ids.min() < c0 or ids.max() >= c1):
                raise ValueError("rank %d asked rank %d for entries it does not own" % (q, self.rank))
            self.send_counts[q] = ids.size
            idx.append((ids - c0).astype(np.int32))
            if ids.size and int(ids[-1] - ids[0]) + 1 == ids.size and np.all(np.diff(ids) == 1):
                self.send_runs[q] = int(ids[0] - c0)
        self.send_idx = np.concatenate(idx) if idx else np.zeros(0, dtype=np.int32)
        self.direct = all(q in self.send_runs for q in range(self.nranks) if self.send_counts[q])
        return self


# --------------------------------------------------------------------------------------
# communicators
# --------------------------------------------------------------------------------------
class TorchComm:
    """torch.distributed (RCCL on GPUs, gloo on CPUs).  Buffers are engine buffers: torch
    tensors for the HIP engine, numpy arrays (shared memory with torch) for the checker."""

    def __init__(self, dist, engine=None):
        import torch
        self.dist, self.torch = dist, torch
        self.engine = engine if engine is not None else get_engine()
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        # gloo cannot move device buffers: stage them through the host (used when several
        # ranks share one GPU in the tests; the GPU box runs RCCL, which takes them directly)
        self.stage = dist.get_backend() == "gloo"

    def _t(self, buf):
        return buf if isinstance(buf, self.torch.Tensor) else self.torch.from_numpy(buf)

    def allreduce_sum_into(self, src, dst):
        """dst = sum over ranks of src (out of place; the fused loop's scalars: see
        nss_bpcg2_t.local_sums)."""
        self.engine.copy(src, dst)
        self.allreduce_sum(dst)

    def allreduce_sum(self, buf):
        if self.size == 1:
            return
        t = self._t(buf)
        if self.stage and t.is_cuda:
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)

    def allreduce_scalar(self, value):
        if self.size == 1:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self._scalar_device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def _scalar_device(self):
        return "cpu" if self.stage else getattr(self.engine, "device", "cpu")

    def gather_objects(self, obj):
        out = [None] * self.size
        self.dist.all_gather_object(out, obj)
        return out

    def gather_requests(self, mine, compute_for_rank):
        """requests[q][p] = global ids rank q wants from rank p (set-up only)."""
        return self.gather_objects(mine)

    def exchange(self, plan, sendbuf, ext):
        """Fill ext[n_owned:] with the ghost entries; `sendbuf` already holds the packed
        entries for every destination (ordered by destination rank)."""
        if self.size == 1:
            return                 # (with more ranks the all-to-all is collective: never skip it)
        tail, send = self._t(ext)[plan.n_owned:], self._t(sendbuf)
        outs, ins = [int(c) for c in plan.recv_counts], [int(c) for c in plan.send_counts]
        if self.stage and tail.is_cuda:
            h = self.torch.empty(tail.shape, dtype=tail.dtype)
            self.dist.all_to_all_single(h, send.cpu(), output_split_sizes=outs, input_split_sizes=ins)
            tail.copy_(h)
        else:
            self.dist.all_to_all_single(tail, send, output_split_sizes=outs, input_split_sizes=ins)


class MailboxTransport:
    """The mailbox transport of the native partitioned loops (csrc/p2p.h, `nss_p2p_*`): every rank owns a small
    fine-grained region -- all-reduce mailbox, and per operand LAYOUT the loop exchanges its arrival flags and a landing
    zone for the ghosts -- that its peers map through HIP IPC and write into with plain remote stores over xGMI.
    `halos`: list of (`nss_halo_t`, owned entries), one per layout (BPCG v2: t1; MINRES / BPCG v1: A's operand and B^T's
    operand); the loops may pass copies of these descriptors with only `ext` changed.  Set-up: create the region,
    gather everybody's blob (IPC handle + where each peer's segments are wanted) over the set-up communicator, map the
    peers.  At most 16 ranks."""

    def __init__(self, comm, engine, halos):
        import ctypes as C
        self.engine, self.comm = engine, comm
        self._halos = [h for h, _ in halos]
        nh = len(halos)
        nbytes = C.c_int64()
        engine._check(engine.lib.nss_p2p_blob_bytes(comm.size, nh, C.byref(nbytes)))
        blob = C.create_string_buffer(nbytes.value)
        self.handle = C.c_void_p()
        ptrs = (C.c_void_p * nh)(*[C.addressof(h) for h in self._halos])
        owned = (C.c_int32 * nh)(*[int(n) for _, n in halos])
        engine._check(engine.lib.nss_p2p_create(comm.size, comm.rank, nh, ptrs, owned, C.byref(self.handle), blob))
        blobs = comm.gather_objects(bytes(blob.raw)) if comm.size > 1 else [bytes(blob.raw)]
        engine._check(engine.lib.nss_p2p_connect(self.handle, b"".join(blobs)))

    def attach(self, dist_handle):
        """Route every exchange / one-double all-reduce of the native loops that take `dist_handle` through this transport."""
        self.engine._check(self.engine.lib.nss_dist_attach_p2p(dist_handle, self.handle))

    def allreduce(self, src, dst):
        """dst[0] = sum over the ranks of src[0] (device buffers), the ranks' values added in rank order."""
        self.engine._check(self.engine.lib.nss_p2p_allreduce_f64(self.handle, src.data_ptr(), dst.data_ptr(), self.engine.stream))

    def exchange(self, which=0):
        import ctypes as C
        self.engine._check(self.engine.lib.nss_p2p_exchange(self.handle, C.byref(self._halos[which]), self.engine.stream))

    def timed_out(self):
        import ctypes as C
        out = C.c_int32()
        self.engine._check(self.engine.lib.nss_p2p_error(self.handle, C.byref(out), self.engine.stream))
        return bool(out.value)

    def close(self):
        if self.handle is not None:
            self.engine.lib.nss_p2p_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --------------------------------------------------------------------------------------
# distributed operands
# --------------------------------------------------------------------------------------
class HaloVector(Vector):
    """Owned entries of an SpMV operand; ``ext`` is the underlying buffer
    ``[owned | ghosts]`` the local CSR block indexes into."""

    def __init__(self, dmat):
        eng = dmat.engine
        self.ext = eng.zeros(dmat.plan.n_owned + dmat.plan.n_ghost)
        super().__init__(buf=eng.view(self.ext, 0, dmat.plan.n_owned), engine=eng, comm=dmat.comm)
        self.plan = dmat.plan


class DistSparseMatrix(BaseMatrix):
    """Rows [row_offsets[r], row_offsets[r+1]) of a global CSR matrix on rank r.

    ``Mult`` / ``MultAdd`` = halo exchange of the operand + local CSR SpMV.  Operands that
    are `HaloVector`s of this matrix are exchanged in place; any other vector is first
    copied into the matrix's private operand buffer."""

    def __init__(self, global_csr, row_offsets, col_offsets, comm, engine=None, extra_ghosts=None):
        super().__init__()
        self.comm = comm
        self.engine = engine if engine is not None else get_engine()
        r = comm.rank
        self.row_offsets = np.asarray(row_offsets, dtype=np.int64)
        self.col_offsets = np.asarray(col_offsets, dtype=np.int64)
        self.row_offset = int(self.row_offsets[r])
        self.n_rows = int(self.row_offsets[r + 1] - self.row_offsets[r])
        self.n_cols_owned = int(self.col_offsets[r + 1] - self.col_offsets[r])
        self.global_shape = global_csr.shape
        loc, ghosts = localize_rows(global_csr, (self.row_offsets[r], self.row_offsets[r + 1]), self.col_offsets, r,
                                    extra_ghosts)
        self.local_scipy = loc
        self.local = SparseMatrix.from_scipy(loc, engine=self.engine)
        self.plan = HaloPlan(r, comm.size, self.n_cols_owned, ghosts, self.col_offsets)

        def requests_of(q):
            _, gq = localize_rows(global_csr, (self.row_offsets[q], self.row_offsets[q + 1]), self.col_offsets, q)
            return HaloPlan(q, comm.size, self.col_offsets[q + 1] - self.col_offsets[q], gq, self.col_offsets).requests()

        wanted = comm.gather_requests(self.plan.requests(), requests_of)   # wanted[q][p] = ids q wants from p
        self.plan.finalize([wanted[q][r] for q in range(comm.size)])
        self._send_idx = self.engine.index_buffer(self.plan.send_idx)
        self._sendbuf = self.engine.zeros(max(1, int(self.plan.send_idx.size)))[: int(self.plan.send_idx.size)]
        self._private = None
        self._transpose = None

    # shapes are the LOCAL ones: vectors of this rank have the owned sizes
    def Height(self):
        return self.n_rows

    def Width(self):
        return self.n_cols_owned

    def CreateColVector(self):
        return Vector(self.n_rows, engine=self.engine, comm=self.comm)

    def CreateRowVector(self):
        return Vector(self.n_cols_owned, engine=self.engine, comm=self.comm)

    def operand(self):
        return HaloVector(self)

    def pack(self, hv):
        """Gather the owned entries other ranks need into the contiguous send buffer."""
        if self.plan.send_idx.size:
            self.engine.gather(self._send_idx, hv.buf, self._sendbuf)
        return self._sendbuf

    def exchange(self, hv):
        """Make the ghost tail of `hv.ext` current (pack -> all_to_all -> tail)."""
        if self.comm.size == 1:
            return
        if self.plan.direct and getattr(self.comm, "direct_sends", False):
            self.comm.exchange_direct(self.plan, hv.ext)       # contiguous runs: no pack kernel
            return
        self.comm.exchange(self.plan, self.pack(hv), hv.ext)

    def interior_row_blocks(self):
        """[begin, end) of the longest run of row blocks (launch plan of the local CSR) whose rows
        reference no ghost column: these can be multiplied while the halo is in flight."""
        rb = self.local.handle.row_blocks().astype(np.int64)
        loc = self.local_scipy
        nb = rb.size - 1
        if nb == 0:
            return 0, 0
        ghost_rows = np.zeros(loc.shape[0] + 1, dtype=np.int64)
        rows_with_ghost = np.unique(np.repeat(np.arange(loc.shape[0]), np.diff(loc.indptr))[loc.indices >= self.n_cols_owned])
        ghost_rows[rows_with_ghost + 1] = 1
        csum = np.cumsum(ghost_rows)
        boundary = (csum[rb[1:]] - csum[rb[:-1]]) > 0
        best, cur, best_range = 0, 0, (0, 0)
        for b in range(nb):
            cur = 0 if boundary[b] else cur + 1
            if cur > best:
                best, best_range = cur, (b - cur + 1, b + 1)
        return best_range

    def native_halo(self, hv, interior=None):
        """ctypes `nss_halo_t` for operand `hv` of this matrix (keeps its host arrays alive)."""
        from hipla.fused import HaloStruct
        plan = self.plan
        h = HaloStruct()
        keep = {}
        direct = bool(plan.direct) and bool(plan.send_idx.size)
        sp, so, sc, off = [], [], [], 0
        for q, c in enumerate(plan.send_counts):
            if c:                     # direct: offset into the operand itself; else into the packed send buffer
                sp.append(q), so.append(plan.send_runs[q] if direct else off), sc.append(int(c))
            off += int(c)
        rp, ro, rc, off = [], [], [], plan.n_owned
        for q, c in enumerate(plan.recv_counts):
            if c:
                rp.append(q), ro.append(off), rc.append(int(c))
            off += int(c)
        for name, vals, dt in (("h_send_peer", sp, np.int32), ("h_send_off", so, np.int64), ("h_send_cnt", sc, np.int64),
                               ("h_recv_peer", rp, np.int32), ("h_recv_off", ro, np.int64), ("h_recv_cnt", rc, np.int64)):
            arr = keep[name] = np.ascontiguousarray(vals, dtype=dt)
            setattr(h, name, arr.ctypes.data if arr.size else None)
        h.n_pack, h.n_send, h.n_recv = (0 if direct else int(plan.send_idx.size)), len(sp), len(rp)
        h.direct = 1 if direct else 0
        def ptr(buf):          # device pointer (HIP engine) or host address (numpy checker engine: descriptor tests)
            return buf.data_ptr() if hasattr(buf, "data_ptr") else buf.ctypes.data

        h.send_idx = ptr(self._send_idx) if plan.send_idx.size else None
        h.sendbuf = ptr(self._sendbuf) if plan.send_idx.size else None
        h.ext = ptr(hv.ext)
        if interior is None and not hasattr(self.local.handle, "row_blocks"):
            interior = (0, 0)      # (checker engine: no launch plan)
        h.int_begin, h.int_end = interior if interior is not None else self.interior_row_blocks()
        h._keep = (keep, hv, self)
        return h

    def _operand_for(self, x):
        if isinstance(x, HaloVector) and x.plan is self.plan:
            return x
        if self._private is None:
            self._private = HaloVector(self)
        self.engine.copy(x.buf, self._private.buf)
        return self._private

    def Mult(self, x, y):
        hv = self._operand_for(x)
        self.exchange(hv)
        self.engine.csr_spmv(self.local.handle, 1.0, hv.ext, 0.0, y.buf)

    def MultAdd(self, s, x, y):
        hv = self._operand_for(x)
        self.exchange(hv)
        self.engine.csr_spmv(self.local.handle, float(s), hv.ext, 1.0, y.buf)

    def MultTrans(self, x, y):
        self.CreateTranspose().Mult(x, y)

    def MultTransAdd(self, s, x, y):
        self.CreateTranspose().MultAdd(s, x, y)

    def attach_transpose(self, t):
        self._transpose, t._transpose = t, self

    def CreateTranspose(self):
        if self._transpose is None:
            raise RuntimeError("distributed transpose must be attached at set-up (attach_transpose)")
        return self._transpose

    @property
    def T(self):
        return self.CreateTranspose()


class DistInner:
    """Global inner product: local deterministic dot + all_reduce."""

    def __init__(self, comm):
        self.comm = comm

    def __call__(self, a, b):
        if getattr(a, "comm", None) is not None:       # slabs that know their communicator reduce themselves
            return InnerProduct(a, b)
        return self.comm.allreduce_scalar(InnerProduct(a, b))


class Form:
    def __init__(self, mat):
        self.mat, self.condense = mat, False


class DistributedAMG(BaseMatrix):
    """Smoothed-aggregation V(1,1)-cycle on a row-partitioned operator with *replicated coarse
    levels*: the finest level (smoothing, residual, restriction, prolongation) works on the slab with
    halo exchanges, the restricted residual is all-reduced (one coarse vector, ~9 % of the fine
    size) and every rank runs levels 1.. of the cycle redundantly on its own GPU.  Two halo
    exchanges and one all-reduce per cycle instead of a halo exchange per level and per SpMV -- the
    coarse levels are latency-bound at 8 GPUs either way, and replicating them keeps the hierarchy
    identical to the single-GPU one (`hipla.amg.build_hierarchy` on the global matrix, built on
    every rank).  Applies to slab vectors; use it as `preA` of the solvers on `DistSparseMatrix`
    operands (protocol path)."""

    def __init__(self, global_csr, dist_A, **amg_options):
        super().__init__()
        from hipla.amg import build_hierarchy
        self.A = dist_A
        self.comm, self.engine = dist_A.comm, dist_A.engine
        eng, r = self.engine, dist_A.comm.rank
        glob = SparseMatrix.from_scipy(global_csr, engine=eng)
        self.omega = float(amg_options.get("omega", 2.0 / 3.0))
        levels = build_hierarchy(glob, **amg_options)
        if len(levels) < 2:
            raise ValueError("DistributedAMG: the hierarchy has a single level")
        self.level_sizes = [lv["n"] for lv in levels]
        r0, r1 = int(dist_A.row_offsets[r]), int(dist_A.row_offsets[r + 1])
        P, R = levels[0]["P"].to_scipy(), levels[0]["R"].to_scipy()
        self.P_loc = SparseMatrix.from_scipy(sp.csr_matrix(P[r0:r1, :]), engine=eng)        # owned rows
        self.R_loc = SparseMatrix.from_scipy(sp.csr_matrix(R[:, r0:r1]), engine=eng)        # owned columns
        self.dinv = DiagonalMatrix(self.omega / global_csr.diagonal()[r0:r1], engine=eng)   # w D^-1 on the slab
        self.coarse_levels = levels[1:]
        self.coarse = eng.amg_create(self.coarse_levels, self.omega)
        nc = levels[1]["n"]
        self.rc, self.ec = Vector(nc, engine=eng), Vector(nc, engine=eng)
        self.x0, self.res = dist_A.CreateRowVector(), dist_A.CreateColVector()
        self.n = r1 - r0
        self._native = None

    def native_handle(self, dist_handle):
        """`nss_dist_amg_t` of this cycle for the native partitioned loops (created once): the C loop then
        issues the cycle's two halo exchanges and its coarse all-reduce itself."""
        import ctypes as C
        if self._native is None:
            eng = self.engine
            self._x_native = self.A.operand()                      # the iterate, halo-extended
            halo = self.A.native_halo(self._x_native)
            out = C.c_void_p()
            eng._check(eng.lib.nss_dist_amg_create(dist_handle, self.A.local.handle.ptr, C.byref(halo),
                                                   self.R_loc.handle.ptr, self.P_loc.handle.ptr,
                                                   self.dinv.d.data_ptr(), self.coarse.ptr, C.byref(out)))
            self._native = (out, halo, dist_handle)
        return self._native[0]

    def native_apply(self, scale, b, y):
        """y = scale * V(b) through the native handle (tests)."""
        eng = self.engine
        eng._check(eng.lib.nss_dist_amg_apply_f64(self._native[0], float(scale), b.buf.data_ptr(), y.buf.data_ptr(),
                                                  eng.stream))

    def __del__(self):
        try:
            if self._native is not None:
                self.engine.lib.nss_dist_amg_destroy(self._native[0])
                self._native = None
        except Exception:
            pass

    def Height(self):
        return self.n

    def Width(self):
        return self.n

    def CreateColVector(self):
        return self.A.CreateColVector()

    CreateRowVector = CreateColVector

    def Mult(self, b, y):
        x0, res, rc, ec = self.x0, self.res, self.rc, self.ec
        x0.data = self.dinv * b                             # pre-smoothing from zero
        res.data = b - self.A * x0                          # halo exchange 1
        rc.data = self.R_loc * res                          # this slab's share of the coarse residual
        self.comm.allreduce_sum(rc.buf)
        self.engine.amg_apply(self.coarse, 1.0, rc.buf, ec.buf)     # levels 1.. on every rank
        x0.data += self.P_loc * ec
        res.data = b - self.A * x0                          # halo exchange 2
        y.data = x0 + self.dinv * res                       # post-smoothing

    MultTrans = Mult

    @property
    def T(self):
        return self


class DistributedAuxiliary(BaseMatrix):
    """The auxiliary-space term ``transform @ preAh1 @ transform.T`` of the reference's MypreA
    (templates/NavierStokesSIMPLE_iterative.py:336-337,357,380,383) on slabs: the stacked nodal space in slab-major
    order (`StokesSystem.auxiliary_space_stacked`), the vertex plane between the cell slabs k and k + 1 owned by the
    rank of slab k; ``transform`` and its transpose as `DistSparseMatrix` (one neighbour plane of halo each) and ONE
    smoothed-aggregation V-cycle on the block-diagonal nodal Laplacian with replicated coarse levels
    (`DistributedAMG`).  Protocol operator on slab vectors; `native_handle` gives the `nss_dist_aux_t` the native
    partitioned loop applies itself."""

    def __init__(self, ops, **amg_options):
        super().__init__()
        sysm, comm, eng = ops.sysm, ops.comm, ops.engine
        self.ops, self.comm, self.engine = ops, comm, eng
        st = sysm.auxiliary_space_stacked()
        slab = np.searchsorted(sysm.velocity_slab_offsets, ops.vel)            # slab index of every partition cut
        if not np.array_equal(sysm.velocity_slab_offsets[slab], ops.vel):
            raise ValueError("DistributedAuxiliary: the row partition must cut between grid slabs")
        self.node_offsets = np.asarray(st["node_slab_offsets"], dtype=np.int64)[slab]
        T = sp.csr_matrix(st["transform"])
        TT = T.T.tocsr()
        TT.sort_indices()
        self.transform = DistSparseMatrix(T, ops.vel, self.node_offsets, comm, eng)
        self.transform_t = DistSparseMatrix(TT, self.node_offsets, ops.vel, comm, eng)
        self.L = DistSparseMatrix(st["laplacian"], self.node_offsets, self.node_offsets, comm, eng)
        self.V = DistributedAMG(st["laplacian"], self.L, **amg_options)
        self.level_sizes = self.V.level_sizes
        self.n = ops.n_u
        self._r, self._e = self.transform_t.CreateColVector(), self.transform_t.CreateColVector()
        self._native = None

    def Height(self):
        return self.n

    def Width(self):
        return self.n

    def CreateColVector(self):
        return self.ops.A.CreateColVector()

    CreateRowVector = CreateColVector

    def Mult(self, x, y):
        self._r.data = self.transform_t * x            # transform.T  (halo exchange of x)
        self._e.data = self.V * self._r       # the V-cycle on the stacked nodal Laplacian
        y.data = self.transform * self._e             # transform    (halo exchange of e)

    MultTrans = Mult

    @property
    def T(self):
        return self

    def native_handle(self, dist_handle, t1=None):
        """`nss_dist_aux_t` (created once).  `t1`: the loop's iterate as A's operand (`HaloVector`), needed by the
        multiplicative MypreA."""
        import ctypes as C
        if self._native is None:
            eng = self.engine
            self._x_native, self._e_native = self.transform_t.operand(), self.transform.operand()
            hx, he = self.transform_t.native_halo(self._x_native), self.transform.native_halo(self._e_native)
            hy = self.ops.A.native_halo(t1) if t1 is not None else None
            out = C.c_void_p()
            eng._check(eng.lib.nss_dist_aux_create(dist_handle, self.transform_t.local.handle.ptr, C.byref(hx),
                                                   self.transform.local.handle.ptr, C.byref(he), self.V.native_handle(dist_handle),
                                                   C.byref(hy) if hy is not None else None, C.byref(out)))
            self._native = (out, hx, he, hy, dist_handle)
        return self._native[0]

    def native_apply(self, scale, b, y):
        eng = self.engine
        eng._check(eng.lib.nss_dist_aux_apply_f64(self._native[0], float(scale), b.buf.data_ptr(), y.buf.data_ptr(), eng.stream))

    def release(self):
        if self._native is not None:
            self.engine.lib.nss_dist_aux_destroy(self._native[0])
            self._native = None
        if getattr(self.V, "_native", None) is not None:
            self.engine.lib.nss_dist_amg_destroy(self.V._native[0])
            self.V._native = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class DistributedMypreA(BaseMatrix):
    """``MypreA.Mult`` (templates/NavierStokesSIMPLE_iterative.py:375-383) on slabs, protocol form: `gs` = the block
    smoother of the slab's diagonal block (sweeps inside the slab, additive across slabs), `aux` = `DistributedAuxiliary`.
    GS=True: y = 0; Smooth; r = x - A y with the PARTITIONED A; y += aux r; SmoothBack.  GS=False: y = (aux + J) x."""

    def __init__(self, ops, gs, aux, GS=True):
        super().__init__()
        self.ops, self.gs, self.aux, self.GS = ops, gs, aux, GS
        self._res = ops.A.CreateColVector()

    def Height(self):
        return self.ops.n_u

    def Width(self):
        return self.ops.n_u

    def CreateColVector(self):
        return self.ops.A.CreateColVector()

    CreateRowVector = CreateColVector

    def Mult(self, x, y):
        if self.GS:
            y[:] = 0.0                                   # :377
            self.gs.Smooth(y, x)                         # :378
            self._res.data = x - self.ops.A * y          # :379
            y.data += self.aux * self._res               # :380
            self.gs.SmoothBack(y, x)                     # :381
        else:
            y.data = self.aux * x + self.gs * x          # :383

    MultTrans = Mult

    @property
    def T(self):
        return self


class DistributedStokes:
    """The operands of the Stokes solve on this rank: A, B, B^T as `DistSparseMatrix`,
    block-Jacobi / Jacobi preA (or, `pre="amg"`, the `DistributedAMG` cycle) and lumped-mass preM
    restricted to the slab."""

    def __init__(self, sysm, blocks, comm, engine=None, partition=None, pre=None, aux_options=None):
        self.comm = comm
        self.engine = engine if engine is not None else get_engine()
        r, size = comm.rank, comm.size
        vel, prs = partition if partition is not None else sysm.partition(size)
        self.vel, self.prs = np.asarray(vel, dtype=np.int64), np.asarray(prs, dtype=np.int64)
        self.sysm = sysm
        self.n_u, self.n_p = int(self.vel[r + 1] - self.vel[r]), int(self.prs[r + 1] - self.prs[r])
        BT = sysm.B.T.tocsr()
        BT.sort_indices()
        # Ghost pressure cells of B^T's operand and the velocity dofs their rows of B touch: B's and A's
        # operands receive those as well, so that the fused loop can keep s1 and t4 current on the
        # ghosts by redundant computation instead of exchanging them (nss_bpcg2_t.ghost_*).
        v0g, v1g = int(self.vel[r]), int(self.vel[r + 1])
        _, ghost_p = localize_rows(BT, (v0g, v1g), self.prs, r)
        rows_gp = sp.csr_matrix(sysm.B[ghost_p, :]) if ghost_p.size else sp.csr_matrix((0, sysm.B.shape[1]))
        touched = np.unique(rows_gp.indices.astype(np.int64))
        extra_v = touched[(touched < v0g) | (touched >= v1g)]
        self.B = DistSparseMatrix(sysm.B, self.prs, self.vel, comm, self.engine, extra_ghosts=extra_v)
        # A's operand also receives the ghost columns of B's operand: the fused loop then derives the
        # ghosts of t4 = t1 - s0 locally instead of exchanging them (nss_bpcg2_t.ghost_mode)
        self.A = DistSparseMatrix(sysm.A, self.vel, self.vel, comm, self.engine, extra_ghosts=self.B.plan.ghosts)
        # rows of B of the ghost pressure cells, columns in the layout of B's operand
        cols = rows_gp.indices.astype(np.int64)
        own = (cols >= v0g) & (cols < v1g)
        newc = np.empty_like(cols)
        newc[own] = cols[own] - v0g
        pos = np.searchsorted(self.B.plan.ghosts, cols[~own])
        if cols[~own].size and not np.array_equal(self.B.plan.ghosts[np.minimum(pos, self.B.plan.ghosts.size - 1)],
                                                  cols[~own]):
            raise RuntimeError("ghost rows of B reference columns outside B's operand")
        newc[~own] = self.B.plan.n_owned + pos
        self.ghost_p = ghost_p
        self._rows_gp, self._v_range = rows_gp, (v0g, v1g)
        self._b_ext = None
        self.ghost_rows_B = sp.csr_matrix((rows_gp.data, newc.astype(np.int32), rows_gp.indptr),
                                          shape=(ghost_p.size, self.B.plan.n_owned + self.B.plan.n_ghost))
        self.ghost_rows_B.sort_indices()
        self.ghost_minv = 1.0 / sysm.mass[ghost_p] if ghost_p.size else np.zeros(0)
        self.BT = DistSparseMatrix(BT, self.vel, self.prs, comm, self.engine)
        self.B.attach_transpose(self.BT)
        v0, v1 = int(self.vel[r]), int(self.vel[r + 1])
        # block-Jacobi on the diagonal block of the slab (blocks never straddle slabs)
        a_diag = sp.csr_matrix(sysm.A[v0:v1, v0:v1])
        a_diag.sort_indices()
        self.A_diag = SparseMatrix.from_scipy(a_diag, engine=self.engine)
        if blocks is not None:
            blocks = np.asarray(blocks, dtype=np.int64)
            live = blocks >= 0
            first = np.where(live, blocks, np.int64(1) << 62).min(axis=0)
            last = blocks.max(axis=0)
            mine = (first >= v0) & (first < v1)
            if np.any(mine & ~(last < v1)):
                raise ValueError("a block-Jacobi block straddles the slab boundary")
            loc = blocks[:, mine]
            loc = np.where(loc >= 0, loc - v0, -1).astype(np.int32)
            self.local_blocks = np.ascontiguousarray(loc)
            if pre in ("bgs", "mypre_a"):
                # multicolour block Gauss-Seidel INSIDE the slab, additive across slabs (a "hybrid" sweep: no exchange
                # inside a sweep; with one slab it is the single-GPU sweep).  Colours: first fit in block order on the
                # slab's own block graph -- what the same colouring gives on the slab-block-diagonal global graph.
                from hipla import BlockGaussSeidel
                self.preA = BlockGaussSeidel(self.A_diag, self.local_blocks)
            else:
                self.preA = BlockJacobi(self.A_diag, self.local_blocks)
        else:
            from hipla import JacobiPreconditioner
            self.preA = JacobiPreconditioner(self.A_diag)
        if pre == "amg":
            self.preA = DistributedAMG(sysm.A, self.A)
        self.gs = self.aux = None
        if pre == "mypre_a":
            # the reference's default preA on slabs: MypreA(GS=True) with the auxiliary-space term
            if blocks is None:
                raise ValueError("pre='mypre_a' needs the facet blocks")
            self.gs = self.preA
            self.aux = DistributedAuxiliary(self, **(aux_options or {}))
            self.preA = DistributedMypreA(self, self.gs, self.aux, GS=True)
        p0, p1 = int(self.prs[r]), int(self.prs[r + 1])
        self.preM = DiagonalMatrix(1.0 / sysm.mass[p0:p1], engine=self.engine)
        self.inner = DistInner(comm)

    def local_slices(self):
        r = self.comm.rank
        return slice(int(self.vel[r]), int(self.vel[r + 1])), slice(int(self.prs[r]), int(self.prs[r + 1]))

    def compact_layout_ok(self):
        """The compact partitioned plan keeps every ghost by recurrence: B's ghost columns must be among those
        of A's operand and the ghost pressure cells must be exactly the ghosts of B^T's operand (both arranged
        by the constructor; checked because a caller may pass its own partition)."""
        gb, ga = self.B.plan.ghosts, self.A.plan.ghosts
        return bool(np.all(np.isin(gb, ga)) and np.array_equal(self.ghost_p, self.BT.plan.ghosts))

    def b_extended_scipy(self):
        """Rows [this slab's pressure rows | rows of the ghost pressure cells of B^T's operand] of B with the
        columns numbered in the layout of A's operand ([owned | A's ghosts]): what the compact partitioned
        plan multiplies with `t1 - s0` formed on the fly (nss_bpcg2_t.dist_compact)."""
        v0, v1 = self._v_range
        n_u, ga = self.A.plan.n_owned, self.A.plan.ghosts

        def on_a(cols_global):
            cols = np.asarray(cols_global, dtype=np.int64)
            own = (cols >= v0) & (cols < v1)
            out = np.empty_like(cols)
            out[own] = cols[own] - v0
            pos = np.searchsorted(ga, cols[~own])
            if cols[~own].size and (pos.max(initial=0) >= ga.size or not np.array_equal(ga[np.minimum(pos, ga.size - 1)], cols[~own])):
                raise RuntimeError("a row of B references a column outside A's operand")
            out[~own] = n_u + pos
            return out.astype(np.int32)

        loc = self.B.local_scipy                                    # columns in B's layout -> global -> A's layout
        cols_b = loc.indices.astype(np.int64)
        glob = np.where(cols_b < self.B.plan.n_owned, cols_b + v0,
                        self.B.plan.ghosts[np.maximum(cols_b - self.B.plan.n_owned, 0)] if self.B.plan.n_ghost else 0)
        width = n_u + ga.size
        own_rows = sp.csr_matrix((loc.data, on_a(glob), loc.indptr), shape=(loc.shape[0], width))
        gp = self._rows_gp
        ghost_rows = sp.csr_matrix((gp.data, on_a(gp.indices), gp.indptr), shape=(gp.shape[0], width))
        ext = sp.vstack([own_rows, ghost_rows]).tocsr()
        ext.sort_indices()
        return ext

    def b_extended(self):
        """`b_extended_scipy()` on the device; no row block of its launch plan spans the owned / ghost boundary."""
        if self._b_ext is None:
            ext = self.b_extended_scipy()
            handle = self.engine.csr_create(ext.shape[0], ext.shape[1], ext.indptr, ext.indices, ext.data, cuts=[self.n_p])
            self._b_ext = SparseMatrix(ext.shape[0], ext.shape[1], ext.indptr, ext.indices, ext.data, engine=self.engine,
                                       handle=handle)
        return self._b_ext

    def vectors(self, f_global, g_global):
        """This rank's slabs of a global (velocity, pressure) pair as vectors that know the
        communicator: `InnerProduct` / `Norm` of them -- and of every vector the solvers create from
        them with `CreateVector()` -- are global, so `MinRes`, `bramble_pasciak_cg`,
        `BramblePasciakCG` and `CGSolver` run unchanged on the partitioned operands."""
        us, ps = self.local_slices()
        fv = Vector.from_numpy(np.asarray(f_global)[us], engine=self.engine)
        gv = Vector.from_numpy(np.asarray(g_global)[ps], engine=self.engine)
        fv.comm = gv.comm = self.comm
        return fv, gv

    def halo_doubles(self):
        return {"A_operand": self.A.plan.n_ghost, "B_operand": self.B.plan.n_ghost,
                "BT_operand": self.BT.plan.n_ghost}


class DistributedBpcg2:
    """Row-partitioned Bramble-Pasciak CG (v2) on this rank: set-up through the operator
    protocol with distributed operands (halo + all_reduce inside ``Mult`` / inner product),
    iteration through the fused device phases (``nss_bpcg2_phase``) with one halo
    exchange (t1) and two all-reduces in between; the ghosts of the other two SpMV operands are kept
    current by redundant computation (nss_bpcg2_t.ghost_*)."""

    # (kind, argument): device phases between two communication points go down in one C call
    SCHEDULE = (("halo", "s1"), ("phases", ("K1", "K1")), ("halo", "t1"), ("phases", ("K2", "K2")),
                ("halo", "t4"), ("phases", ("K3", "SUM1")), ("allreduce", 1), ("phases", ("ALPHA", "SUM2")),
                ("allreduce", 2), ("phases", ("BETA", "K5")))
    # the compact plan (default): 6 launches + 3 collectives per iteration instead of 9 + 3
    SCHEDULE_COMPACT = (("cphases", ("C1", "C1")), ("halo", "t1"), ("cphases", ("C23", "SUMA")), ("allreduce", 1),
                        ("cphases", ("C4", "SUMW")), ("allreduce", 2))

    def __init__(self, sysm, f, g, blocks, dist, engine=None, comm=None, quiet=True, native=True, pre=None, plan=None,
                 aux_options=None, transport=None):
        """`native=False` keeps the Python-driven schedule even when `comm` is an `RcclComm` (its
        collectives are then single ctypes calls into librccl between the device phases).
        `pre="amg"`: preA = the V-cycle with replicated coarse levels (`DistributedAMG`), applied inside the
        native loop (needs the RCCL communicator: the cycle's exchanges and its coarse all-reduce are issued
        from C); `pre="amg+bjac"` adds the block Jacobi (additive MypreA).
        `pre="bgs"`: multicolour block Gauss-Seidel inside the slab, additive across slabs (no communication inside a
        sweep; any communicator).  `pre="mypre_a"`: the reference's default -- MypreA(GS=True): those sweeps around the
        auxiliary-space term on slabs (`DistributedAuxiliary`; `aux_options` go to its V-cycle), the residual between
        them with the partitioned A; applied natively inside the loop (RCCL communicator, as `pre="amg"`).
        `plan`: "compact" (default; NSS_DIST_PLAN overrides) = C1 / preA / exchange / C23 / sum / all-reduce / C4 /
        sum / all-reduce with every ghost kept by recurrence behind the owned entries of its vector; "classic" = the
        eight-phase form (the only one with the interior / boundary overlap)."""
        import contextlib
        self.want_native = bool(native)
        self.want_transport = transport          # "mailbox": csrc/p2p.h instead of RCCL inside the iterations
        import io
        from hipla import BlockVector
        from solvers.bramblepasciak_new import BpcgSession
        self.engine = engine if engine is not None else get_engine()
        self.comm = comm if comm is not None else TorchComm(dist, self.engine)
        ops = self.ops = DistributedStokes(sysm, blocks, self.comm, self.engine,
                                           pre=pre if pre in ("bgs", "mypre_a") else None, aux_options=aux_options)
        self.dist_amg = None
        if pre in ("amg", "amg+bjac"):
            self.dist_amg = DistributedAMG(sysm.A, ops.A)
            self.jacobi_part = ops.preA if pre == "amg+bjac" else None
            ops.preA = self.dist_amg if self.jacobi_part is None else self.dist_amg + self.jacobi_part
        us, ps = ops.local_slices()
        fv = Vector.from_numpy(np.asarray(f)[us], engine=self.engine)
        gv = Vector.from_numpy(np.asarray(g)[ps], engine=self.engine)
        self.sol = BlockVector([Vector(ops.n_u, engine=self.engine), Vector(ops.n_p, engine=self.engine)])
        # operands of the three SpMVs of the loop live in halo-extended buffers
        self.t1, self.t4, self.s1 = ops.A.operand(), ops.B.operand(), ops.BT.operand()
        plan = plan or os.environ.get("NSS_DIST_PLAN", "compact")
        if plan not in ("compact", "classic"):
            raise ValueError("plan must be 'compact' or 'classic'")
        self.compact = plan == "compact" and ops.compact_layout_ok() and os.environ.get("NSS_GHOST_T4", "1") == "1" \
            and os.environ.get("NSS_GHOST_S1", "1") == "1"
        workspace = dict(t1=self.t1, t4=self.t4, s1=self.s1)
        if self.compact:       # ghost copies behind the owned entries: s0, w0 like A's operand, w1, t3 like B^T's
            workspace.update(s0=ops.A.operand(), w0=ops.A.operand(), w1=ops.BT.operand(), t3=ops.BT.operand())
        sink = io.StringIO() if quiet or self.comm.rank != 0 else None
        with (contextlib.redirect_stdout(sink) if sink is not None else contextlib.nullcontext()):
            ses = BpcgSession(Form(ops.A), Form(ops.B), None, fv, gv, ops.preA, ops.preM, sol=self.sol,
                              initialize=True, inner=ops.inner, workspace=workspace)
        self.k, self.wdn, self.err0 = ses.k, ses.wdn, ses.err0
        self.first_direction = ses.first_direction
        self._attach(dict(u0=ses.u[0], u1=ses.u[1], d0=ses.d[0], d1=ses.d[1], w0=ses.w[0], w1=ses.w[1],
                          s0=ses.s[0], s1=self.s1, z0=ses.z[0], q=ses.As0, t0=ses.t0, t1=self.t1, t2=ses.t2,
                          t3=ses.t3, t4=self.t4))
        self.ses = ses

    @classmethod
    def from_state(cls, ops, k, wdn, err0, vecs):
        """Attach the fused loop to an already prepared state (`vecs`: the owned slices of
        u, d, w, s, z0, q = A s0; t1 / t4 / s1 must be `HaloVector`s of ops.A / ops.B / ops.BT)."""
        self = cls.__new__(cls)
        self.engine, self.comm, self.ops = ops.engine, ops.comm, ops
        self.k, self.wdn, self.err0 = k, wdn, err0
        self.t1, self.t4, self.s1 = vecs["t1"], vecs["t4"], vecs["s1"]
        self.compact = False
        self.first_direction = lambda: None
        self._attach(vecs)
        return self

    def _attach(self, vecs):
        from hipla.fused import Bpcg2Loop
        ops = self.ops
        self.vecs = vecs
        dist_amg = getattr(self, "dist_amg", None)
        if dist_amg is not None:
            import ctypes as C
            comm_handle = getattr(self.comm, "comm", None)
            if comm_handle is None:
                raise RuntimeError("pre='amg' inside the fused partitioned loop needs the RCCL communicator "
                                   "(with torch.distributed use BramblePasciakCG on the distributed operands)")
            handle = C.c_void_p()
            self.engine._check(self.engine.lib.nss_dist_create(comm_handle, self.comm.size, self.comm.rank, C.byref(handle)))
            self._amg_dist_handle = handle
        compact = getattr(self, "compact", False)
        matB = ops.b_extended() if compact else ops.B.local
        extra = dict(ghost_rows_b=int(ops.BT.plan.n_ghost)) if compact else {}
        if getattr(ops, "aux", None) is not None:           # MypreA(GS=True) on slabs, natively inside the loop
            import ctypes as C
            comm_handle = getattr(self.comm, "comm", None)
            if comm_handle is None:
                raise RuntimeError("pre='mypre_a' inside the fused partitioned loop needs the RCCL communicator "
                                   "(with torch.distributed use BramblePasciakCG on the distributed operands)")
            handle = C.c_void_p()
            self.engine._check(self.engine.lib.nss_dist_create(comm_handle, self.comm.size, self.comm.rank, C.byref(handle)))
            self._amg_dist_handle = handle
            self.loop = Bpcg2Loop.try_create(ops.A.local, matB, ops.BT.local, ops.gs, self.k, ops.preM, vecs,
                                             distributed=True, dist_aux=ops.aux.native_handle(handle, vecs["t1"]), **extra)
            if self.loop is not None:
                self.loop.keep.append(ops.aux)
        elif dist_amg is not None:
            self.loop = Bpcg2Loop.try_create(ops.A.local, matB, ops.BT.local, self.jacobi_part, self.k, ops.preM,
                                             vecs, distributed=True, dist_amg=dist_amg.native_handle(self._amg_dist_handle),
                                             **extra)
            self.loop.keep.append(dist_amg)        # the object, not just its raw handle
        else:
            self.loop = Bpcg2Loop.try_create(ops.A.local, matB, ops.BT.local, ops.preA, self.k, ops.preM, vecs,
                                             distributed=True, **extra)
        if self.loop is None:
            raise RuntimeError("fused distributed BPCG loop needs the HIP engine and native operands")
        self.halo = {"s1": (ops.BT, self.s1), "t1": (ops.A, self.t1), "t4": (ops.B, self.t4)}
        if compact:
            self.ghost_mode = self._setup_ghosts_compact()
        else:
            self.ghost_mode = os.environ.get("NSS_GHOST_T4", "1") == "1" and self._setup_ghosts()
        self.native = None
        # 0: exchange, then one launch per SpMV, all on the compute stream.  1: exchange on a second
        # stream while the interior row blocks are multiplied.  Measured on one GPU at 1/8 of the
        # headline size (tools/partition_overhead.py): the split launches + cross-stream events of
        # mode 1 cost 98 us per iteration -- more than the three small exchanges they hide.
        self.overlap = int(os.environ.get("NSS_OVERLAP", "0"))
        comm_handle = getattr(self.comm, "comm", None)         # RcclComm: an ncclComm_t
        if comm_handle is not None and getattr(self, "want_native", True) and hasattr(self.loop.lib, "nss_bpcg2_iterate_dist"):
            self.enable_native(comm_handle)
        self.mailbox = None
        if getattr(self, "want_transport", None) == "mailbox":
            self.enable_mailbox()

    def _setup_ghosts(self):
        """Ghost copies of s0 / w0 on the ghost columns of B's operand (nss_bpcg2_t.ghost_*): every
        ghost of B must also be a ghost of A's operand, which DistributedStokes arranges."""
        ops, eng = self.ops, self.engine
        gb, ga = ops.B.plan.ghosts, ops.A.plan.ghosts
        pos = np.searchsorted(ga, gb)
        if gb.size and (pos.max(initial=0) >= ga.size or not np.array_equal(ga[np.minimum(pos, ga.size - 1)], gb)):
            return False
        self._ghost_map = eng.index_buffer((ops.A.plan.n_owned + pos).astype(np.int32))
        self._ghost_s0 = eng.zeros(max(1, gb.size))
        self._ghost_w0 = eng.zeros(max(1, gb.size))
        st = self.loop.state
        st.ghost_mode, st.ghost_n = 1, int(gb.size)
        st.ghost_map = self._ghost_map.data_ptr()
        st.ghost_s0, st.ghost_w0 = self._ghost_s0.data_ptr(), self._ghost_w0.data_ptr()
        self._ghost_tmp = ops.B.operand()
        # pressure part: s1 on the ghost cells of B^T's operand (nss_bpcg2_t.ghost_p_*)
        self.ghost_p_mode = False
        gp = getattr(ops, "ghost_p", None)
        if (os.environ.get("NSS_GHOST_S1", "1") == "1" and gp is not None
                and np.array_equal(gp, ops.BT.plan.ghosts)):
            self._ghost_b = SparseMatrix.from_scipy(ops.ghost_rows_B, engine=eng) if gp.size else None
            self._ghost_t3 = eng.zeros(max(1, gp.size))
            self._ghost_w1 = eng.zeros(max(1, gp.size))
            self._ghost_minv = eng.from_host(ops.ghost_minv) if gp.size else eng.zeros(1)
            st.ghost_p_mode, st.ghost_p_n = 1, int(gp.size)
            st.ghost_b = self._ghost_b.handle.ptr if gp.size else None
            st.ghost_t3, st.ghost_w1 = self._ghost_t3.data_ptr(), self._ghost_w1.data_ptr()
            st.ghost_minv = self._ghost_minv.data_ptr()
            self._ghost_tmp_p = ops.BT.operand()
            self.ghost_p_mode = True
        return True

    def _setup_ghosts_compact(self):
        """nss_bpcg2_t.dist_compact: the ghost copies are the tails of the vectors themselves."""
        ops, eng, st, v = self.ops, self.engine, self.loop.state, self.vecs
        n_u, n_p = ops.n_u, ops.n_p
        for name, mat in (("s0", ops.A), ("w0", ops.A), ("t1", ops.A), ("s1", ops.BT), ("w1", ops.BT), ("t3", ops.BT)):
            if not (isinstance(v[name], HaloVector) and v[name].plan is mat.plan):
                raise RuntimeError("compact partitioned plan: %s must be an operand buffer of %s" % (name, "A" if mat is ops.A else "B^T"))
        self._ghost_minv = eng.from_host(ops.ghost_minv) if ops.ghost_p.size else eng.zeros(1)
        st.dist_compact = 1
        st.ghost_mode, st.ghost_n, st.ghost_map = 1, int(ops.A.plan.n_ghost), None
        st.ghost_s0 = v["s0"].ext.data_ptr() + 8 * n_u
        st.ghost_w0 = v["w0"].ext.data_ptr() + 8 * n_u
        st.ghost_p_mode, st.ghost_p_n, st.ghost_b = 1, int(ops.BT.plan.n_ghost), None
        st.ghost_t3 = v["t3"].ext.data_ptr() + 8 * n_p
        st.ghost_w1 = v["w1"].ext.data_ptr() + 8 * n_p
        st.ghost_minv = self._ghost_minv.data_ptr()
        self.ghost_p_mode = True
        return True

    def _fill_ghosts(self):
        """Initial values of the ghost copies: one exchange each of s0 and w0 over B's halo plan."""
        ops, eng = self.ops, self.engine
        if getattr(self, "compact", False):      # the vectors are operand buffers: exchange them in place, once
            v = self.vecs
            for name, mat in (("s0", ops.A), ("w0", ops.A), ("s1", ops.BT), ("w1", ops.BT)):
                mat.exchange(v[name])
            return
        n_own, n_g = ops.B.plan.n_owned, ops.B.plan.n_ghost
        for src, dst in ((self.vecs["s0"], self._ghost_s0), (self.vecs["w0"], self._ghost_w0)):
            eng.copy(src.buf, self._ghost_tmp.buf)
            ops.B.exchange(self._ghost_tmp)
            if n_g:
                eng.copy(eng.view(self._ghost_tmp.ext, n_own, n_own + n_g), eng.view(dst, 0, n_g))
        if getattr(self, "ghost_p_mode", False):
            n_own, n_g = ops.BT.plan.n_owned, ops.BT.plan.n_ghost
            ops.BT.exchange(self.s1)                                   # s1's ghost tail: once per solve
            eng.copy(self.vecs["w1"].buf, self._ghost_tmp_p.buf)
            ops.BT.exchange(self._ghost_tmp_p)
            if n_g:
                eng.copy(eng.view(self._ghost_tmp_p.ext, n_own, n_own + n_g), eng.view(self._ghost_w1, 0, n_g))

    def enable_native(self, comm_handle, interior=None):
        """Issue the partitioned iterations from C (nss_bpcg2_iterate_dist): RCCL calls, halo
        packs, events and the interior/boundary split without Python in the loop."""
        import ctypes as C
        self.close()
        handle = C.c_void_p()
        eng = self.engine
        eng._check(eng.lib.nss_dist_create(comm_handle, self.comm.size, self.comm.rank, C.byref(handle)))
        ops = self.ops
        interior = interior or {}
        halos = (ops.BT.native_halo(self.s1, interior.get("s1")), ops.A.native_halo(self.t1, interior.get("t1")),
                 ops.B.native_halo(self.t4, interior.get("t4")))      # (the compact plan uses the middle one only)
        self.native = (handle, halos)

    def enable_mailbox(self):
        """Run the native compact loop over the mailbox transport (`MailboxTransport`): the all-reduces inside the sum
        kernels, the halo of t1 by put / wait-copy kernels -- no RCCL call in an iteration.  The set-up communicator
        (any `TorchComm`) only gathers the IPC blobs."""
        import ctypes as C
        if not getattr(self, "compact", False):
            raise RuntimeError("the mailbox transport serves the compact partitioned plan")
        if self.ops.A.plan.n_ghost and not self.ops.A.plan.direct:
            raise RuntimeError("the mailbox transport needs contiguous send runs (slab partitions have them)")
        self.close()
        eng = self.engine
        handle = C.c_void_p()
        eng._check(eng.lib.nss_dist_create(None, self.comm.size, self.comm.rank, C.byref(handle)))
        halo = self.ops.A.native_halo(self.t1, (0, 0))
        self.native = (handle, (None, halo, None))
        self.mailbox = MailboxTransport(self.comm, eng, [(halo, self.ops.n_u)])
        self.loop.state.p2p = self.mailbox.handle
        self.loop.keep.append(self.mailbox)

    def close(self):
        if getattr(self, "mailbox", None) is not None:
            self.loop.state.p2p = None
            self.mailbox.close()
            self.mailbox = None
        if getattr(self, "native", None) is not None:
            self.engine.lib.nss_dist_destroy(self.native[0])
            self.native = None

    def release(self):
        """Free the native handles in dependency order: the loop's dist handle, the V-cycle's native handle (it
        points into the dist handle created for it), then that dist handle."""
        self.close()
        aux = getattr(getattr(self, "ops", None), "aux", None)
        if aux is not None:
            aux.release()
        amg = getattr(self, "dist_amg", None)
        if amg is not None and getattr(amg, "_native", None) is not None:
            amg.engine.lib.nss_dist_amg_destroy(amg._native[0])
            amg._native = None
        if getattr(self, "_amg_dist_handle", None) is not None:
            self.engine.lib.nss_dist_destroy(self._amg_dist_handle)
            self._amg_dist_handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def start(self, tol, maxsteps, rel_err=True):
        self.first_direction()
        self.loop.start(self.wdn, self.err0, tol, rel_err, maxsteps)
        if self.ghost_mode:
            self._fill_ghosts()

    def iterate(self, it_begin, it_end):
        if self.native is not None:
            self.loop.enqueue_dist(self.native[0], self.native[1], self.overlap, it_begin, it_end)
            return
        loop, comm = self.loop, self.comm
        for it in range(it_begin, it_end):
            for kind, what in (self.SCHEDULE_COMPACT if getattr(self, "compact", False) else self.SCHEDULE):
                if kind == "cphases":
                    loop.cphases(what[0], what[1], it)
                    continue
                if kind == "halo" and what == "t4" and self.ghost_mode:
                    continue                             # t4's ghosts are derived from t1's in K2
                if kind == "halo" and what == "s1" and getattr(self, "ghost_p_mode", False):
                    continue                             # s1's ghosts follow their own recurrence
                if kind == "phases":
                    loop.phases(what[0], what[1], it)
                elif kind == "halo":
                    mat, hv = self.halo[what]
                    mat.exchange(hv)
                else:                                    # local sum in scal[8 + what] -> global in scal[what]
                    comm.allreduce_sum_into(loop.scal[8 + what:9 + what], loop.scal[what:what + 1])

    PHASE_NAMES = ("K1_BT_preA", "exchange_t1", "K2_A", "K3_B_sum", "allreduce_sKs", "K4_sum", "allreduce_wd", "K5")
    PHASE_NAMES_COMPACT = ("C1_BT_preA", "exchange_t1", "C23_A_B", "sum_sKs", "allreduce_sKs", "C4_sum", "allreduce_wd",
                           "unused")

    def phase_names(self):
        return self.PHASE_NAMES_COMPACT if getattr(self, "compact", False) else self.PHASE_NAMES

    def profile(self, it_begin, iterations):
        """Per-phase device times (ms, averaged) of `iterations` further iterations.  Native loop: HIP
        events recorded by the C loop itself (nss_dist_profile_*); Python-driven schedule: torch events
        around the same segments (includes the host's issue gaps)."""
        import ctypes as C
        eng = self.engine
        if self.native is not None:
            eng._check(eng.lib.nss_dist_profile_begin(self.native[0], int(iterations)))
            self.iterate(it_begin, it_begin + iterations)
            out = (C.c_double * 8)()
            n = C.c_int32()
            eng._check(eng.lib.nss_dist_profile_end(self.native[0], out, C.byref(n)))
            return dict(zip(self.phase_names(), [float(v) for v in out])), n.value
        torch = eng.torch
        loop, comm = self.loop, self.comm
        acc = [0.0] * 8
        marks = []
        for it in range(it_begin, it_begin + (iterations if getattr(self, "compact", False) else 0)):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
            ev[0].record()
            loop.cphases("C1", "C1", it)
            ev[1].record()
            self.halo["t1"][0].exchange(self.halo["t1"][1])
            ev[2].record()
            loop.cphases("C23", "C23", it)
            ev[3].record()
            loop.cphases("SUMA", "SUMA", it)
            ev[4].record()
            comm.allreduce_sum_into(loop.scal[9:10], loop.scal[1:2])
            ev[5].record()
            loop.cphases("C4", "SUMW", it)
            ev[6].record()
            comm.allreduce_sum_into(loop.scal[10:11], loop.scal[2:3])
            ev[7].record()
            ev[8].record()
            marks.append(ev)
        for it in range(it_begin, it_begin + (0 if getattr(self, "compact", False) else iterations)):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
            ev[0].record()
            if not getattr(self, "ghost_p_mode", False):
                self.halo["s1"][0].exchange(self.halo["s1"][1])
            loop.phases("K1", "K1", it)
            ev[1].record()
            self.halo["t1"][0].exchange(self.halo["t1"][1])
            ev[2].record()
            loop.phases("K2", "K2", it)
            ev[3].record()
            if not self.ghost_mode:
                self.halo["t4"][0].exchange(self.halo["t4"][1])
            loop.phases("K3", "SUM1", it)
            ev[4].record()
            comm.allreduce_sum_into(loop.scal[9:10], loop.scal[1:2])
            ev[5].record()
            loop.phases("ALPHA", "SUM2", it)
            ev[6].record()
            comm.allreduce_sum_into(loop.scal[10:11], loop.scal[2:3])
            ev[7].record()
            loop.phases("BETA", "K5", it)
            ev[8].record()
            marks.append(ev)
        torch.cuda.synchronize()
        for ev in marks:
            for k in range(8):
                acc[k] += ev[k].elapsed_time(ev[k + 1]) / len(marks)
        return dict(zip(self.phase_names(), acc)), len(marks)

    def poll(self):
        return self.loop.poll()

    def history(self, upto):
        return self.loop.history(upto)

    def halo_summary(self):
        return self.ops.halo_doubles()

    def solve(self, tol=1e-10, maxsteps=100000, poll_every=16):
        """Full solve; returns (it, converged).  Every rank takes the same decision because
        the all-reduced scalars are bit-identical on all ranks."""
        self.start(tol, maxsteps)
        it, done, it_final = 0, False, 0
        while it < maxsteps:
            end = min(maxsteps, it + poll_every)
            self.iterate(it, end)
            it = end
            done, it_final, _ = self.poll()
            if done:
                break
        return (it_final if done else maxsteps - 1), done


def b_in_layout_of_a(A, B, engine):
    """B's slab with its ghost columns numbered in the layout of A's operand (A's ghosts contain B's): the SpMVs
    with A and with B then multiply the SAME halo-extended buffer."""
    n_u, n_p = A.n_cols_owned, B.n_rows
    gb, ga = B.plan.ghosts, A.plan.ghosts
    pos = np.searchsorted(ga, gb)
    if gb.size and (pos.max(initial=0) >= ga.size or not np.array_equal(ga[np.minimum(pos, ga.size - 1)], gb)):
        raise RuntimeError("ghost columns of B are not among those of A's operand")
    loc = B.local_scipy
    cols = loc.indices.astype(np.int64)
    ghost = cols >= n_u
    cols[ghost] = n_u + pos[cols[ghost] - n_u]
    b_on_a = sp.csr_matrix((loc.data, cols.astype(np.int32), loc.indptr), shape=(n_p, n_u + ga.size))
    b_on_a.sort_indices()
    return SparseMatrix.from_scipy(b_on_a, engine=engine)


class Bpcg1DistLoop:
    """Row-partitioned device loop of the textbook Bramble-Pasciak CG (bramble_pasciak_cg.py:110-143) behind
    `bramble_pasciak_cg(...)` called with distributed operands: the fused kernels of `nss_bpcg1_*` on this rank's
    slab; per iteration the exchange of d (both components in one grouped phase), of t2_u and of a_u, and two
    all-reduces of one double -- issued natively from C over RCCL (`nss_bpcg1_iterate_dist`) or, with any other
    communicator, between the device phases (`nss_bpcg1_phases`).  The scalars are identical on every rank, so
    every rank takes the same stop decision."""

    NATIVE = True        # False: keep the host-driven schedule even over an RCCL communicator (tests)
    TRANSPORT = None     # "mailbox": the native loop over the mailbox transport (csrc/p2p.h) with any set-up communicator

    @classmethod
    def try_create(cls, a_matrix, b_matrix, c_matrix, pre_a, pre_s, k, vecs, native=None):
        from hipla import fused
        native = cls.NATIVE if native is None else native
        if c_matrix is not None or not (isinstance(a_matrix, DistSparseMatrix) and isinstance(b_matrix, DistSparseMatrix)):
            return None
        eng = a_matrix.engine
        if not fused.ENABLED or not hasattr(getattr(eng, "lib", None), "nss_bpcg1_phases"):
            return None
        bt = b_matrix.T
        if not isinstance(bt, DistSparseMatrix):
            return None
        pa_d, pa_b, ps = fused.native_diag(pre_a), fused.native_bjac(pre_a), fused.native_diag(pre_s)
        if ps is None or (pa_d is None and pa_b is None):
            return None
        return cls(eng, a_matrix, b_matrix, bt, pa_d, pa_b, ps, k, vecs, native)

    def __init__(self, eng, A, B, BT, pa_d, pa_b, ps, k, vecs, native):
        import ctypes as C
        from hipla.fused import Bpcg1Loop
        self.engine, self.comm, self.A, self.BT = eng, A.comm, A, BT
        self.B_onA = b_in_layout_of_a(A, B, eng)
        # the SpMV operands of the loop in halo-extended buffers (d_u, t2_u, a_u: layout of A's operand; d_p: B^T's)
        self.d0, self.t20, self.a0, self.d1 = A.operand(), A.operand(), A.operand(), BT.operand()
        self.d0.data = vecs["d"][0]
        self.d1.data = vecs["d"][1]
        self.a0.data = vecs["a"][0]
        local = dict(vecs)
        local["d"], local["a"], local["t2"] = [self.d0, self.d1], [self.a0, vecs["a"][1]], [self.t20, vecs["t2"][1]]
        self.loop = Bpcg1Loop(eng, A.local, self.B_onA, pa_d, pa_b, ps, k, local, BT=BT.local)
        self.loop.state.local_sums = 1
        self.loop.enqueue = self.enqueue
        self.native = None
        self.mailbox = None
        comm_handle = getattr(self.comm, "comm", None)        # RcclComm: an ncclComm_t
        if self.TRANSPORT == "mailbox" and hasattr(eng.lib, "nss_p2p_create"):
            handle = C.c_void_p()
            eng._check(eng.lib.nss_dist_create(None, self.comm.size, self.comm.rank, C.byref(handle)))
            halos = (A.native_halo(self.d0, (0, 0)), BT.native_halo(self.d1, (0, 0)))
            self.mailbox = MailboxTransport(self.comm, eng, [(halos[0], A.n_cols_owned), (halos[1], BT.n_cols_owned)])
            self.mailbox.attach(handle)
            self.native = (handle, halos)
        elif native and comm_handle is not None and hasattr(eng.lib, "nss_bpcg1_iterate_dist"):
            handle = C.c_void_p()
            eng._check(eng.lib.nss_dist_create(comm_handle, self.comm.size, self.comm.rank, C.byref(handle)))
            self.native = (handle, (A.native_halo(self.d0), BT.native_halo(self.d1)))

    def close(self):
        if getattr(self, "native", None) is not None:
            self.engine.lib.nss_dist_destroy(self.native[0])
            self.native = None
        if getattr(self, "mailbox", None) is not None:
            self.mailbox.close()
            self.mailbox = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def enqueue(self, it_begin, it_end):
        import ctypes as C
        eng, loop, st = self.engine, self.loop, self.loop.state
        if self.native is not None:
            handle, (hu, hp) = self.native
            eng._check(eng.lib.nss_bpcg1_iterate_dist(C.byref(st), handle, C.byref(hu), C.byref(hp), int(it_begin),
                                                      int(it_end), eng.stream))
            return
        A, BT, comm, scal = self.A, self.BT, self.comm, loop.scal

        def phases(first, last, it):
            eng._check(eng.lib.nss_bpcg1_phases(C.byref(st), first, last, it, eng.stream))

        for it in range(it_begin, it_end):
            A.exchange(self.d0)
            BT.exchange(self.d1)
            phases(1, 1, it)
            A.exchange(self.t20)
            phases(2, 2, it)
            comm.allreduce_sum_into(scal[8:9], scal[1:2])            # <d, t1>
            phases(3, 3, it)
            A.exchange(self.a0)
            phases(4, 4, it)
            comm.allreduce_sum_into(scal[9:10], scal[2:3])           # rho_new
            phases(5, 5, it)

    def run(self, rho, err0, tolerance, max_steps, poll_every=None):
        return self.loop.run(rho, err0, tolerance, max_steps, poll_every)


class DistributedMinres:
    """Row-partitioned preconditioned MINRES (minres.py:12-149) on this rank, K = [[A, B^T], [B, 0]],
    C = diag(preA, preM): set-up through the operator protocol with distributed operands, iteration
    through the fused device kernels (`nss_minres_*`).  Per iteration ONE grouped halo exchange -- z0 in
    the layout of A's operand, which also serves B (its local columns are renumbered into that layout), and
    z1 in the layout of B^T's operand -- and two all-reduces of one double (delta, gamma_new^2), issued
    natively from C over RCCL (`nss_minres_iterate_dist`) or, with any other communicator, between the
    device phases (`nss_minres_phases`).  The scalars are identical on every rank, so every rank takes the
    same stop decision."""

    def __init__(self, sysm, f, g, blocks, dist, engine=None, comm=None, native=True, sol=None, initialize=True,
                 transport=None):
        import ctypes as C
        from math import sqrt
        from hipla import BlockVector
        from hipla.fused import MinresLoop, native_bjac, native_diag
        self.engine = eng = engine if engine is not None else get_engine()
        self.comm = comm if comm is not None else TorchComm(dist, eng)
        ops = self.ops = DistributedStokes(sysm, blocks, self.comm, eng)
        n_u, n_p = ops.n_u, ops.n_p
        self.B_onA = b_in_layout_of_a(ops.A, ops.B, eng)

        us, ps = ops.local_slices()
        fv, gv = ops.vectors(f, g)
        rhs = BlockVector([fv, gv])

        def plain():
            return BlockVector([fv.CreateVector(), gv.CreateVector()])

        def extended():            # SpMV operands: owned views of halo-extended buffers
            z0, z1 = ops.A.operand(), ops.BT.operand()
            z0.comm = z1.comm = self.comm
            return BlockVector([z0, z1])

        self.u = u = sol if sol is not None else plain()
        v_ring, w_ring, z_ring, kz = [plain() for _ in range(3)], [plain() for _ in range(3)], [extended(), extended()], plain()
        K = BlockMatrix([[ops.A, ops.B.T], [ops.B, None]])
        Cm = BlockMatrix([[ops.preA, None], [None, ops.preM]])
        # minres.py:62-75 -- at iteration k = 1 the ring indices are v = v[1], v_old = v[0], z = z[1]
        v, z = v_ring[1], z_ring[1]
        if initialize:
            u[:] = 0.0
            v.data = rhs
        else:
            v.data = rhs - K * u
        z.data = Cm * v
        self.gamma = sqrt(InnerProduct(z, v))                # global: the slabs know their communicator
        z.data = 1.0 / self.gamma * z
        v.data = 1.0 / self.gamma * v
        for ring in (v_ring, w_ring):
            for j in (0, 2) if ring is v_ring else (0, 1, 2):
                ring[j][:] = 0.0
        pa_d, pa_b, pm = native_diag(ops.preA), native_bjac(ops.preA), native_diag(ops.preM)
        if pm is None or (pa_d is None and pa_b is None):
            raise RuntimeError("DistributedMinres: preA must be a (block) Jacobi, preM diagonal")
        self.loop = MinresLoop(eng, ops.A.local, self.B_onA, ops.BT.local, pa_d, pa_b, pm, u, v_ring, w_ring, z_ring, kz)
        self.loop.state.local_sums = 1
        self.z_ring = z_ring
        self.native = None
        self.mailbox = None
        comm_handle = getattr(self.comm, "comm", None)        # RcclComm: an ncclComm_t
        if transport == "mailbox":        # the native loop over the mailbox transport (csrc/p2p.h), any set-up communicator
            handle = C.c_void_p()
            eng._check(eng.lib.nss_dist_create(None, self.comm.size, self.comm.rank, C.byref(handle)))
            halos = (ops.A.native_halo(z_ring[0][0], (0, 0)), ops.BT.native_halo(z_ring[0][1], (0, 0)))
            self.mailbox = MailboxTransport(self.comm, eng, [(halos[0], ops.n_u), (halos[1], ops.n_p)])
            self.mailbox.attach(handle)
            self.native = (handle, halos)
        elif native and comm_handle is not None and hasattr(eng.lib, "nss_minres_iterate_dist"):
            handle = C.c_void_p()
            eng._check(eng.lib.nss_dist_create(comm_handle, self.comm.size, self.comm.rank, C.byref(handle)))
            self.native = (handle, (ops.A.native_halo(z_ring[0][0]), ops.BT.native_halo(z_ring[0][1])))

    def close(self):
        if getattr(self, "native", None) is not None:
            self.engine.lib.nss_dist_destroy(self.native[0])
            self.native = None
        if getattr(self, "mailbox", None) is not None:
            self.mailbox.close()
            self.mailbox = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _iterate(self, k_begin, k_end):
        import ctypes as C
        eng, loop, st = self.engine, self.loop, self.loop.state
        if self.native is not None:
            handle, (h0, h1) = self.native
            eng._check(eng.lib.nss_minres_iterate_dist(C.byref(st), handle, C.byref(h0), C.byref(h1), int(k_begin),
                                                       int(k_end), eng.stream))
            return
        ops, comm = self.ops, self.comm
        for k in range(k_begin, k_end):
            z = self.z_ring[k % 2]
            ops.A.exchange(z[0])
            ops.BT.exchange(z[1])
            base = ((k - 1) & 1) * 32
            eng._check(eng.lib.nss_minres_phases(C.byref(st), 1, 2, k, eng.stream))
            comm.allreduce_sum_into(loop.scal[base + 19:base + 20], loop.scal[base + 0:base + 1])     # delta
            eng._check(eng.lib.nss_minres_phases(C.byref(st), 3, 4, k, eng.stream))
            comm.allreduce_sum_into(loop.scal[base + 20:base + 21], loop.scal[base + 2:base + 3])     # gamma_new^2
            eng._check(eng.lib.nss_minres_phases(C.byref(st), 5, 5, k, eng.stream))

    def solve(self, tol=1e-7, maxsteps=100, poll_every=16):
        """Returns (u, errors, hit_relative_tol) as `MinRes` does (errors[0] == 1.0)."""
        import ctypes as C
        from hipla.fused import (M_C, M_C_OLD, M_ERR0, M_ETA_OLD, M_GAMMA, M_RES_OLD, M_TOL)
        eng, loop, st = self.engine, self.loop, self.loop.state
        loop.hist = eng.zeros(maxsteps + 2)
        st.hist = loop.hist.data_ptr()
        scal = np.zeros(64)
        g = self.gamma
        scal[M_GAMMA], scal[M_ETA_OLD], scal[M_C_OLD], scal[M_C] = g, g, 1.0, 1.0
        scal[M_RES_OLD], scal[M_ERR0], scal[M_TOL] = g, g, tol
        scal[16:19] = 1.0
        eng.upload(scal, loop.scal)
        loop.ctrl.zero_()
        stop, k_stop, reason, last = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        k = 1
        while k < maxsteps + 1:
            end = min(maxsteps + 1, k + poll_every)
            self._iterate(k, end)
            k = end
            eng._check(eng.lib.nss_minres_poll(C.byref(st), C.byref(stop), C.byref(k_stop), C.byref(reason),
                                               C.byref(last), eng.stream))
            if stop.value:
                break
        last_k = k_stop.value if stop.value else maxsteps
        errors = [1.0] + [float(x) for x in eng.to_host(loop.hist)[1: last_k + 1]]
        return self.u, errors, bool(stop.value and reason.value == 1)
