"""RCCL communicator driven through ctypes (no torch.distributed on the data path).

`TorchComm` costs 30-60 us of host time per collective (c10d dispatch, work objects); with
two to three halo exchanges and two all-reduces per Krylov iteration that makes an 8-GPU run
host-bound.  Here the same operations are issued straight into librccl on the compute stream
(a handful of ctypes calls each): ``ncclAllReduce`` for the inner products and one
``ncclGroupStart / ncclSend.. / ncclRecv.. / ncclGroupEnd`` per halo exchange -- point-to-point
over xGMI, only between slab neighbours.  The communicator is bootstrapped from the existing
torch.distributed group (the ncclUniqueId travels in a broadcast); everything at set-up time
(index-list gathers, scalar reductions of the Lanczos phase) still uses torch.distributed.
"""

import ctypes as C

from distributed import TorchComm

NCCL_FLOAT64 = 8
NCCL_SUM = 0


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_byte * 128)]


def _load_rccl():
    import torch  # noqa: F401  (torch has already loaded its librccl; bind to the same SONAME)
    for name in ("librccl.so.1", "librccl.so"):
        try:
            return C.CDLL(name)
        except OSError:
            continue
    raise OSError("librccl not found")


class RcclComm(TorchComm):
    """Same interface as TorchComm; `allreduce_sum` and `exchange` go to RCCL directly."""

    def __init__(self, dist, engine=None):
        super().__init__(dist, engine)
        if not hasattr(self.engine, "stream"):
            raise RuntimeError("RcclComm needs the HIP engine")
        lib = self.lib = _load_rccl()
        lib.ncclGetErrorString.restype = C.c_char_p
        lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        uid = _UniqueId()
        if self.rank == 0:
            self._check(lib.ncclGetUniqueId(C.byref(uid)))
        box = [bytes(uid.internal) if self.rank == 0 else None]
        if self.size > 1:
            dist.broadcast_object_list(box, src=0)
        C.memmove(C.byref(uid), box[0], 128)
        self.comm = C.c_void_p()
        self._check(lib.ncclCommInitRank(C.byref(self.comm), self.size, uid, self.rank))
        self._plans = {}

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("RCCL error %d: %s" % (rc, self.lib.ncclGetErrorString(rc).decode()))

    def _stream(self):
        return C.c_void_p(self.engine.stream)

    def allreduce_sum(self, buf):
        if self.size == 1:
            return
        p = buf.data_ptr()
        self._check(self.lib.ncclAllReduce(p, p, buf.shape[0], NCCL_FLOAT64, NCCL_SUM, self.comm, self._stream()))

    def allreduce_sum_into(self, src, dst):
        if self.size == 1:
            self.engine.copy(src, dst)
            return
        self._check(self.lib.ncclAllReduce(src.data_ptr(), dst.data_ptr(), src.shape[0], NCCL_FLOAT64, NCCL_SUM,
                                           self.comm, self._stream()))

    def exchange(self, plan, sendbuf, ext):
        if self.size == 1:
            return
        key = (id(plan), sendbuf.data_ptr(), ext.data_ptr())
        ops = self._plans.get(key)
        if ops is None:
            sends, recvs = [], []
            off = 0
            for q, c in enumerate(plan.send_counts):
                if c:
                    sends.append((sendbuf.data_ptr() + 8 * off, int(c), q))
                off += int(c)
            off = plan.n_owned
            for q, c in enumerate(plan.recv_counts):
                if c:
                    recvs.append((ext.data_ptr() + 8 * off, int(c), q))
                off += int(c)
            ops = self._plans[key] = (sends, recvs)
        sends, recvs = ops
        if not sends and not recvs:
            return
        lib, comm, st = self.lib, self.comm, self._stream()
        self._check(lib.ncclGroupStart())
        for ptr, cnt, peer in sends:
            self._check(lib.ncclSend(ptr, cnt, NCCL_FLOAT64, peer, comm, st))
        for ptr, cnt, peer in recvs:
            self._check(lib.ncclRecv(ptr, cnt, NCCL_FLOAT64, peer, comm, st))
        self._check(lib.ncclGroupEnd())

    direct_sends = True

    def exchange_direct(self, plan, ext):
        """Halo exchange of a plan whose destinations are all served by one contiguous run of the
        owned entries (`plan.direct`): sends straight out of the operand, no pack kernel."""
        if self.size == 1:
            return
        key = ("direct", id(plan), ext.data_ptr())
        ops = self._plans.get(key)
        if ops is None:
            sends = [(ext.data_ptr() + 8 * plan.send_runs[q], int(c), q) for q, c in enumerate(plan.send_counts) if c]
            recvs, off = [], plan.n_owned
            for q, c in enumerate(plan.recv_counts):
                if c:
                    recvs.append((ext.data_ptr() + 8 * off, int(c), q))
                off += int(c)
            ops = self._plans[key] = (sends, recvs)
        sends, recvs = ops
        if not sends and not recvs:
            return
        lib, comm, st = self.lib, self.comm, self._stream()
        self._check(lib.ncclGroupStart())
        for ptr, cnt, peer in sends:
            self._check(lib.ncclSend(ptr, cnt, NCCL_FLOAT64, peer, comm, st))
        for ptr, cnt, peer in recvs:
            self._check(lib.ncclRecv(ptr, cnt, NCCL_FLOAT64, peer, comm, st))
        self._check(lib.ncclGroupEnd())

    def self_test(self, torch):
        """Ring shift + all-reduce against torch.distributed; raises on any mismatch."""
        dev = self.engine.device
        x = torch.full((4,), float(self.rank + 1), dtype=torch.float64, device=dev)
        ref = x.clone()
        self.allreduce_sum(x)
        if self.size > 1:
            self.dist.all_reduce(ref)
        torch.cuda.synchronize()
        if not torch.equal(x, ref):
            raise RuntimeError("RCCL all-reduce self-test mismatch")
        if self.size > 1:
            # halo-style exchange: every rank sends 3 doubles to its right neighbour
            import numpy as np

            class _Plan:
                pass

            plan = _Plan()
            plan.n_owned = 2
            right, left = (self.rank + 1) % self.size, (self.rank - 1) % self.size
            plan.send_counts = np.zeros(self.size, dtype=np.int64)
            plan.recv_counts = np.zeros(self.size, dtype=np.int64)
            plan.send_counts[right] = 3
            plan.recv_counts[left] = 3
            send = torch.full((3,), 100.0 + self.rank, dtype=torch.float64, device=dev)
            ext = torch.zeros(5, dtype=torch.float64, device=dev)
            self.exchange(plan, send, ext)
            torch.cuda.synchronize()
            if ext.tolist() != [0.0, 0.0] + [100.0 + left] * 3:
                raise RuntimeError("RCCL send/recv self-test mismatch: %r" % (ext.tolist(),))
            self._plans.clear()

    def close(self):
        if getattr(self, "comm", None):
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None
