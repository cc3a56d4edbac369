"""What the row-partitioned BPCG loop costs on the device before any link latency, on ONE GPU:

(a) a whole system of 1/8 of the headline size (n = 68) as a 1-rank partition over a real 1-rank RCCL
    communicator (the all-reduces are real RCCL kernels, the exchange has no peer): single-GPU compact loop vs the
    partitioned compact plan vs the partitioned eight-phase plan, with the per-phase device times of the native loop;
(b) the MIDDLE SLAB of the headline system (rank 3 of 8 of n = 136) with its real ghost rows and columns -- built
    with a loop-back communicator, the ghost values are never refreshed, so the numbers it iterates on are
    meaningless but every launch has its true shape: the per-rank device time of an 8-GPU run minus the links.

    python tools/partition_overhead.py [grid_a] [grid_b]"""
import contextlib
import io
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "navier-stokes-solver_amd"))

import torch
import torch.distributed as dist

import hipla
from distributed import DistributedBpcg2
from rccl_comm import RcclComm
from staggered_grid import mac_stokes


def rate(run, its=400, warm=40):
    run.start(tol=0.0, maxsteps=its + warm + 64)
    run.iterate(0, warm)
    torch.cuda.synchronize()
    t = time.perf_counter()
    run.iterate(warm, warm + its)
    torch.cuda.synchronize()
    us = 1e6 * (time.perf_counter() - t) / its
    phases, n = run.profile(warm + its, 48)
    return us, phases


def single_gpu(s, f, g, its=400, warm=40):
    from solvers.bramblepasciak_new import BpcgSession

    class Form:
        def __init__(self, mat):
            self.mat, self.condense = mat, False

    A, B = hipla.SparseMatrix.from_scipy(s.A), hipla.SparseMatrix.from_scipy(s.B)
    sol = hipla.BlockVector([hipla.Vector(s.n_u), hipla.Vector(s.n_p)])
    with contextlib.redirect_stdout(io.StringIO()):
        ses = BpcgSession(Form(A), Form(B), None, hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g),
                          hipla.BlockJacobi(A, s.line_blocks(3)), hipla.DiagonalMatrix(1.0 / s.mass), sol=sol)
    ses.first_direction()
    ses.fused.start(ses.wdn, ses.err0, 0.0, True, its + warm)
    ses.fused.enqueue(0, warm)
    torch.cuda.synchronize()
    t = time.perf_counter()
    ses.fused.enqueue(warm, warm + its)
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t) / its, ses.fused.folds_sums()


class LoopbackComm:
    """Rank `rank` of `size` with nobody else there: scalars and sums pass through, exchanges move nothing.  Carries
    the handle of a real 1-rank RCCL communicator for the native loop's all-reduces."""

    def __init__(self, rccl, rank, size):
        self.rccl, self.rank, self.size, self.comm, self.engine = rccl, rank, size, rccl.comm, rccl.engine
        self.stage = False

    def allreduce_scalar(self, value):
        return float(value)

    def allreduce_sum(self, buf):
        pass

    def allreduce_sum_into(self, src, dst):
        self.engine.copy(src, dst)

    def gather_requests(self, mine, compute_for_rank):
        return [compute_for_rank(q) for q in range(self.size)]      # (extra ghosts of the other ranks: not needed here)

    def exchange(self, plan, sendbuf, ext):
        pass

    def exchange_direct(self, plan, ext):
        pass


class SlabRun(DistributedBpcg2):
    def enable_native(self, comm_handle, interior=None):
        """the native handle as a 1-rank job: exchange() returns before any send, the all-reduces run"""
        import ctypes as C
        self.close()
        handle = C.c_void_p()
        eng = self.engine
        eng._check(eng.lib.nss_dist_create(comm_handle, 1, 0, C.byref(handle)))
        ops = self.ops
        halos = (ops.BT.native_halo(self.s1), ops.A.native_halo(self.t1), ops.B.native_halo(self.t4))
        self.native = (handle, halos)


def show(label, us, phases=None):
    print("  %-58s %7.1f us / iteration" % (label, us))
    if phases:
        print("      " + "  ".join("%s %.1f" % (k, 1e3 * v) for k, v in phases.items() if v > 0.0) + "   (us, device, per phase)")


def main():
    grid_a = int(sys.argv[1]) if len(sys.argv) > 1 else 68
    grid_b = int(sys.argv[2]) if len(sys.argv) > 2 else 136
    eng = hipla.get_engine()
    dist.init_process_group("gloo", init_method="file://" + os.path.join(tempfile.mkdtemp(), "rdv"), rank=0, world_size=1)
    comm = RcclComm(dist, eng)
    s = mac_stokes(3, grid_a, 0.01)
    f, g = s.rhs(0)
    print("(a) 3-D MAC Stokes n=%d, %d DoF as a 1-rank partition (1-rank RCCL communicator)" % (grid_a, s.ndof))
    us, folds = single_gpu(s, f, g)
    show("single-GPU compact loop (sums %s)" % ("folded: 4 launches" if folds else "stand-alone: 6 launches"), us)
    for plan in ("compact", "classic"):
        run = DistributedBpcg2(s, f, g, s.line_blocks(3), dist, eng, comm=comm, plan=plan)
        assert run.native is not None and run.compact == (plan == "compact")
        us, phases = rate(run)
        show("partitioned, native, %s plan (%d launches + 3 collectives)" % (plan, 6 if plan == "compact" else 9), us, phases)
        run.release()
        del run
    # the mailbox transport (csrc/p2p.h) on one rank: the all-reduce is part of the sum kernel (a store to and a load
    # from the own mailbox), no exchange -- the launch structure of an N-rank run without any link
    run = DistributedBpcg2(s, f, g, s.line_blocks(3), dist, eng, comm=comm, transport="mailbox")
    assert run.mailbox is not None
    us, phases = rate(run)
    show("partitioned, native, compact plan over the mailbox transport (6 launches, no collective library)", us, phases)
    run.release()
    del run
    if grid_b > 0:
        s = mac_stokes(3, grid_b, 0.01)
        f, g = s.rhs(0)
        lb = LoopbackComm(comm, 3, 8)
        print("(b) middle slab (rank 3 of 8) of n=%d, %d DoF: real ghost rows / columns, loop-back exchange" % (grid_b, s.ndof))
        for plan in ("compact", "classic"):
            run = SlabRun(s, f, g, s.line_blocks(3), dist, eng, comm=lb, plan=plan)
            assert run.native is not None and run.compact == (plan == "compact")
            print("    slab: n_u %d, n_p %d, ghosts A %d / B^T %d" % (run.ops.n_u, run.ops.n_p, run.ops.A.plan.n_ghost,
                                                                    run.ops.BT.plan.n_ghost))
            us, phases = rate(run)
            show("slab, native, %s plan" % plan, us, phases)
            run.release()
            del run
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
