"""Device time per BPCG iteration of the native partitioned loop on ONE GPU (1-rank RCCL
communicator), for a system of 1/8 of the headline size and an artificial middle-slab split:
what the schedule itself costs (split launches, second stream + events, all-reduce kernels)
before any real communication latency.   python tools/partition_overhead.py [grid]"""
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
    run.start(tol=0.0, maxsteps=its + warm)
    run.iterate(0, warm)
    torch.cuda.synchronize()
    t = time.perf_counter()
    run.iterate(warm, warm + its)
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t) / its


def main():
    grid = int(sys.argv[1]) if len(sys.argv) > 1 else 68
    s = mac_stokes(3, grid, 0.01)
    f, g = s.rhs(0)
    eng = hipla.get_engine()
    dist.init_process_group("gloo", init_method="file://" + os.path.join(tempfile.mkdtemp(), "rdv"), rank=0, world_size=1)
    comm = RcclComm(dist, eng)
    print("3-D MAC Stokes n=%d, %d DoF (one eighth of the 1e7-DoF system for n=68)" % (grid, s.ndof))
    rows = []
    for label, overlap, split in (("unsplit, all-reduces only", 0, False),
                                  ("split launches (3 per SpMV), one stream", 3, True),
                                  ("split + second stream + events", 2, True)):
        run = DistributedBpcg2(s, f, g, s.line_blocks(3), dist, eng, comm=comm)
        if split:
            nbs = {k: m.local.handle.info()["row_blocks"] for k, m in (("s1", run.ops.BT), ("t1", run.ops.A), ("t4", run.ops.B))}
            run.enable_native(comm.comm, {k: (max(1, nb // 16), nb - max(1, nb // 16)) for k, nb in nbs.items()})
        run.overlap = overlap
        rows.append((label, rate(run)))
        del run
    # the plain single-GPU fused loop for reference
    from solvers.bramblepasciak_new import BpcgSession
    import contextlib, io

    class Form:
        def __init__(self, mat):
            self.mat, self.condense = mat, False

    A, B = hipla.SparseMatrix.from_scipy(s.A), hipla.SparseMatrix.from_scipy(s.B)
    sol = hipla.BlockVector([hipla.Vector(s.n_u), hipla.Vector(s.n_p)])
    with contextlib.redirect_stdout(io.StringIO()):
        ses = BpcgSession(Form(A), Form(B), None, hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g),
                          hipla.BlockJacobi(A, s.line_blocks(3)), hipla.DiagonalMatrix(1.0 / s.mass), sol=sol)
    ses.first_direction()
    ses.fused.start(ses.wdn, ses.err0, 0.0, True, 440)
    ses.fused.enqueue(0, 40)
    torch.cuda.synchronize()
    t = time.perf_counter()
    ses.fused.enqueue(40, 440)
    torch.cuda.synchronize()
    rows.insert(0, ("single-GPU fused loop (8 launches)", 1e6 * (time.perf_counter() - t) / 400))
    for label, us in rows:
        print("  %-48s %7.1f us / iteration" % (label, us))
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
