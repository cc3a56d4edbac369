#!/usr/bin/env python3
"""Where the wall time of one scale-factor estimate goes at cfg2 (6.7e4 rows): the marks of hipla/eigen.py::TRACE
(workspace, pinned buffers, the step at which the host saw convergence, the final full spectrum, the drain) and a
cProfile of five calls.   python tools/lanczos_trace.py"""
import cProfile, pstats, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "navier-stokes-solver_amd")):
    sys.path.insert(0, p)
import torch, hipla
from hipla import eigen
from staggered_grid import mac_stokes
s = mac_stokes(2, 183, 0.01)
A = hipla.SparseMatrix.from_scipy(s.A)
pre = hipla.BlockJacobi(A, s.line_blocks(3))
for _ in range(3):
    eigen.EigenValues_Preconditioner(mat=A, pre=pre, tol=1e-3)
import time
for _ in range(3):
    eigen.TRACE = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eigen.EigenValues_Preconditioner(mat=A, pre=pre, tol=1e-3)
    print("  ".join("%s +%.2f ms" % (l, 1e3 * (t - t0)) for l, t in eigen.TRACE), " | return +%.2f ms" % (1e3 * (time.perf_counter() - t0)))
eigen.TRACE = None
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    eigen.EigenValues_Preconditioner(mat=A, pre=pre, tol=1e-3)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
