#!/usr/bin/env python3
"""What the bench's `setup_s.upload_lanczos_initial_residual` consists of at cfg4 in a FRESH process (first launches
of every kernel included): cProfile of BpcgSession(...) + first_direction(), twice (cold, then warm).
python tools/setup_trace.py [grid]"""
import contextlib, cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "navier-stokes-solver_amd")):
    sys.path.insert(0, p)
import torch
import hipla
from hipla import eigen
from staggered_grid import mac_stokes
from solvers.bramblepasciak_new import BpcgSession


class Form:
    def __init__(self, mat):
        self.mat, self.condense = mat, False


n = int(sys.argv[1]) if len(sys.argv) > 1 else 136
s = mac_stokes(3, n, 0.01)
f, g = s.rhs(0)
A, B = hipla.SparseMatrix.from_scipy(s.A), hipla.SparseMatrix.from_scipy(s.B)
for label in ("cold", "warm", "warm"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    B._transpose = None
    B.CreateTranspose()
    torch.cuda.synchronize()
    print("B.CreateTranspose() %s: %.1f ms" % (label, 1e3 * (time.perf_counter() - t0)))
B._transpose = None
preA, preM = hipla.BlockJacobi(A, s.line_blocks(3)), hipla.DiagonalMatrix(1.0 / s.mass)
torch.cuda.synchronize()
for label in ("cold", "warm"):
    sol = hipla.BlockVector([hipla.Vector(s.n_u), hipla.Vector(s.n_p)])
    eigen.TRACE = []
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    with contextlib.redirect_stdout(io.StringIO()):
        ses = BpcgSession(Form(A), Form(B), None, hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g), preA, preM,
                          sol=sol, initialize=True)
    ses.first_direction()
    torch.cuda.synchronize()
    pr.disable()
    print("%s: %.1f ms;  Lanczos marks: %s" % (label, 1e3 * (time.perf_counter() - t0),
                                              "  ".join("%s +%.1f" % (l, 1e3 * (t - t0)) for l, t in eigen.TRACE)))
    pstats.Stats(pr).sort_stats("cumtime").print_stats(18)
