#!/usr/bin/env python3
"""Block Jacobi applied in C1's epilogue (nss_bpcg2_fuse_block_jacobi) against the stand-alone apply, over system
sizes: microseconds per fused BPCG v2 iteration, interleaved in one process, best of 3 windows.
python tools/fuse_bjac_sizes.py"""
import contextlib
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "navier-stokes-solver_amd")):
    sys.path.insert(0, p)
import torch

import hipla
from solvers.bramblepasciak_new import BpcgSession
from staggered_grid import mac_stokes


class Form:
    def __init__(self, mat):
        self.mat, self.condense = mat, False


eng = hipla.get_engine()
CASES = [(2, 60), (2, 120), (2, 183), (2, 300), (2, 577), (3, 40), (3, 68), (3, 100), (3, 136)]
for dim, n in CASES:
    s = mac_stokes(dim, n, 0.01)
    f, g = s.rhs(0)
    A, B = hipla.SparseMatrix.from_scipy(s.A), hipla.SparseMatrix.from_scipy(s.B)
    sol = hipla.BlockVector([hipla.Vector(s.n_u), hipla.Vector(s.n_p)])
    eng.lib.nss_bpcg2_fuse_block_jacobi(1)                 # (B^T is planned around the blocks at every size)
    with contextlib.redirect_stdout(io.StringIO()):
        ses = BpcgSession(Form(A), Form(B), None, hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g),
                          hipla.BlockJacobi(A, s.line_blocks(3)), hipla.DiagonalMatrix(1.0 / s.mass), sol=sol)
    loop = ses.fused
    assert loop.c1_applies_bjac
    ses.first_direction()
    its = 2000 if s.ndof < 2e6 else 300
    loop.start(ses.wdn, ses.err0, 0.0, True, 8 * its + 100)
    loop.enqueue(0, 50)
    best = {0: 1e9, 1: 1e9}
    it = 50
    for rep in range(3):
        for on in (0, 1):
            eng.lib.nss_bpcg2_fuse_block_jacobi(on)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loop.enqueue(it, it + its)
            torch.cuda.synchronize()
            best[on] = min(best[on], 1e6 * (time.perf_counter() - t0) / its)
            it += its
    eng.lib.nss_bpcg2_fuse_block_jacobi(-1)
    print("%d-D n=%-4d %9d DoF  folds sums %-5s: stand-alone apply %7.1f us / iteration | in C1 %7.1f us | %+5.1f %%"
          % (dim, n, s.ndof, bool(loop.folds_sums()), best[0], best[1], 100.0 * (best[1] / best[0] - 1.0))
          + ("   [default: in C1]" if loop.c1_applies_preA() else ""))
    del ses, loop, A, B
    torch.cuda.empty_cache()
