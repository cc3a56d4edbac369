#!/usr/bin/env python3
"""Textbook BPCG (bramble_pasciak_cg.py) in its small-system form (6 dependent launches: nss_bpcg1_fold_mode 1) against
the 10-launch form (mode 0) over system sizes: microseconds per iteration through the drop-in entry point, interleaved
in one process.   python tools/bpcg1_sizes.py"""
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
from bramble_pasciak_cg import bramble_pasciak_cg
from staggered_grid import mac_stokes

eng = hipla.get_engine()
for dim, n in [(2, 60), (2, 120), (2, 183), (2, 300), (2, 577), (3, 40), (3, 68), (3, 100), (3, 136)]:
    s = mac_stokes(dim, n, 0.01)
    f, g = s.rhs(0)
    A, B = hipla.SparseMatrix.from_scipy(s.A), hipla.SparseMatrix.from_scipy(s.B)
    preA, preS = hipla.BlockJacobi(A, s.line_blocks(3)), hipla.DiagonalMatrix(1.0 / s.mass)
    fv, gv = hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g)
    k1, k2 = (300, 2300) if s.ndof < 2e6 else (50, 350)

    def run(k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            bramble_pasciak_cg(A, B, None, preA, preS, fv, gv, tolerance=0.0, max_steps=k, print_rates=False)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    best = {}
    for mode in (0, 1, -1):
        eng.lib.nss_bpcg1_fold_mode(mode)
        run(20)
        best[mode] = min((run(k2) - run(k1)) / (k2 - k1) for _ in range(3)) * 1e6
    eng.lib.nss_bpcg1_fold_mode(-1)
    print("%d-D n=%-4d %9d DoF: mode 0 %7.1f us / iteration | mode 1 %7.1f us | %+5.1f %%   (default rule: %.1f us)"
          % (dim, n, s.ndof, best[0], best[1], 100.0 * (best[1] / best[0] - 1.0), best[-1]))
    del preA, preS, A, B
    torch.cuda.empty_cache()
