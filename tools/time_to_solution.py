#!/usr/bin/env python3
"""Time to solution of the fused Bramble-Pasciak CG (v2) with the available velocity
preconditioners on one MI355X: iterations to a relative tolerance, loop seconds (the reference's
`timer_its`), set-up seconds (hierarchy / colouring on the host).  Markdown table on stdout."""
import contextlib
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "navier-stokes-solver_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

import hipla
from solvers.bramblepasciak_new import BramblePasciakCG
from staggered_grid import mac_stokes


class Form:
    def __init__(self, mat):
        self.mat, self.condense = mat, False


def main():
    grid = int(sys.argv[1]) if len(sys.argv) > 1 else 136
    tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-8
    s = mac_stokes(3, grid, 0.01)
    f, g = s.rhs(0)
    b = np.concatenate([f, g])
    A, B = hipla.SparseMatrix.from_scipy(s.A), hipla.SparseMatrix.from_scipy(s.B)
    preS = hipla.DiagonalMatrix(1.0 / s.mass)
    K = s.saddle_matrix()
    print("3-D MAC Stokes n=%d, %d DoF, BPCG v2 (fused loop), tol %g, one MI355X\n" % (grid, s.ndof, tol))
    print("| preA | set-up s | condition(preA A) | iterations | loop s | ms / iteration | true residual |")
    print("|---|---|---|---|---|---|---|")
    from templates.NavierStokesSIMPLE_iterative import MypreA, auxiliary_space_preconditioner

    def mypre(gs):
        _, _, aux = auxiliary_space_preconditioner(s)
        return MypreA(None, Form(A), s.line_blocks(3), GS=gs, aux=aux)

    makers = [("block Jacobi bs=3", lambda: hipla.BlockJacobi(A, s.line_blocks(3))),
              ("MypreA(GS=True): Gauss-Seidel sweeps + auxiliary-space term (the reference's default)", lambda: mypre(True)),
              ("MypreA(GS=False): block Jacobi + auxiliary-space term", lambda: mypre(False)),
              ("symmetric block Gauss-Seidel bs=3 (GS=True)", lambda: hipla.BlockGaussSeidel(A, s.line_blocks(3))),
              ("AMG V(1,1)", lambda: hipla.SmoothedAggregationAMG(A)),
              ("AMG V(1,1) + block Jacobi (additive MypreA)", lambda: hipla.SmoothedAggregationAMG(A) + hipla.BlockJacobi(A, s.line_blocks(3)))]
    only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
    for label, make in makers:
        if only and not any(label.startswith(o) for o in only):
            continue
        t0 = time.perf_counter()
        pre = make()
        torch.cuda.synchronize()
        t_setup = time.perf_counter() - t0
        sol = hipla.BlockVector([hipla.Vector(s.n_u), hipla.Vector(s.n_p)])
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            it, seconds = BramblePasciakCG(Form(A), Form(B), None, hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g),
                                           pre, preS, sol, tol=tol, maxsteps=200000, printrates=False)
        cond = float(out.getvalue().split("condition")[1].split()[0])
        res = np.linalg.norm(b - K @ sol.numpy()) / np.linalg.norm(b)
        print("| %s | %.1f | %.1f | %d | %.3f | %.3f | %.1e |" % (label, t_setup, cond, it + 1, seconds, 1e3 * seconds / (it + 1), res))
        del pre
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
