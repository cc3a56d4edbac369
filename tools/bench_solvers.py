#!/usr/bin/env python3
"""Iteration rates of the three fused Krylov loops (BPCG v2, BPCG v1, MINRES) through their
drop-in entry points, at the BASELINE.json sizes.  Each solver runs K1 and K2 iterations with the
stop test disabled (tolerance 0); the difference of the two wall times cancels the set-up
(Lanczos, initial residuals).  Prints a markdown table (kept under profiles/)."""
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
from minres import MinRes
from solvers.bramblepasciak_new import BramblePasciakCG
from staggered_grid import mac_stokes


class Form:
    def __init__(self, mat):
        self.mat, self.condense = mat, False


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def main():
    cases = [("cfg2", 2, 183), ("cfg3", 2, 577), ("cfg4", 3, 136)]
    if "--with-cfg5" in sys.argv:
        cases.append(("cfg5", 3, 232))
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
    if only:
        cases = [c for c in cases if c[0] in only]
    print("| config | DoF | solver | iterations/s | ms/iteration | SpMV/iteration |")
    print("|---|---|---|---|---|---|")
    for name, dim, n in cases:
        # the set-up (Lanczos, uploads) jitters by milliseconds: the small systems need thousands of
        # iterations for the difference of two runs to resolve 30 us per iteration
        k1, k2 = (500, 4500) if name in ("cfg2", "cfg3") else (50, 350)
        s = mac_stokes(dim, n, 0.01)
        f, g = s.rhs(0)
        A, B = hipla.SparseMatrix.from_scipy(s.A), hipla.SparseMatrix.from_scipy(s.B)
        preA = hipla.BlockJacobi(A, s.line_blocks(3))
        preS = hipla.DiagonalMatrix(1.0 / s.mass)
        fv, gv = hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g)
        K = hipla.BlockMatrix([[A, B.T], [B, None]])
        Cm = hipla.BlockMatrix([[preA, None], [None, preS]])

        def v2(k):
            sol = hipla.BlockVector([hipla.Vector(s.n_u), hipla.Vector(s.n_p)])
            BramblePasciakCG(Form(A), Form(B), None, fv, gv, preA, preS, sol, tol=0.0, maxsteps=k, printrates=False)

        def v1(k):
            bramble_pasciak_cg(A, B, None, preA, preS, fv, gv, tolerance=0.0, max_steps=k, print_rates=False)

        def mr(k):
            MinRes(mat=K, pre=Cm, rhs=hipla.BlockVector([fv, gv]), maxsteps=k, tol=0.0, printrates=False)

        for label, fn, spmv in (("BPCG v2 (bramblepasciak_new)", v2, 3), ("BPCG v1 (bramble_pasciak_cg)", v1, 6),
                                ("MINRES", mr, 3)):
            timed(lambda: fn(10))                             # warm-up (transposes, workspaces)
            t1 = min(timed(lambda: fn(k1)) for _ in range(2))
            t2 = min(timed(lambda: fn(k2)) for _ in range(2))
            per = (t2 - t1) / (k2 - k1)
            print("| %s | %d | %s | %.0f | %.4f | %d |" % (name, s.ndof, label, 1.0 / per, 1e3 * per, spmv))
        del A, B, preA, preS, K, Cm
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
