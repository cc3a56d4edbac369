#!/usr/bin/env python3
"""Where does the SpMV rate go beyond 1e7 rows?  The plain y = A x launch (staged operand, 7 non-zeros per row) on
(a) the vector Laplacian of the MAC grid (operand runs one grid slab away: re-read 549 row blocks later at n = 232) and
(b) a banded matrix with the same 7 entries per row all within +-3 of the diagonal (every operand run is the row
block's own neighbourhood: nothing is ever re-read), at the row counts of cfg4 and cfg5; against the STREAM triad of
the same run.  If (b) falls off like (a), the loss beyond 1e7 rows has nothing to do with operand reuse.
    python tools/size_falloff_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "navier-stokes-solver_amd")):
    sys.path.insert(0, p)
import numpy as np
import scipy.sparse as sp
import torch

import hipla
from staggered_grid import mac_stokes


def event_ms(fn, flush, reps=12):
    ev = []
    for _ in range(reps + 2):
        flush()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev[2:]) / reps


def main():
    eng = hipla.get_engine()
    big = [eng.zeros(1 << 26) for _ in range(3)]
    flush = lambda: eng.stream_triad(0.5, big[0], big[1], big[2])
    tri = event_ms(flush, lambda: None)
    print("triad %.0f GB/s" % (24.0 * (1 << 26) / (tri * 1e-3) / 1e9))
    for grid in (136, 232):
        s = mac_stokes(3, grid, 0.01)
        n = s.n_u
        band = sp.diags([np.full(n - abs(k), 1.0 + 0.1 * k) for k in range(-3, 4)], list(range(-3, 4)), format="csr")
        for label, mat in (("grid Laplacian", s.A), ("banded, same 7 entries per row", band)):
            A = hipla.SparseMatrix.from_scipy(mat)
            info = A.handle.info()
            x, y = eng.zeros(n), eng.zeros(n)
            x.fill_(1.0)
            ms = event_ms(lambda: eng.csr_spmv(A.handle, 1.0, x, 0.0, y), flush)
            print("n_u %9d  %-32s %s  %.4f ms  %5.0f GB/s algorithmic (as stored)" % (n, label, info["operand_form"], ms,
                  info["algorithmic_bytes"] / (ms * 1e-3) / 1e9))
            del A, x, y
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
