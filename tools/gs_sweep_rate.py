#!/usr/bin/env python3
"""Device time of the multicolour block Gauss-Seidel sweeps (templates/NavierStokesSIMPLE_iterative.py:376-381, the
reference's default GS=True) at a given grid: the symmetric operator y = 0; Smooth; SmoothBack and one sweep, for the
colour-major layout inside the sweep (default) and round 1's row-permuted form, greedy and Luby colours; against the
block-Jacobi apply and the STREAM triad of the same run.   python tools/gs_sweep_rate.py [grid] [--inflate B]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "navier-stokes-solver_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

import hipla
from staggered_grid import mac_stokes


def event_ms(fn, reps=20, between=None):
    marks = []
    for _ in range(reps + 3):
        if between is not None:
            between()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        marks.append((a, b))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in marks[3:]) / reps


def main():
    grid = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 136
    inflate = int(sys.argv[sys.argv.index("--inflate") + 1]) if "--inflate" in sys.argv else 1
    s = mac_stokes(3, grid, 0.01)
    if inflate > 1:
        s = s.inflate(inflate)
    blocks = s.line_blocks(3) if inflate == 1 else s.line_blocks(1)
    eng = hipla.get_engine()
    A = hipla.SparseMatrix.from_scipy(s.A)
    n = s.n_u
    a_info = A.handle.info()
    x = hipla.Vector.from_numpy(np.random.default_rng(0).standard_normal(n))
    y = hipla.Vector(n)
    # something that sweeps the caches between two timed applies (the loop never applies preA back to back)
    big = [eng.zeros(1 << 25) for _ in range(3)]
    flush = lambda: eng.stream_triad(0.5, big[0], big[1], big[2])
    triad_ms = event_ms(flush)
    triad_gbs = 24.0 * (1 << 25) / (triad_ms * 1e-3) / 1e9
    J = hipla.BlockJacobi(A, blocks)
    bj_ms = event_ms(lambda: J.Mult(x, y), between=flush)
    bs, nb = J.idx_host.shape if hasattr(J, "idx_host") else (blocks.shape[0], blocks.shape[1])
    print("3-D MAC Stokes n=%d%s: n_u %d, %.1f non-zeros per row, blocks %d x %d; triad %.0f GB/s"
          % (grid, " inflated x%d" % inflate if inflate > 1 else "", n, a_info["nnz"] / n, bs, nb, triad_gbs))
    print("  block-Jacobi apply                         %8.4f ms" % bj_ms)
    rows = []
    for layout, method in (("colour-major", "greedy"), ("colour-major", "luby"), ("rows", "greedy"), ("rows", "luby")):
        t0 = time.perf_counter()
        G = hipla.BlockGaussSeidel(A, blocks, layout=layout, coloring_method=method)
        torch.cuda.synchronize()
        setup = time.perf_counter() - t0
        p_info = G.perm_handle.info()
        sym_ms = event_ms(lambda: G.Mult(x, y), between=flush)
        one_ms = event_ms(lambda: G.Smooth(y, x), between=flush)
        # algorithmic bytes of ONE sweep: the permuted matrix as stored (values + column stream + row pointers), the
        # inverse blocks (bs doubles per row), x once, y read and written once
        mat = 8 * p_info["nnz"] + p_info["index_bytes"] * (p_info["nnz"] // p_info["index_group"]) + 4 * (p_info["rows"] + 1)
        sweep_bytes = mat + 8 * bs * n + 8 * n + 16 * n
        sym_bytes = 2 * sweep_bytes + (2 * 16 * n if layout == "colour-major" else 8 * n)     # + gather in / scatter out
        print("  %-12s %-6s colours %d (blocks per colour %s), set-up %.2f s, matrix form %s / %d B index"
              % (layout, method, G.ncolors, np.diff(G.color_ptr).tolist(), setup, p_info["operand_form"], p_info["index_bytes"]))
        print("      symmetric operator %8.4f ms  (%5.2f GB algorithmic -> %4.0f GB/s = %.2f of the triad);  one sweep call %8.4f ms"
              % (sym_ms, sym_bytes / 1e9, sym_bytes / (sym_ms * 1e-3) / 1e9, sym_bytes / (sym_ms * 1e-3) / 1e9 / triad_gbs, one_ms))
        rows.append((layout, method, sym_ms))
        del G
        torch.cuda.empty_cache()
    base = [r for r in rows if r[0] == "rows" and r[1] == "luby"][0][2]
    best = rows[0][2]
    print("  symmetric operator: round-1 form (rows, luby) %.4f ms -> colour-major + greedy %.4f ms (x %.2f)" % (base, best, base / best))


if __name__ == "__main__":
    main()
