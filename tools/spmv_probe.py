#!/usr/bin/env python3
"""Time the plain SpMV of one operator:  spmv_probe.py <dim> <grid> <inflate> [reps].  One line per run; used with
NSS_LIB_PATH to compare kernel variants (tools/build_variant.sh), e.g. the NSS_PROBE_GATHER builds that replace
the operand gather by a constant (1) or by an always-L1-resident load (2) to see what the gather costs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "navier-stokes-solver_amd")):
    sys.path.insert(0, p)
import torch

import hipla
from staggered_grid import mac_stokes

dim, grid, inflate = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
which = sys.argv[5] if len(sys.argv) > 5 else "A"            # A | B | BT
eng = hipla.get_engine()
s = mac_stokes(dim, grid, 0.01)
if inflate > 1:
    s = s.inflate(inflate)
mat = {"A": s.A, "B": s.B, "BT": s.B.T.tocsr()}[which]
A = hipla.SparseMatrix.from_scipy(mat)
info = A.handle.info()
x = torch.ones(mat.shape[1], dtype=torch.float64, device="cuda")
y = torch.zeros(mat.shape[0], dtype=torch.float64, device="cuda")
for _ in range(5):
    eng.csr_spmv(A.handle, 1.0, x, 0.0, y)
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        eng.csr_spmv(A.handle, 1.0, x, 0.0, y)
    b.record()
    b.synchronize()
    best = min(best, a.elapsed_time(b) / reps)
print("%s %s rows %d nnz/row %.1f group %d %s (%d row blocks): %.4f ms  %.0f GB/s" % (
    os.environ.get("NSS_LIB_PATH", "default").split("/")[-1], which, mat.shape[0], info["nnz"] / mat.shape[0],
    info["index_group"], info["operand_form"], info["row_blocks"], best,
    info["algorithmic_bytes"] / best / 1e6), flush=True)
