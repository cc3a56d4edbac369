#!/usr/bin/env python3
"""Why does the same SpMV on the same HDG-like matrix (82 non-zeros per row, 2.5 GB) run in one of two
"modes" from process to process (round 1: K2 0.50 or 0.59 ms)?  One process, one matrix; the operand /
result vectors are re-allocated at shifted addresses (padding allocations in between) and the plain SpMV is
timed for each placement.  Prints address bits next to the time.  Run on the GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "navier-stokes-solver_amd")):
    sys.path.insert(0, p)
import torch

import hipla
from staggered_grid import mac_stokes

grid, inflate = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (44, 12)
eng = hipla.get_engine()
s = mac_stokes(3, grid, 0.01).inflate(inflate)
A = hipla.SparseMatrix.from_scipy(s.A)
n = s.n_u
info = A.handle.info()


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


print("matrix: %d rows, %d nnz, index group %d, %.2f GB streamed per SpMV" % (n, info["nnz"], info["index_group"],
                                                                            info["algorithmic_bytes"] / 1e9))
keep = []
print("| pad before x (MiB) | pad between x and y (MiB) | x addr mod 2^21 | y addr mod 2^21 | (y - x) mod 2^21 | ms | GB/s |")
print("|---|---|---|---|---|---|---|")
for pad0 in (0, 1, 3, 8):
    for pad1 in (0, 1, 2, 5, 16):
        if pad0:
            keep.append(torch.empty(pad0 << 17, dtype=torch.float64, device="cuda"))      # pad0 MiB
        x = torch.ones(n, dtype=torch.float64, device="cuda")
        if pad1:
            keep.append(torch.empty(pad1 << 17, dtype=torch.float64, device="cuda"))
        y = torch.zeros(n, dtype=torch.float64, device="cuda")
        ms = timed(lambda: eng.csr_spmv(A.handle, 1.0, x, 0.0, y))
        mask = (1 << 21) - 1
        print("| %d | %d | %#x | %#x | %#x | %.4f | %.0f |" % (pad0, pad1, x.data_ptr() & mask, y.data_ptr() & mask,
                                                           (y.data_ptr() - x.data_ptr()) & mask, ms,
                                                           info["algorithmic_bytes"] / ms / 1e6), flush=True)
        keep += [x, y]
# the same vectors, timed again in reverse order: is the time a property of the placement or of the moment?
print("re-timed, last placement first:")
for i in range(len(keep) - 1, 0, -1):
    if keep[i].numel() == n and keep[i - 1].numel() == n and i % 2 == 1:
        pass
vecs = [t for t in keep if t.numel() == n]
for x, y in list(zip(vecs[0::2], vecs[1::2]))[::-4]:
    ms = timed(lambda: eng.csr_spmv(A.handle, 1.0, x, 0.0, y))
    print("x %#x y %#x : %.4f ms" % (x.data_ptr(), y.data_ptr(), ms), flush=True)
