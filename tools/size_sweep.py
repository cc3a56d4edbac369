#!/usr/bin/env python3
"""Iteration rate and HBM rates of the fused BPCG loop over a sweep of problem sizes: one
`bench.py --grid n --cpu-iters 0 --steps 200 --warmup 20` child process per size (run on the GPU
box).  Prints the markdown table kept as profiles/r01_size_sweep.md."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
grids = [int(a) for a in sys.argv[1:]] or [24, 34, 48, 68, 86, 108, 136, 180, 232]
print("| grid n | DoF | iterations/s | ms/iteration | whole-iteration algorithmic GB/s | K2 in the loop GB/s (algorithmic) "
      "| same-run triad GB/s |")
print("|---|---|---|---|---|---|---|")
for n in grids:
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--grid", str(n), "--cpu-iters", "0", "--steps", "200",
                          "--warmup", "20"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    d = json.loads(out.decode().strip().splitlines()[-1])
    h = d["hbm_GBs"]
    print("| %d | %d | %.0f | %.4f | %.0f | %.0f | %.0f |" % (n, d["config"]["n_u"] + d["config"]["n_p"], d["value"], d["ms_per_step"],
                                                           h["whole_iteration_algorithmic"], h["spmv_AB_fused_C23"],
                                                           h["stream_triad"]), flush=True)
