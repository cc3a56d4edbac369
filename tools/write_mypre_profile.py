#!/usr/bin/env python3
"""profiles/r03_rocprof_mypre_a.md from the three passes of `tools/profile_bench.sh r03_mypre ...` under gpurun_out/prof:
the bench object of the kernel-trace pass, the per-level durations of the joint cycle, the generated summary tables.
python tools/write_mypre_profile.py"""
import collections
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "prof")
summ = open(os.path.join(P, "r03_mypre_summary.md")).read()
d = json.load(open(os.path.join(P, "r03_mypre_trace.json")))["roofline_mypre_a_gs"]
obj = {k: d[k] for k in ("iters_per_s", "ms_per_iteration", "C1_with_whole_preA_ms_in_loop", "sweep_call", "auxiliary_space_term")}
rows = list(csv.DictReader(open(os.path.join(P, "r03_mypre_trace", "r03_mypre_kernel_trace.csv"))))
lev = collections.defaultdict(list)
other = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "csr_multi" in n or "amg_diag_multi" in n:
        key = ("diag" if "diag" in n else "residual" if "MResidual" in n else "smoothing" if "MJacobi" in n else "transfer / coarse solve")
        lev[(key, int(r["Grid_Size_X"]) // 256)].append(us)
    for tag in ("EpiGsFused", "gs_enter_kernel", "gs_leave_kernel"):
        if tag in n:
            other[tag].append(us)
    if "EpiAxpby" in n and "csr_stream_kernel" in n and int(r["Grid_Size_X"]) // 256 > 4000:
        other["T / T^T"].append(us)
lines = ["| kernel of the joint cycle | workgroups | launches | avg (us) |", "|---|---|---|---|"]
for (k, g), v in sorted(lev.items(), key=lambda kv: (-kv[0][1], kv[0][0])):
    lines.append("| %s | %d | %d | %.1f |" % (k, g, len(v), sum(v) / len(v)))
avg = lambda k: sum(other[k]) / max(1, len(other[k]))
aux = obj["auxiliary_space_term"]
body = summ.split("\n", 1)[1]
txt = """# rocprofv3: the reference's default preconditioner MypreA(GS=True) + auxiliary-space term at cfg4 (round 3)

Command (tools/profile_bench.sh r03_mypre, three passes: --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE):
`python3 bench.py --mypre-a 1 --hdg 0 --secondary 0 --cpu-iters 0 --steps 40 --warmup 10 --windows 1`
(n = 136, 1.0e7 DoF; the headline block-Jacobi loop runs first, then `roofline_mypre_a_gs`: BPCG v2 with
preA = forward multicolour block Gauss-Seidel sweep, residual, auxiliary-space correction, backward sweep --
templates/NavierStokesSIMPLE_iterative.py:168,376-381,397).  Re-taken at the end of the round with the component
V-cycles of the auxiliary-space term cycled together (csr_multi_kernel; this file is written by
tools/write_mypre_profile.py).  Bench object of the kernel-trace pass:

```
""" + json.dumps(obj, indent=1) + """
```

Reading the tables below for this path (per BPCG iteration: 2 sweep calls, 1 auxiliary-space term, 1 residual SpMV):

* **Sweep** (`gs_enter_kernel` + `csr_stream_kernel<EpiGsFused>` x 2 colours + `gs_leave_kernel` per call):
  %.1f us per colour launch, %.1f us entry gather, %.1f us exit scatter -> %.2f ms per call.  Drawn per colour launch
  ~506 MB (PMC, FETCH x 2 + WRITE) for ~485 MB algorithmic (half of P A P^T as stored 0.26 GB, the colour's rows of the
  inverse blocks 0.09, x 0.03, y in / out 0.06, the other colour's y through the operand copy ~0.045): **drawn /
  algorithmic = 1.04** (round 1's form: ~2.4 GB per pass for a 0.63-GB matrix).  The entry gather draws 395 MB for 240
  algorithmic: with two colours every 64-byte line of x and y holds dofs of both, so each half of the gather touches
  every line (1.6 x); per sweep call 1.61 GB drawn for 1.27 GB algorithmic = 1.27 x.  Inside the loop the FORWARD sweep starts
  from zero: its entry gathers x only, its first colour is `gs_first_color_kernel` (a block solve, no pass over A) and the
  backward sweep's entry gathers y only -- the averages above mix those launches with the stand-alone calls.
* **Auxiliary-space term**: T^T and T SpMVs (`csr_stream_kernel<EpiAxpby>`, %.1f us each) around ONE joint cycle over the
  shared hierarchy (levels 2 460 375 / 253 314 / 12 487 / 1 487) for the three components: %.3f ms per apply for %.2f GB
  algorithmic (%.2f of peak) against 0.97 ms / 4.1 GB for three cycles one after the other
  (profiles/r03_ab_amg_batch.txt).  Per level (kernel-trace of this run):

""" % (avg("EpiGsFused"), avg("gs_enter_kernel"), avg("gs_leave_kernel"), obj["sweep_call"]["avg_ms"], avg("T / T^T"),
       aux["avg_ms"], aux["algorithmic_bytes"] / 1e9, aux["frac"]) + "\n".join(lines) + """

  The fine level is 16 784 workgroups of 1024 products each; its residual moves 412 MB (216 MB of matrix at 12 bytes per
  entry + 3 x (x, b, r)).  The Galerkin operator of level 1 has ~32 entries per row (116 MB per product): two lanes share
  a (row, k) pair there.
* The headline kernels in the same trace: C23, C1, C4 and the block Jacobi as in r03_rocprof_summary.md.

""" + body
open(os.path.join(ROOT, "profiles", "r03_rocprof_mypre_a.md"), "w").write(txt)
print("\n".join(lines))
