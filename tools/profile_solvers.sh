#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace stats of the three fused loops (BPCG v2, BPCG v1, MINRES)
# at the 1e7-DoF config, through their drop-in entry points (tools/bench_solvers.py --only=cfg4).
# Writes gpurun_out/prof/<tag>_solvers.md (top kernels by total time).
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof
TAG=${1:-r01}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_solvers -o ${TAG}s -- python3 $REPO/tools/bench_solvers.py --only=cfg4 > $OUT/${TAG}_solvers_table.md 2> $OUT/${TAG}_solvers.err || exit 1
python3 - "$OUT/${TAG}_solvers" "$OUT/${TAG}_solvers_table.md" > $OUT/${TAG}_solvers.md <<'PY'
import csv, glob, os, sys
root, table = sys.argv[1], sys.argv[2]
path = glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(path)))
print("# rocprofv3 --kernel-trace --stats -- python3 tools/bench_solvers.py --only=cfg4\n")
print("BPCG v2, BPCG v1 and MINRES through their entry points at 1e7 DoF (block Jacobi bs=3); rates printed by the same run (under the profiler):\n")
print(open(table).read())
print("| kernel | calls | avg (us) | total (ms) | % |\n|---|---|---|---|---|")
for r in rows[:28]:
    print("| %s | %s | %.2f | %.3f | %s |" % (r["Name"].replace("nss::", "")[:100], r["Calls"], float(r["AverageNs"]) / 1e3,
                                          float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
cat $OUT/${TAG}_solvers.md
