#!/bin/bash
# rocprofv3 kernel trace of the fused loops at a small configuration (launch-bound regime):
#   tools/profile_small.sh <tag> "<bench args>"      e.g.  r02_cfg2 "--dim 2 --grid 183"
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof
TAG=${1:-small}
ARGS=${2:-"--dim 2 --grid 183"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o ${TAG} -- python3 $REPO/bench.py $ARGS --cpu-iters 0 --steps 2000 --warmup 100 > $OUT/${TAG}.json 2> $OUT/${TAG}.err || exit 1
python3 - <<PY
import csv, glob
p = glob.glob("$OUT/${TAG}_trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(p)))
print("| kernel | calls | avg (us) | total (ms) |")
print("|---|---|---|---|")
for r in rows[:14]:
    print("| %s | %s | %.2f | %.3f |" % (r["Name"].replace("nss::", "")[:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
