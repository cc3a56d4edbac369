#!/bin/bash
# One rocprofv3 PMC pass (FETCH_SIZE) of bench.py on the GPU box; prints the average per kernel (top entries) with the
# x2 correction calibrated on the triad kernel.   tools/pmc_fetch.sh <tag> "<bench args>"    (env switches: set them
# in front of this script)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof
TAG=${1:-pmc}
ARGS=${2:-"--steps 20 --warmup 5 --windows 1 --cpu-iters 0 --hdg 0 --secondary 0 --mypre-a 0"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch -o ${TAG} -- python3 $REPO/bench.py $ARGS > $OUT/${TAG}_fetch.json 2> $OUT/${TAG}_fetch.err || { tail -5 $OUT/${TAG}_fetch.err; exit 1; }
python3 - "$OUT/${TAG}_fetch" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
path = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
acc = defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(path)):
    if r.get("Counter_Name") == "FETCH_SIZE":
        a = acc[r["Kernel_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
tri = [k for k in acc if "triad_kernel" in k]
scale = 1.0
if tri:
    v, n = acc[tri[0]]
    # triad reads 2 x 8 x 2^26 bytes per launch; FETCH_SIZE counts 64-byte units (guide) -> calibrate
    scale = (16.0 * (1 << 26)) / (v / n)
    print("calibration: triad FETCH_SIZE avg %.4g -> %.1f bytes per unit" % (v / n, scale))
rows = sorted(((v / n * scale, n, k) for k, (v, n) in acc.items()), reverse=True)[:8]
for b, n, k in rows:
    print("%10.1f MB fetched / launch  x%-5d %s" % (b / 1e6, n, k.replace("nss::", "")[:110]))
PY
