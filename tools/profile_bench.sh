#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + two PMC passes (FETCH_SIZE, WRITE_SIZE)
# of bench.py.  Summaries land in gpurun_out/prof/; copy the ones to keep into profiles/.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof
TAG=${1:-r01}
ARGS=${2:-"--steps 60 --warmup 10 --cpu-iters 0"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o ${TAG} -- python3 $REPO/bench.py $ARGS > $OUT/${TAG}_trace.json 2> $OUT/${TAG}_trace.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch -o ${TAG} -- python3 $REPO/bench.py $ARGS > $OUT/${TAG}_fetch.json 2> $OUT/${TAG}_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_write -o ${TAG} -- python3 $REPO/bench.py $ARGS > $OUT/${TAG}_write.json 2> $OUT/${TAG}_write.err || exit 1
find $OUT -name "*.csv" | head -50
python3 $REPO/tools/summarize_prof.py $OUT $TAG $OUT/${TAG}_traffic.json "$ARGS" > $OUT/${TAG}_summary.md
cat $OUT/${TAG}_summary.md
