#!/bin/bash
# Interleaved A/B of kernel variants on the GPU box: tools/ab_bench.sh "base nt" [rounds] [bench args]
VARS=${1:-"base nt"}; ROUNDS=${2:-3}; ARGS=${3:-"--cpu-iters 0 --steps 100 --warmup 10"}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for r in $(seq 1 $ROUNDS); do
  for v in $VARS; do
    NSS_LIB_PATH=$REPO/build/ab/libnss_$v.so python $REPO/bench.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); k=d['kernel_ms']; h=d['hbm_GBs']
print('$v r$r it/s %.1f ms %.4f | C1 %.4f C23 %.4f C4 %.4f sums %.4f spmvA %.4f triad %.4f | C23/triad %.3f' % (d['value'], d['ms_per_step'], k['C1_BT_preA'], k['C23_A_B'], k['C4_update'], k['sum_kernels'], k['spmv_A_plain'], k['triad_1.6GB'], h['spmv_AB_fused_C23']/h['stream_triad']))"
  done
done
