#!/bin/bash
# Interleaved same-box A/B of run-time variants (environment switches): tools/ab_env.sh "A:VAR=0 B:VAR=1" [rounds] [bench args]
VARS=${1}; ROUNDS=${2:-3}; ARGS=${3:-"--cpu-iters 0 --hdg 0 --secondary 0 --mypre-a 0 --steps 200 --warmup 20"}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for r in $(seq 1 $ROUNDS); do
  for v in $VARS; do
    name=${v%%:*}; assign=${v#*:}
    env ${assign//,/ } python $REPO/bench.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); k=d['kernel_ms']; h=d['hbm_GBs']
print('$name r$r it/s %.1f ms %.4f | C1 %.4f C23 %.4f C4 %.4f sums %.4f spmvA(loop) %.4f triad %.4f | C23/triad %.3f' % (d['value'], d['ms_per_step'], k['C1_BT_preA'], k['C23_A_B'], k['C4_update'], k['sum_kernels'], k['spmv_A_plain_in_loop_cache_state'], k['triad_1.6GB'], h['spmv_AB_fused_C23']/h['stream_triad']))"
  done
done
