#!/bin/bash
# Interleaved same-box A/B of the default-preconditioner path (roofline_mypre_a_gs of bench.py):
#   tools/ab_mypre.sh "A:VAR=0 B:VAR=1" [rounds]
VARS=${1}; ROUNDS=${2:-2}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for r in $(seq 1 $ROUNDS); do
  for v in $VARS; do
    name=${v%%:*}; assign=${v#*:}
    env ${assign//,/ } python $REPO/bench.py --mypre-a 1 --hdg 0 --secondary 0 --cpu-iters 0 --steps 50 --warmup 10 --windows 1 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); m=d['roofline_mypre_a_gs']
print('$name r$r it/s %.1f ms/it %.4f | C1+preA %.4f sweep %.4f aux %.4f (%.2f GB) | setup %.2f s k %.4f' % (m['iters_per_s'], m['ms_per_iteration'], m['C1_with_whole_preA_ms_in_loop'], m['sweep_call']['avg_ms'], m['auxiliary_space_term']['avg_ms'], m['auxiliary_space_term']['algorithmic_bytes']/1e9, m['setup_s']['amg_hierarchies_colouring_permutation'], m['scale_factor_k']))"
  done
done
