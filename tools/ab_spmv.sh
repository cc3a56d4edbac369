#!/bin/bash
# Interleaved same-box A/B of the plain SpMV:  tools/ab_spmv.sh "<lib a> <lib b> ..." "<dim grid inflate matrix>" ... [rounds]
# (lib = name under build/ab/ or "default").  Run on the GPU box.
LIBS=$1; shift
ROUNDS=3; CFGS=("$@")
for r in $(seq $ROUNDS); do
  for cfg in "$@"; do
    for l in $LIBS; do
      if [ $l = default ]; then unset NSS_LIB_PATH; else export NSS_LIB_PATH=build/ab/libnss_$l.so; fi
      read d g i m <<< "$cfg"
      python tools/spmv_probe.py $d $g $i 50 $m 2>&1 | grep -v amdgpu.ids
    done
  done
done
