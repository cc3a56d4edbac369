#!/bin/bash
# Build a kernel A/B variant of libnsskrylov.so:  tools/build_variant.sh <name> "<extra hipcc flags>"
# -> build/ab/libnss_<name>.so (travels to the GPU box; select it with NSS_LIB_PATH).
set -e
NAME=$1; shift
FLAGS="$*"
SRC=$(dirname $0)/../navier-stokes-solver_amd/csrc
OUT=$(dirname $0)/../build/ab
mkdir -p $OUT/obj_$NAME
for f in $SRC/*.hip; do
  EXTRA=""; [ "$(basename $f)" = amg_setup.hip ] && EXTRA="-ffp-contract=off"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 $FLAGS $EXTRA -c $f -o $OUT/obj_$NAME/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libnss_$NAME.so $OUT/obj_$NAME/*.o
echo built $OUT/libnss_$NAME.so
