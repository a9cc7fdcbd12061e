#!/bin/bash
# Builds an A/B variant of the library from the same sources with extra -D flags:
#   tools/build_variant.sh NAME -DCERES_HIP_AB_YE_STORE=1   ->  ceres-solver_amd/csrc/variants/libceres_hip_NAME.so
# Run on the CPU container (hipcc cross-compiles gfx950); the .so travels to the GPU box with the snapshot and is selected with
# CERES_HIP_LIBRARY=ceres-solver_amd/csrc/variants/libceres_hip_NAME.so.
set -e
NAME=$1; shift
REPO=$(cd $(dirname $0)/.. && pwd)
CSRC=$REPO/ceres-solver_amd/csrc
OUT=$CSRC/variants; TMP=$(mktemp -d)
mkdir -p $OUT
SRCS=$(python3 -c "import sys; sys.path.insert(0, '$REPO/ceres-solver_amd'); import build; print(' '.join(build.SOURCES))")
OBJS=""
for S in $SRCS; do
  O=$TMP/$(basename $S).o
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics "$@" -x hip -c $CSRC/$S -o $O &
  OBJS="$OBJS $O"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libceres_hip_$NAME.so $OBJS -L/opt/rocm/lib -lrccl
rm -rf $TMP
echo $OUT/libceres_hip_$NAME.so
