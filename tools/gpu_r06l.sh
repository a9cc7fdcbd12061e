#!/bin/bash
TAG=${1:-r06l}
REPO=$(cd $(dirname $0)/.. && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
exec < /dev/null
echo "== shape tests ($(date +%T))"
timeout 1500 python -m pytest tests/test_gpu_shapes.py tests/test_gpu_remainder.py tests/test_gpu_operators.py -m gpu -q -x --timeout 600 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -8 | tee $OUT/pytest_$TAG.log
echo "== done ($(date +%T))"
