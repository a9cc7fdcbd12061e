#!/bin/bash
TAG=${1:-r06m}
REPO=$(cd $(dirname $0)/.. && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
exec < /dev/null
for V in "X=1" "CERES_HIP_BAL_BLOCK=512" "CERES_HIP_BAL_BLOCK=512 CERES_HIP_PIPELINE=0"; do
  for N in 8 4 2; do echo "-- $V venice N=$N"; env $V timeout 300 python tools/shard_step.py venice1778 $N iterative_schur 20 2 2>&1 | grep "^{" | cut -c1-120; done
  echo "-- $V venice N=8 cgnr"; env $V timeout 300 python tools/shard_step.py venice1778 8 cgnr 20 2 2>&1 | grep "^{" | cut -c1-120
  echo "-- $V ladybug N=1"; env $V timeout 300 python tools/shard_step.py ladybug1723 1 iterative_schur 30 2>&1 | grep "^{" | cut -c1-200
  echo "-- $V venice N=1"; env $V timeout 300 python tools/shard_step.py venice1778 1 iterative_schur 20 2>&1 | grep "^{" | cut -c1-200
done
echo "== done ($(date +%T))"
