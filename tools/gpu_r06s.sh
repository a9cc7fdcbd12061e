#!/bin/bash
REPO=$(cd $(dirname $0)/.. && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
exec < /dev/null
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -5
for S in iterative_schur cgnr; do timeout 600 python tools/shard_step.py venice1778 1,2,4,8 $S 20 2>&1 | grep "^{" | cut -c1-140 | tee -a $OUT/shard_step_r06s.jsonl; done
timeout 300 python tools/shard_step.py ladybug1723 1 iterative_schur 30 2>&1 | grep "^{"
timeout 300 python tools/shard_step.py dubrovnik16 1 cgnr 50 2>&1 | grep "^{"
