#!/bin/bash
# Round 6, session e: camera exchange split from the inversion, collapse+exchange; multirank tests, shard ceilings (Venice, synthetic10M)
TAG=${1:-r06e}
REPO=$(cd $(dirname $0)/.. && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
exec < /dev/null
echo "== multirank + lm + solvers tests ($(date +%T))"
timeout 1500 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_lm_step.py tests/test_gpu_solvers.py tests/test_gpu_shapes.py -m gpu -q -x --timeout 600 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -8 | tee $OUT/pytest_$TAG.log
for SOLVER in iterative_schur cgnr; do
  timeout 600 python tools/shard_step.py venice1778 1,2,4,8 $SOLVER 20 2>&1 | grep "^{" | tee -a $OUT/shard_step_$TAG.jsonl
done
echo "== synthetic10M ($(date +%T))"
timeout 1200 python tools/shard_step.py synthetic10M 1,8 iterative_schur 5 2>&1 | grep "^{\|Error\|error" | tee -a $OUT/shard_step_$TAG.jsonl
cd /tmp && export TMPDIR=/tmp
for CASE in "venice1778 8 iterative_schur" "venice1778 8 cgnr" "synthetic10M 8 iterative_schur"; do
  set -- $CASE
  NAME=$1_n$2_$3
  echo "== trace $NAME ($(date +%T))"
  rm -rf /tmp/trace_$NAME
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$NAME -o t -- python $REPO/tools/shard_step.py $1 $2 $3 4 2 > /dev/null 2> $OUT/trace_${NAME}_$TAG.err
  F=$(timeout 20 find /tmp/trace_$NAME -name "*kernel_trace.csv" | head -1)
  if [ -n "$F" ]; then python $REPO/tools/kernel_timeline.py "$F" 90 > $OUT/timeline_${NAME}_$TAG.txt; grep -n "fillBuffer" $OUT/timeline_${NAME}_$TAG.txt | tail -3; tail -64 $OUT/timeline_${NAME}_$TAG.txt | head -40 | cut -c1-150; else echo "no trace"; tail -3 $OUT/trace_${NAME}_$TAG.err; fi
done
echo "== done ($(date +%T))"
