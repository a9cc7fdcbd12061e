#!/bin/bash
TAG=${1:-r06g}
REPO=$(cd $(dirname $0)/.. && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
exec < /dev/null
echo "== multirank tests ($(date +%T))"
timeout 1500 python -m pytest tests/test_gpu_multirank.py -m gpu -q -x --timeout 600 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4 | tee $OUT/pytest_$TAG.log
echo "== synthetic10M ($(date +%T))"
timeout 1200 python tools/shard_step.py synthetic10M 1,2,4,8 iterative_schur 5 2>&1 | grep "^{\|Error\|error" | tee -a $OUT/shard_step_$TAG.jsonl
timeout 1200 python tools/shard_step.py synthetic10M 1,8 cgnr 5 2>&1 | grep "^{\|Error\|error" | tee -a $OUT/shard_step_$TAG.jsonl
echo "== fused grid experiment ($(date +%T))"
for G in 256 192 128 96; do
  echo "-- grid $G"; CERES_HIP_FUSED_GRID=$G timeout 300 python tools/shard_step.py venice1778 8 iterative_schur 20 2 2>&1 | grep "^{" | cut -c1-200
done
for G in 256 128; do
  echo "-- grid $G ladybug N=1"; CERES_HIP_FUSED_GRID=$G timeout 300 python tools/shard_step.py ladybug1723 1 iterative_schur 20 2>&1 | grep "^{" | cut -c1-200
done
cd /tmp && export TMPDIR=/tmp
for CASE in "synthetic10M 8 iterative_schur"; do
  set -- $CASE
  NAME=$1_n$2_$3
  echo "== trace $NAME ($(date +%T))"
  rm -rf /tmp/trace_$NAME
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$NAME -o t -- python $REPO/tools/shard_step.py $1 $2 $3 4 2 > /dev/null 2> $OUT/trace_${NAME}_$TAG.err
  F=$(timeout 20 find /tmp/trace_$NAME -name "*kernel_trace.csv" | head -1)
  if [ -n "$F" ]; then python $REPO/tools/kernel_timeline.py "$F" 90 > $OUT/timeline_${NAME}_$TAG.txt; tail -54 $OUT/timeline_${NAME}_$TAG.txt | head -28 | cut -c1-150; else echo "no trace"; tail -3 $OUT/trace_${NAME}_$TAG.err; fi
done
echo "== done ($(date +%T))"
