#!/bin/bash
# Round 6, session b: exchange inside the producers, mailbox read-back, opt-in phase events — tests, shard ceilings, timelines.
TAG=${1:-r06b}
REPO=$(cd $(dirname $0)/.. && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
exec < /dev/null
echo "== smoke ($(date +%T))"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | cut -c1-300
echo "== multirank + lm tests ($(date +%T))"
timeout 1500 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_lm_step.py tests/test_gpu_solvers.py -m gpu -q -x --timeout 600 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -15 | tee $OUT/pytest_$TAG.log
for SOLVER in iterative_schur cgnr; do
for N in 1 2 4 8; do
  echo "== shard $SOLVER venice1778 N=$N ($(date +%T))"
  timeout 300 python tools/shard_step.py venice1778 $N $SOLVER 20 2>&1 | grep "^{" | tee -a $OUT/shard_step_$TAG.jsonl
done; done
echo "== A/B ($(date +%T))"
for V in "CERES_HIP_MAILBOX=0" "CERES_HIP_FINAL_SYNC=0" "CERES_HIP_P2P_FUSE=0" "CERES_HIP_P2P_FENCES=1"; do
  for N in 1 8; do
    echo "-- $V N=$N"; env $V timeout 300 python tools/shard_step.py venice1778 $N iterative_schur 20 2>&1 | grep "^{" | tee -a $OUT/shard_step_ab_$TAG.jsonl
  done
done
cd /tmp && export TMPDIR=/tmp
for CASE in "venice1778 8 iterative_schur" "venice1778 1 iterative_schur" "venice1778 8 cgnr"; do
  set -- $CASE
  NAME=$1_n$2_$3
  echo "== trace $NAME ($(date +%T))"
  rm -rf /tmp/trace_$NAME
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$NAME -o t -- python $REPO/tools/shard_step.py $1 $2 $3 6 > /dev/null 2> $OUT/trace_${NAME}_$TAG.err
  F=$(timeout 20 find /tmp/trace_$NAME -name "*kernel_trace.csv" | head -1)
  if [ -n "$F" ]; then python $REPO/tools/kernel_timeline.py "$F" 64 > $OUT/timeline_${NAME}_$TAG.txt; tail -30 $OUT/timeline_${NAME}_$TAG.txt | cut -c1-170; else echo "no trace"; tail -3 $OUT/trace_${NAME}_$TAG.err; fi
done
echo "== done ($(date +%T))"
