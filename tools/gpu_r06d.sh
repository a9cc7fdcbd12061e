#!/bin/bash
# Round 6, session d: full GPU suite on the exchange-in-producer build, shard ceilings, timelines at N = 2, 4, 8.
TAG=${1:-r06d}
REPO=$(cd $(dirname $0)/.. && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
exec < /dev/null
echo "== smoke ($(date +%T))"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | cut -c1-300
echo "== gpu tests ($(date +%T))"
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -25 | tee $OUT/pytest_$TAG.log
for SOLVER in iterative_schur cgnr; do
for N in 1 2 4 8; do
  timeout 300 python tools/shard_step.py venice1778 $N $SOLVER 20 2>&1 | grep "^{" | tee -a $OUT/shard_step_$TAG.jsonl
done; done
cd /tmp && export TMPDIR=/tmp
for CASE in "venice1778 8 iterative_schur" "venice1778 2 iterative_schur" "venice1778 4 iterative_schur"; do
  set -- $CASE
  NAME=$1_n$2_$3
  echo "== trace $NAME ($(date +%T))"
  rm -rf /tmp/trace_$NAME
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$NAME -o t -- python $REPO/tools/shard_step.py $1 $2 $3 6 > /dev/null 2> $OUT/trace_${NAME}_$TAG.err
  F=$(timeout 20 find /tmp/trace_$NAME -name "*kernel_trace.csv" | head -1)
  if [ -n "$F" ]; then python $REPO/tools/kernel_timeline.py "$F" 64 > $OUT/timeline_${NAME}_$TAG.txt; tail -26 $OUT/timeline_${NAME}_$TAG.txt | cut -c1-150; else echo "no trace"; tail -3 $OUT/trace_${NAME}_$TAG.err; fi
done
echo "== done ($(date +%T))"
