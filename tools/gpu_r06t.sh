#!/bin/bash
# r06t: camera-exchange kernel A/B on one rank's eighth of Venice; timelines of Ladybug (1 rank) and Venice (1 of 8) at HEAD
REPO=$(cd $(dirname $0)/.. && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
exec < /dev/null
for FEW in 1 0 1 0; do
  echo "== CAM_EXCHANGE_FEW=$FEW"
  CERES_HIP_CAM_EXCHANGE_FEW=$FEW timeout 600 python tools/shard_step.py venice1778 8 iterative_schur 30 2 2>&1 | grep "^{" | cut -c1-200 | tee -a $OUT/cam_exchange_ab_r06t.jsonl
done
cd /tmp && export TMPDIR=/tmp
for CASE in "venice1778 8 iterative_schur 1" "venice1778 8 iterative_schur 0" "ladybug1723 1 iterative_schur 1" "dubrovnik16 1 cgnr 1" "venice1778 8 cgnr 1"; do
  set -- $CASE
  NAME=$1_n$2_$3_few$4
  rm -rf /tmp/trace_$NAME
  CERES_HIP_CAM_EXCHANGE_FEW=$4 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$NAME -o t -- python $REPO/tools/shard_step.py $1 $2 $3 4 2 > /dev/null 2> $OUT/trace_${NAME}_r06t.err
  F=$(timeout 20 find /tmp/trace_$NAME -name "*kernel_trace.csv" | head -1)
  if [ -n "$F" ]; then python $REPO/tools/kernel_timeline.py "$F" 90 > $OUT/timeline_${NAME}_r06t.txt; echo "timeline $NAME: $(wc -l < $OUT/timeline_${NAME}_r06t.txt) lines"; fi
done
