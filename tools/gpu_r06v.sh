#!/bin/bash
# r06v: the next tile's VALUES in flight in the unpipelined tile loop of a step's first pass (kInit / kCgnrInit; A/B against
# variants/libceres_hip_pre1.so = index words only, -DCERES_HIP_AB_PREFETCH=1), and the live PMC passes of bench.py (tools/pmc_live.py)
REPO=$(cd $(dirname $0)/.. && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
exec < /dev/null
echo "== gpu tests ($(date +%T))"
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -5 | tee $OUT/pytest_gpu_r06v.log
PRE1=$REPO/ceres-solver_amd/csrc/variants/libceres_hip_pre1.so
rm -f $OUT/values_prefetch_ab_r06v.jsonl
for REP in 1 2; do
for V in values index; do
  E=""
  [ $V = index ] && E="CERES_HIP_LIBRARY=$PRE1"
  echo "== $V ($(date +%T))"
  for CASE in "venice1778 8 iterative_schur 30 2" "ladybug1723 1 iterative_schur 40" "dubrovnik16 1 cgnr 60" "venice1778 1 iterative_schur 20" "venice1778 1 cgnr 20" "ladybug1723 1 cgnr 40"; do
    env $E timeout 600 python tools/shard_step.py $CASE 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); d['variant'] = '$V'; print(json.dumps({k: d[k] for k in ('variant', 'workload', 'solver', 'ranks', 'ms_per_step', 'cg_ms', 'setup_ms', 'back_substitute_ms') if k in d}))" | tee -a $OUT/values_prefetch_ab_r06v.jsonl
  done
done
done
echo "== pmc live ($(date +%T))"
timeout 900 python tools/pmc_live.py venice1778 > $OUT/pmc_live_r06v.json 2> $OUT/pmc_live_r06v.err; tail -30 $OUT/pmc_live_r06v.json; tail -3 $OUT/pmc_live_r06v.err
echo "== timelines ($(date +%T))"
cd /tmp && export TMPDIR=/tmp
for CASE in "venice1778 1 iterative_schur" "ladybug1723 1 iterative_schur"; do
  set -- $CASE
  NAME=$1_n$2_$3
  rm -rf /tmp/trace_$NAME
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$NAME -o t -- python $REPO/tools/shard_step.py $1 $2 $3 4 2 > /dev/null 2> $OUT/trace_${NAME}_r06v.err
  F=$(timeout 20 find /tmp/trace_$NAME -name "*kernel_trace.csv" | head -1)
  if [ -n "$F" ]; then python $REPO/tools/kernel_timeline.py "$F" 60 > $OUT/timeline_${NAME}_r06v.txt; grep "bal_fused_kernel<3" $OUT/timeline_${NAME}_r06v.txt | head -3; fi
done
echo "== done ($(date +%T))"
