#!/bin/bash
# r06u: index words one tile ahead in the unpipelined tile loop (A/B against a build without: variants/libceres_hip_nopre.so) and
# plain tile stores for tile streams that fit the Infinity Cache (that run: plain by default; since then opt-in, CERES_HIP_PLAIN_TILE_STORES=1)
REPO=$(cd $(dirname $0)/.. && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
exec < /dev/null
echo "== gpu tests ($(date +%T))"
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -5 | tee $OUT/pytest_gpu_r06u.log
NOPRE=$REPO/ceres-solver_amd/csrc/variants/libceres_hip_nopre.so
for REP in 1 2; do
for V in "pre plain" "nopre plain" "pre nt" "nopre nt"; do
  set -- $V
  E=""
  [ $1 = nopre ] && E="CERES_HIP_LIBRARY=$NOPRE"
  [ $2 = nt ] && E="$E CERES_HIP_PLAIN_TILE_STORES=0"
  echo "== $V ($(date +%T))"
  for CASE in "venice1778 8 iterative_schur 30 2" "ladybug1723 1 iterative_schur 40" "dubrovnik16 1 cgnr 60" "venice1778 1 iterative_schur 20"; do
    env $E timeout 600 python tools/shard_step.py $CASE 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); d['variant'] = '$V'; print(json.dumps({k: d[k] for k in ('variant', 'workload', 'ranks', 'ms_per_step', 'cg_ms', 'setup_ms', 'back_substitute_ms') if k in d}))" | tee -a $OUT/prefetch_ab_r06u.jsonl
  done
done
done
