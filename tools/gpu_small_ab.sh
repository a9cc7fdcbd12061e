#!/bin/bash
# launch- / latency-bound shapes with A/B libraries (VARIANTS="base nont ..."): LM step time and phases per variant
REPO=$(cd $(dirname $0)/.. && pwd); cd $REPO
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
for R in 1 2; do for V in ${VARIANTS:-base}; do for WL in dubrovnik16 ladybug1723; do
  LIB=""; [ "$V" != "base" ] && LIB=$REPO/ceres-solver_amd/csrc/variants/libceres_hip_$V.so
  echo -n "$V $WL "
  CERES_HIP_LIBRARY=$LIB timeout 300 python bench.py --workload $WL --steps 200 --warmup 20 --no-cpu-baseline --minimizer-iterations 0 --host-boundary-steps 0 --scene-step-steps 0 --both-solvers 1 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'ms_per_step': d['ms_per_step'], 'cgnr': d['extra'].get('cgnr',{}).get('ms_per_step'), 'sx_ms': d['roofline']['avg_launch_ms'], 'phases': d['extra']['solve_phases_ms']}))"
done; done; done
