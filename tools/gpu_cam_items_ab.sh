#!/bin/bash
# A/B of the camera-major item kernel (CERES_HIP_CAM_ITEMS_MFMA) and of the item length (CERES_HIP_CAM_CHUNK_MIN) on the small shapes, the
# Venice shape and the real / heavy-tailed graphs.  usage: MODES="auto 1 0" CHUNKS="64 128 256" tools/gpu_cam_items_ab.sh [notest]
cd $(dirname $0)/..
[ "$1" != "notest" ] && timeout 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_fullsize.py tests/test_gpu_real_visibility.py tests/test_gpu_lm_step.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -5
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
for M in ${MODES:-auto 1 0}; do for C in ${CHUNKS:-64}; do
  if [ "$M" = "auto" ]; then unset CERES_HIP_CAM_ITEMS_MFMA; else export CERES_HIP_CAM_ITEMS_MFMA=$M; fi
  echo "== CERES_HIP_CAM_ITEMS_MFMA=$M CERES_HIP_CAM_CHUNK_MIN=$C"
  for WL in ${WLS:-dubrovnik16 ladybug1723 venice1778}; do
    echo -n "$WL "
    CERES_HIP_CAM_CHUNK_MIN=$C timeout 300 python bench.py --workload $WL --steps 100 --warmup 10 --no-cpu-baseline --minimizer-iterations 0 --host-boundary-steps 0 --scene-step-steps 0 --both-solvers 1 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'ms_per_step': d['ms_per_step'], 'cgnr': d['extra'].get('cgnr',{}).get('ms_per_step'), 'phases': d['extra']['solve_phases_ms']}))"
  done
  [ -n "$GRAPHS" ] && CERES_HIP_CAM_CHUNK_MIN=$C timeout 300 python tools/real_graph_times.py $GRAPHS 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['graph'], d['schur_solve'], d['cgnr_solve'])"
done; done
