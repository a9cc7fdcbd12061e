export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
timeout 300 python tools/kernel_times.py venice1778 > /dev/null 2>&1
for R in 1 2; do for W in strided blocked; do
  echo -n "$W "
  CERES_HIP_TILE_WALK=$W timeout 300 python tools/kernel_times.py venice1778 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('jtjx_ms','jtjx_frac','sx_ms','sx_frac')})"
done; done
CERES_HIP_TILE_WALK=blocked timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "ladybug" 2>&1 | grep -E "passed|failed" | tail -2
