#!/bin/bash
cd $(dirname $0)/..; mkdir -p gpurun_out
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_lm_step.py tests/test_gpu_bal_frontend.py tests/test_gpu_operators.py -m gpu -q -x --timeout 900 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -8 | tee gpurun_out/pytest_r05q.log
for R in 1 2; do
  timeout 300 python tools/kernel_times.py venice1778 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('jtjx_ms','sx_ms','back_substitute_ms','schur_init_ms','schur_jacobi_ms')}, d['schur_solve'])" | tee -a gpurun_out/ktimes_r05q.txt
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --extra-synthetic10m 0 --extra-other-shapes 0 --extra-dense-cholesky 0 --minimizer-iterations 0 --host-boundary-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline_jtjx']['frac'], d.get('oracle_check'), d['extra'].get('fp32_tiles'))" | tee -a gpurun_out/ktimes_r05q.txt
