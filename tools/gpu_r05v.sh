#!/bin/bash
cd $(dirname $0)/..; mkdir -p gpurun_out
exec < /dev/null
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
timeout 200 python tools/kernel_times.py venice1778 > /dev/null 2>&1
for R in 1 2; do for B in default 512; do
  if [ $B = default ]; then unset CERES_HIP_BAL_BLOCK; else export CERES_HIP_BAL_BLOCK=$B; fi
  timeout 200 python tools/kernel_times.py venice1778 2>/dev/null | tail -1 | timeout 20 python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('block $B', {k:d.get(k) for k in ('sx_ms','back_substitute_ms','schur_init_ms','schur_jacobi_ms')}, d['schur_solve']['backsub_ms'])" | tee -a gpurun_out/ab_backsub_block_r05v.txt
done; done
