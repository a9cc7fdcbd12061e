#!/bin/bash
cd $(dirname $0)/..; mkdir -p gpurun_out
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
python tools/kernel_times.py venice1778 > /dev/null 2>&1
for R in 1 2; do for L in default ab128; do
  if [ $L = default ]; then unset CERES_HIP_LIBRARY; else export CERES_HIP_LIBRARY=$(pwd)/ceres-solver_amd/csrc/variants/libceres_hip_$L.so; fi
  timeout 300 python tools/kernel_times.py venice1778 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L', {k:d.get(k) for k in ('jtjx_ms','sx_ms','back_substitute_ms','schur_init_ms','schur_jacobi_ms','cgnr_setup_ms')}, d['schur_solve'])" | tee -a gpurun_out/ablation_backsub_r05p.txt
done; done
