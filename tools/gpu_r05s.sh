#!/bin/bash
# many-camera regime (synthetic1M: 50 000 cameras, 3 M observations; hybrid accumulation): what the camera-part gathers cost there
# (CERES_HIP_AB_ABLATE 32: the 440 lowest camera ids = the most popular ones read one line; 64: every lane does) — timing only
cd $(dirname $0)/..; mkdir -p gpurun_out
exec < /dev/null
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
timeout 300 python tools/kernel_times.py synthetic1M > /dev/null 2>&1
for R in 1 2; do for L in default ab32 ab64; do
  if [ $L = default ]; then unset CERES_HIP_LIBRARY; else export CERES_HIP_LIBRARY=$(pwd)/ceres-solver_amd/csrc/variants/libceres_hip_$L.so; fi
  timeout 200 python tools/kernel_times.py synthetic1M 2>/dev/null | tail -1 | timeout 20 python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L', {k:d.get(k) for k in ('jtjx_ms','sx_ms','back_substitute_ms','schur_init_ms')})" | tee -a gpurun_out/ablation_many_cameras_r05s.txt
done; done
