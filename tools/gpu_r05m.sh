#!/bin/bash
# Timing-only ablations of the S.x / JtJx streaming kernels (CERES_HIP_AB_ABLATE bits, kernels_bal.inc), Venice shape, fp64 and fp32 tiles.
cd $(dirname $0)/..; mkdir -p gpurun_out
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
python tools/kernel_times.py venice1778 > /dev/null 2>&1
for ST in 1 0; do for L in ${VARIANTS:-default ab1 ab2 ab4 ab8 ab31 default}; do
  if [ $L = default ]; then unset CERES_HIP_LIBRARY; else export CERES_HIP_LIBRARY=$(pwd)/ceres-solver_amd/csrc/variants/libceres_hip_$L.so; fi
  STORAGE=$ST timeout 300 python tools/kernel_times.py venice1778 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L', 'storage', d['storage'], {k:d.get(k) for k in ('jtjx_ms','sx_ms','read_stream_ms')})" | tee -a gpurun_out/ablations_r05m.txt
done; done
