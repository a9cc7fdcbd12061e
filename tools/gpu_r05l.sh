#!/bin/bash
# A/B of the main loop's pipeline depth (CERES_HIP_AB_PIPE_DEPTH=2: variants/libceres_hip_d2.so) on the Venice shape, fp64 and fp32 tiles,
# interleaved processes on one box; operator parity of the variant first.
cd $(dirname $0)/..; mkdir -p gpurun_out
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
V=$(pwd)/ceres-solver_amd/csrc/variants/libceres_hip_d2.so
CERES_HIP_LIBRARY=$V timeout 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_real_visibility.py tests/test_gpu_edge_cases.py -m gpu -q -x --timeout 600 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -8 | tee gpurun_out/pytest_d2_r05l.log
python tools/kernel_times.py venice1778 > /dev/null 2>&1
for R in 1 2; do for ST in 0 1; do for L in default d2; do
  if [ $L = d2 ]; then export CERES_HIP_LIBRARY=$V; else unset CERES_HIP_LIBRARY; fi
  STORAGE=$ST timeout 300 python tools/kernel_times.py venice1778 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L', 'storage', d['storage'], {k:d.get(k) for k in ('jtjx_ms','sx_ms','read_stream_ms')})" | tee -a gpurun_out/ab_pipe_depth_r05l.txt
done; done; done
