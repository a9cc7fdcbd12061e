#!/bin/bash
# x_f of the popular cameras in LDS (BalPlan::xhot_cam): parity of the affected test files, then A/B against CERES_HIP_XHOT=0 on the
# Venice / Ladybug shapes (fp64 and fp32 tiles), interleaved processes on one box.
cd $(dirname $0)/..; mkdir -p gpurun_out
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
timeout 1200 python -m pytest tests/test_gpu_operators.py tests/test_gpu_real_visibility.py tests/test_gpu_edge_cases.py tests/test_gpu_shapes.py tests/test_gpu_solvers.py tests/test_gpu_fullsize.py tests/test_gpu_remainder.py -m gpu -q -x --timeout 900 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -8 | tee gpurun_out/pytest_xhot_r05n.log
for WL in venice1778 ladybug1723; do
python tools/kernel_times.py $WL > /dev/null 2>&1
for R in 1 2; do for ST in 0 1; do for X in 1 0; do
  CERES_HIP_XHOT=$X STORAGE=$ST timeout 300 python tools/kernel_times.py $WL 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$WL xhot=$X', 'storage', d['storage'], {k:d.get(k) for k in ('jtjx_ms','sx_ms','read_stream_ms')})" | tee -a gpurun_out/ab_xhot_r05n.txt
done; done; done; done
