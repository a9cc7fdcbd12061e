#!/bin/bash
cd $(dirname $0)/..; mkdir -p gpurun_out
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
python tools/kernel_times.py venice1778 > /dev/null 2>&1
for R in 1 2 3 4; do for X in 1 0; do
  CERES_HIP_XHOT=$X timeout 300 python tools/kernel_times.py venice1778 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('venice1778 xhot=$X', 'storage', d['storage'], {k:d.get(k) for k in ('jtjx_ms','sx_ms','read_stream_ms')})" | tee -a gpurun_out/ab_xhot_r05o.txt
done; done
for X in 1 0; do CERES_HIP_XHOT=$X timeout 300 python tools/kernel_times.py ladybug1723 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ladybug xhot=$X', {k:d.get(k) for k in ('jtjx_ms','sx_ms')})" | tee -a gpurun_out/ab_xhot_r05o.txt; done
for X in 1 0; do echo "long cg xhot=$X"; CERES_HIP_XHOT=$X timeout 300 python tools/gpu_long_cg.py 2>/dev/null | tail -2 | tee -a gpurun_out/ab_xhot_r05o.txt; done
