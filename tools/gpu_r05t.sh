#!/bin/bash
cd $(dirname $0)/..; mkdir -p gpurun_out
exec < /dev/null
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
timeout 500 python -m pytest tests/test_gpu_shapes.py tests/test_plan_cpu.py -q -x --timeout 300 -k "read_from_lds or staged" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -15 | tee gpurun_out/pytest_r05t.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
