#!/bin/bash
cd $(dirname $0)/..; mkdir -p gpurun_out
exec < /dev/null
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
timeout 600 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -12 | tee gpurun_out/pytest_all_r05u.log
