#!/bin/bash
cd $(dirname $0)/..; mkdir -p gpurun_out
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -12 | tee gpurun_out/pytest_all_r05r.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_r05r; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_r05r -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --both-solvers 0 --minimizer-iterations 0 --host-boundary-steps 0 --extra-synthetic10m 0 --extra-other-shapes 0 --extra-dense-cholesky 0 > $GRAFT_REPO_ROOT/gpurun_out/bench_under_rocprof_r05r.json 2>/dev/null
F=$(find /tmp/prof_r05r -name "*kernel_stats.csv" | head -1); cp $F $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_r05r.csv; head -14 $F | cut -c1-160
