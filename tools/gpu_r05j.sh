cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
timeout 900 python -m pytest tests/test_gpu_remainder.py tests/test_gpu_operators.py tests/test_gpu_real_visibility.py tests/test_gpu_concurrency.py -m gpu -q -x --timeout 600 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -25 | tee gpurun_out/pytest_r05j.log
for P in 1 0; do
  echo "--- venice fp32 tiles, CERES_HIP_F32_PIPELINE=$P"
  STORAGE=1 CERES_HIP_F32_PIPELINE=$P timeout 300 python tools/kernel_times.py venice1778 2>/dev/null | tail -1 | tee gpurun_out/ktimes_venice_fp32_pipe${P}_r05j.json
done
echo "--- venice fp64"
timeout 300 python tools/kernel_times.py venice1778 2>/dev/null | tail -1 | tee gpurun_out/ktimes_venice_fp64_r05j.json
