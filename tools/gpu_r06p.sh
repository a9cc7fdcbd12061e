#!/bin/bash
REPO=$(cd $(dirname $0)/.. && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
exec < /dev/null
timeout 1500 python -m pytest tests/test_gpu_streamed_upload.py tests/test_gpu_lm_step.py -m gpu -q -x --timeout 600 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -25
