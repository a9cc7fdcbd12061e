#!/bin/bash
# quick GPU iteration: tests + kernel timings at both workgroup sizes
TAG=${1:-q}
cd $(dirname $0)/..
OUT=$(pwd)/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q --timeout 300 -x 2>&1 | tail -15 | tee $OUT/pytest_$TAG.log
for B in 1024 512; do CERES_HIP_BAL_BLOCK=$B timeout 600 python tools/kernel_times.py venice1778 2>/dev/null | tee -a $OUT/ktimes_$TAG.jsonl; done
