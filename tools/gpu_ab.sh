#!/bin/bash
cd $(dirname $0)/..
OUT=$(pwd)/gpurun_out; mkdir -p $OUT; TAG=${1:-ab}
python tools/kernel_times.py venice1778 > /dev/null 2>&1
for V in "CERES_HIP_SKEW=1" "CERES_HIP_SKEW=0" "CERES_HIP_SKEW=1" "CERES_HIP_SKEW=0"; do
  echo "== $V"; env $V timeout 300 python tools/kernel_times.py venice1778 2>/dev/null | tee -a $OUT/ab_$TAG.jsonl | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('jtjx_ms','jtjx_frac','sx_ms','sx_frac')})"
done
