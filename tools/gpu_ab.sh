#!/bin/bash
# A/B of kernel variants (env switches) on the cached Venice-shaped problem
cd $(dirname $0)/..
OUT=$(pwd)/gpurun_out; mkdir -p $OUT; TAG=${1:-ab}
python tools/kernel_times.py venice1778 > /dev/null 2>&1
for V in "" "CERES_HIP_BAL_BLOCK=512"; do
  echo "== $V"; env $V timeout 300 python tools/kernel_times.py venice1778 2>/dev/null | tee -a $OUT/ab_$TAG.jsonl | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('jtjx_ms','jtjx_frac','sx_ms','sx_frac','back_substitute_ms','schur_init_ms','read_stream_ms','read_stream_GBs','block_jacobi_ms','schur_jacobi_ms')})"
done
timeout 600 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -5
