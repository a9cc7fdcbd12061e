#!/bin/bash
# the GPU suite twice more (flakiness check of the multi-rank tests and the spin waits), then smoke
REPO=$(cd $(dirname $0)/.. && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
exec < /dev/null
for i in 1 2; do
  echo "== run $i ($(date +%T))"
  timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
