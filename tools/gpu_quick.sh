#!/bin/bash
# quick GPU iteration: tests + kernel timings (+ fp32 storage) + a short bench of both solvers
TAG=${1:-q}
cd $(dirname $0)/..
OUT=$(pwd)/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --timeout 300 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -25 | tee $OUT/pytest_$TAG.log | tail -${PYTAIL:-4}
for W in venice1778 ${EXTRA_WORKLOADS}; do timeout 600 python tools/kernel_times.py $W 2>/dev/null | tee -a $OUT/ktimes_$TAG.jsonl; echo; done
STORAGE=1 timeout 600 python tools/kernel_times.py venice1778 2>/dev/null | tee -a $OUT/ktimes_$TAG.jsonl; echo
for S in cgnr iterative_schur; do
timeout 600 python bench.py --no-cpu-baseline --both-solvers 0 --solver $S 2>/dev/null | tee $OUT/bench_${S}_$TAG.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['solver'], 'steps/s', d['value'], 'ms', d['ms_per_step'], 'its', d['config']['cg_iterations_per_step'], d['roofline']['frac'], d['extra']['solve_phases_ms'])"
done
