#!/bin/bash
TAG=${1:-r06j}
REPO=$(cd $(dirname $0)/.. && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
exec < /dev/null
echo "== gpu tests ($(date +%T))"
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -12 | tee $OUT/pytest_$TAG.log
echo "== bench --gpus 8 one-GPU validation, oracle check ($(date +%T))"
for SOLVER in iterative_schur cgnr; do
  timeout 1200 python bench.py --gpus 8 --workload venice1778 --oracle-check 1 --steps 3 --warmup 1 --solver $SOLVER > $OUT/bench_n8_one_gpu_validation_${SOLVER}_$TAG.json 2> $OUT/bench_n8_$SOLVER_$TAG.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$OUT/bench_n8_one_gpu_validation_${SOLVER}_$TAG.json").read().splitlines() if l.startswith("{")][-1])
    print("$SOLVER", d["n_gpus"], d["config"]["parallelism"][:90], "collectives", d["config"]["collectives_per_step"], "oracle", d["oracle_check"])
except Exception as ex:
    print("unreadable:", ex); print(open("$OUT/bench_n8_$SOLVER_$TAG.err").read()[-1500:])
PY
done
echo "== done ($(date +%T))"
