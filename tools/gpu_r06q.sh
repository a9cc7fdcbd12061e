#!/bin/bash
# one-GPU validation of bench.py's N > 1 path on other workloads (timings meaningless)
REPO=$(cd $(dirname $0)/.. && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
exec < /dev/null
for CASE in "4 synthetic1M iterative_schur" "4 synthetic1M cgnr" "8 ladybug1723 cgnr" "3 ladybug1723 iterative_schur"; do
  set -- $CASE
  echo "== --gpus $1 $2 $3 ($(date +%T))"
  timeout 900 python bench.py --gpus $1 --workload $2 --solver $3 --steps 3 --warmup 1 --oracle-check 1 > $OUT/bench_n$1_$2_$3_r06q.json 2> $OUT/bench_n$1_$2_$3_r06q.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$OUT/bench_n$1_$2_$3_r06q.json").read().splitlines() if l.startswith("{")][-1])
    print(d["n_gpus"], d["config"]["cg_iterations_per_step"], d["config"]["termination"], d["config"]["collectives_per_step"], d["config"]["step_finite"], (d.get("oracle_check") or {}).get("step_rel_diff_vs_oracle"))
except Exception as ex:
    print("unreadable:", ex); print(open("$OUT/bench_n$1_$2_$3_r06q.err").read()[-1200:])
PY
done
