#!/bin/bash
# bench.py --gpus 2 typed without a launcher on a one-GPU box (both ranks on device 0: validation mode) at the HEADLINE workload: the
# shards are large enough (39 k tiles) for the LDS copies of x to be on; the assembled step against the oracle at full size.
cd $(dirname $0)/..; mkdir -p gpurun_out
exec < /dev/null
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
for SV in iterative_schur cgnr; do
  timeout 280 python bench.py --gpus 2 --workload venice1778 --solver $SV --steps 5 --warmup 2 --no-cpu-baseline --oracle-check 1 --extra-synthetic10m 0 --extra-other-shapes 0 --extra-dense-cholesky 0 --minimizer-iterations 0 --host-boundary-steps 0 2> gpurun_out/bench_self2_venice_${SV}_r05w.err | tail -1 > gpurun_out/bench_self2_venice_${SV}_r05w.json
  timeout 20 python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_self2_venice_${SV}_r05w.json").read())
    print("$SV", {k: d[k] for k in ("n_gpus", "value", "ms_per_step")}, d["config"]["parallelism"][:80], d["config"].get("collectives_per_step"), d.get("oracle_check"))
except Exception as ex:
    print("$SV unreadable", ex)
PY
  tail -2 gpurun_out/bench_self2_venice_${SV}_r05w.err | cut -c1-200
done
