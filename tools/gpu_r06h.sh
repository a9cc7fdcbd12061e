#!/bin/bash
TAG=${1:-r06h}
REPO=$(cd $(dirname $0)/.. && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
exec < /dev/null
echo "== host_driver ($(date +%T))"
timeout 300 ceres-solver_amd/host/host_driver 2>&1 | tail -6 | cut -c1-250
echo "== streamed upload tests ($(date +%T))"
timeout 900 python -m pytest tests/test_gpu_streamed_upload.py -m gpu -q -x --timeout 600 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -15 | tee $OUT/pytest_$TAG.log
echo "== bench ($(date +%T))"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --minimizer-iterations 0 --extra-synthetic10m 0 --extra-other-shapes 0 --extra-dense-cholesky 0 --extra-real-graph 0 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; tail -3 $OUT/bench_$TAG.err
python - <<PY
import json
try:
    d = json.loads([l for l in open("$OUT/bench_$TAG.json").read().splitlines() if l.startswith("{")][-1])
    print({k: d[k] for k in ("value", "ms_per_step")}, "sx", d["roofline"]["frac"], "jtjx", d["roofline_jtjx"]["frac"], "oracle", (d.get("oracle_check") or {}).get("step_rel_diff_vs_oracle"))
    hb = d["host_boundary"]; print("host boundary", {k: hb[k] for k in ("ms_per_step", "upload_ms")}, "streamed", hb["streamed"])
    print("shard_ceiling", json.dumps(d["extra"].get("shard_ceiling"))[:1500])
    print("phases", d["extra"]["solve_phases_ms"])
except Exception as ex:
    print("bench line unreadable:", ex)
PY
echo "== done ($(date +%T))"
