#!/bin/bash
cd $(dirname $0)/..; mkdir -p gpurun_out
exec < /dev/null
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_default_r05x.json 2> gpurun_out/bench_default_r05x.err; echo "rc=$?"
timeout 20 python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/bench_default_r05x.json").read().splitlines() if l.startswith("{")][-1])
    print({k: d[k] for k in ("value", "ms_per_step")}, "sx", d["roofline"]["frac"], "jtjx", d["roofline_jtjx"]["frac"], "oracle", d["oracle_check"]["step_rel_diff_vs_oracle"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["step_rel_diff_vs_gpu"])
    print(d["roofline"]["kernel"], "|", d["roofline"]["traffic_source"][:90]); print(d["extra"]["fp32_tiles"])
except Exception as ex:
    print("unreadable:", ex)
PY
tail -3 gpurun_out/bench_default_r05x.err | cut -c1-200
