#!/bin/bash
# full default bench line (all legs), as the driver runs it
TAG=${1:-r06o}
REPO=$(cd $(dirname $0)/.. && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
exec < /dev/null
echo "== bench default ($(date +%T))"
T0=$(date +%s); timeout 1500 python bench.py > $OUT/bench_default_$TAG.json 2> $OUT/bench_default_$TAG.err; echo "bench wall seconds: $(( $(date +%s) - T0 ))"; tail -2 $OUT/bench_default_$TAG.err
python - <<PY
import json
d = json.loads([l for l in open("$OUT/bench_default_$TAG.json").read().splitlines() if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "steps", "warmup")}, "sx", d["roofline"]["frac"], "jtjx", d["roofline_jtjx"]["frac"], "step_roofline", d["step_roofline"]["frac"])
print("oracle", (d.get("oracle_check") or {}).get("step_rel_diff_vs_oracle"), "cpu", {k: d["cpu_baseline"][k] for k in ("value", "cores", "step_rel_diff_vs_gpu")})
e = d["extra"]
for k in e:
    v = e[k]
    print("extra." + k, (json.dumps(v)[:400]))
print("host_boundary.streamed", d["host_boundary"]["streamed"])
PY
echo "== done ($(date +%T))"
