#!/bin/bash
TAG=${1:-r06k}
REPO=$(cd $(dirname $0)/.. && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
exec < /dev/null
echo "== bench banded ($(date +%T))"
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --minimizer-iterations 0 --extra-synthetic10m 0 --extra-other-shapes 0 --extra-dense-cholesky 0 --extra-real-graph 0 --host-boundary-steps 0 --shard-ceiling 0 --extra-configs 0 --both-solvers 0 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; tail -2 $OUT/bench_$TAG.err
python - <<PY
import json
d = json.loads([l for l in open("$OUT/bench_$TAG.json").read().splitlines() if l.startswith("{")][-1])
print(json.dumps(d["extra"].get("banded50k")))
PY
echo "== synthetic10M kernel times ($(date +%T))"
timeout 900 python tools/kernel_times.py synthetic10M 2>&1 | tail -1 | tee $OUT/kernel_times_synthetic10M_$TAG.json
cd /tmp && export TMPDIR=/tmp
echo "== pmc ($(date +%T))"
for WL in synthetic10M; do for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -o pmc -- python $REPO/tools/kernel_times.py $WL > /dev/null 2> $OUT/pmc_${C}_${WL}_$TAG.err
  F=$(timeout 20 find /tmp/pmc_$C -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ]; then timeout 120 python - "$F" $C > $OUT/pmc_${C}_${WL}_$TAG.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if r.get("Counter_Name") == sys.argv[2]:
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
print("columns:", list(rows[0].keys()) if rows else None)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{k[:110]:110s} n={len(v):5d} mean={sum(v)/len(v):14.1f} median={v2[len(v2)//2]:14.1f} max={v2[-1]:14.1f}")
PY
  head -8 $OUT/pmc_${C}_${WL}_$TAG.txt | cut -c1-200; else echo "no counter csv for $C $WL"; tail -3 $OUT/pmc_${C}_${WL}_$TAG.err; fi
done; done
echo "== done ($(date +%T))"
