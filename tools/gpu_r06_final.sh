#!/bin/bash
# Round-6 end-state evidence (one gpurun call; every command under its own timeout, nothing reads stdin):
#   smoke, GPU tests, the default bench line, rocprofv3 --kernel-trace --stats of the bench command for both solvers, PMC FETCH_SIZE / WRITE_SIZE
#   passes (Venice, synthetic1M, synthetic10M, banded50k), fixed-length CG solves, the shard ceilings and the kernel timeline of one rank of eight.
TAG=${1:-r06_final}
REPO=$(cd $(dirname $0)/.. && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
exec < /dev/null
echo "== smoke ($(date +%T))"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$TAG.log 2>&1; tail -3 $OUT/smoke_$TAG.log | cut -c1-300
echo "== gpu tests ($(date +%T))"
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -8 | tee $OUT/pytest_gpu_$TAG.log
echo "== bench ($(date +%T))"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default_iterative_schur_$TAG.json 2> $OUT/bench_default_$TAG.err
timeout 30 python - <<PY
import json
try:
    d = json.loads([l for l in open("$OUT/bench_default_iterative_schur_$TAG.json").read().splitlines() if l.startswith("{")][-1])
    print({k: d[k] for k in ("value", "ms_per_step")}, "sx", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "jtjx", d["roofline_jtjx"]["frac"], "step_roofline", d["step_roofline"]["frac"], "oracle", (d.get("oracle_check") or {}).get("step_rel_diff_vs_oracle"))
    print("shard", [(c["ranks"], c["ms_per_step"], c["efficiency_ceiling"]) for c in d["extra"]["shard_ceiling"]["cases"]], "streamed", d["host_boundary"]["streamed"]["ms_after_last_push"])
except Exception as ex:
    print("bench line unreadable:", ex)
PY
cd /tmp && export TMPDIR=/tmp
for SOLVER in iterative_schur cgnr; do
  echo "== rocprof $SOLVER ($(date +%T))"
  rm -rf /tmp/prof_$SOLVER
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$SOLVER -o $SOLVER -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --both-solvers 0 --minimizer-iterations 0 --host-boundary-steps 0 --extra-synthetic10m 0 --extra-other-shapes 0 --extra-dense-cholesky 0 --extra-real-graph 0 --extra-banded 0 --extra-configs 0 --shard-ceiling 0 --solver $SOLVER > $OUT/bench_under_rocprof_${SOLVER}_$TAG.json 2> $OUT/rocprof_${SOLVER}_$TAG.err
  F=$(timeout 20 find /tmp/prof_$SOLVER -name "*kernel_stats.csv" | head -1)
  if [ -n "$F" ]; then cp "$F" $OUT/kernel_stats_${SOLVER}_venice_$TAG.csv; head -8 "$F" | cut -c1-170; else echo "no kernel_stats.csv"; tail -3 $OUT/rocprof_${SOLVER}_$TAG.err; fi
done
echo "== pmc ($(date +%T))"
for WL in venice1778 synthetic1M synthetic10M banded50k; do for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -o pmc -- python $REPO/tools/kernel_times.py $WL > /dev/null 2> $OUT/pmc_${C}_${WL}_$TAG.err
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
  head -3 $OUT/pmc_${C}_${WL}_$TAG.txt | cut -c1-200; else echo "no counter csv for $C $WL"; fi
done; done
echo "== kernel times ($(date +%T))"
cd $REPO
for WL in venice1778 synthetic10M banded50k; do timeout 600 python tools/kernel_times.py $WL 2>/dev/null | tail -1 > $OUT/kernel_times_${WL}_$TAG.json; cut -c1-400 $OUT/kernel_times_${WL}_$TAG.json; done
echo "== long cg ($(date +%T))"
timeout 300 python tools/gpu_long_cg.py 30 2>/dev/null | tail -1 | tee $OUT/long_cg_$TAG.jsonl
echo "== shard ceilings ($(date +%T))"
for S in iterative_schur cgnr; do timeout 600 python tools/shard_step.py venice1778 1,2,4,8 $S 20 2>&1 | grep "^{" | tee -a $OUT/shard_step_$TAG.jsonl | cut -c1-160; done
timeout 1200 python tools/shard_step.py synthetic10M 1,8 iterative_schur 5 2>&1 | grep "^{" | tee -a $OUT/shard_step_$TAG.jsonl | cut -c1-160
cd /tmp
for CASE in "venice1778 8 iterative_schur" "venice1778 1 iterative_schur" "synthetic10M 8 iterative_schur"; do
  set -- $CASE
  NAME=$1_n$2_$3
  rm -rf /tmp/trace_$NAME
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$NAME -o t -- python $REPO/tools/shard_step.py $1 $2 $3 4 2 > /dev/null 2> $OUT/trace_${NAME}_$TAG.err
  F=$(timeout 20 find /tmp/trace_$NAME -name "*kernel_trace.csv" | head -1)
  if [ -n "$F" ]; then python $REPO/tools/kernel_timeline.py "$F" 90 > $OUT/timeline_${NAME}_$TAG.txt; echo "timeline $NAME: $(wc -l < $OUT/timeline_${NAME}_$TAG.txt) lines"; fi
done
echo "== done ($(date +%T))"
