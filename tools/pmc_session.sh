#!/bin/bash
# HBM traffic of the dominant kernels from the PMC counters (separate passes for
# FETCH_SIZE and WRITE_SIZE: TCC has 4 slots, FETCH_SIZE takes 3 and WRITE_SIZE 2), plus a
# kernel-trace --stats pass of bench.py.  Runs on the GPU box via gpurun.
TAG=${1:-r01}
cd $(dirname $0)/..
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
python tools/kernel_times.py venice1778 > $OUT/ktimes_$TAG.json 2>/dev/null   # also fills the /tmp cache
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -o pmc -- python $REPO/tools/kernel_times.py venice1778 > /dev/null 2> $OUT/pmc_$C\_$TAG.err
  F=$(find /tmp/pmc_$C -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python - "$F" $C > $OUT/pmc_$C\_$TAG.txt <<'PY'
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
done
for SOLVER in cgnr iterative_schur; do
  rm -rf /tmp/prof_$SOLVER
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$SOLVER -o $SOLVER -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --both-solvers 0 --solver $SOLVER > $OUT/rocprof_bench_$SOLVER\_$TAG.json 2> $OUT/rocprof_$SOLVER\_$TAG.err
  F=$(find /tmp/prof_$SOLVER -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp $F $OUT/kernel_stats_${SOLVER}_$TAG.csv
done
head -5 $OUT/pmc_FETCH_SIZE_$TAG.txt | cut -c1-250
