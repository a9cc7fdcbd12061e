#!/bin/bash
# launch-/latency-bound problem sizes: LM step rate and kernel time per step (rocprofv3 kernel stats)
cd $(dirname $0)/..
REPO=$(pwd)
for W in dubrovnik16 ladybug1723; do
  timeout 300 python bench.py --workload $W --no-cpu-baseline --both-solvers 0 --minimizer-iterations 0 --steps 200 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$W', d['value'], 'steps/s', d['ms_per_step'], 'ms', d['config']['cg_iterations_per_step'], d['extra']['solve_phases_ms'])"
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_small
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_small -o small -- python $REPO/bench.py --workload ladybug1723 --no-cpu-baseline --both-solvers 0 --minimizer-iterations 0 --steps 200 --warmup 10 --kernel-iters 1 > /dev/null 2>&1
F=$(find /tmp/prof_small -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print("total kernel ms", tot / 1e6, "calls", calls, "-> per step (210 steps):", tot / 1e6 / 210, "ms,", calls / 210, "launches")
for r in rows[:14]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
