#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command (the Venice-shaped step, both solvers; the extras of the default line switched
# off so that the trace holds the step's kernels only) + the JSON line of that same run.  usage: tools/gpu_rocprof_quick.sh TAG
TAG=$1
REPO=$(cd $(dirname $0)/.. && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
cd /tmp && export TMPDIR=/tmp
for SOLVER in iterative_schur cgnr; do
  rm -rf /tmp/prof_$SOLVER
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$SOLVER -o $SOLVER -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --both-solvers 0 --minimizer-iterations 0 --host-boundary-steps 0 --scene-step-steps 0 --conditioned-steps 0 --extra-synthetic10m 0 --extra-real-graph 0 --extra-other-shapes 0 --extra-dense-cholesky 0 --solver $SOLVER > $OUT/rocprof_bench_${SOLVER}_$TAG.json 2> $OUT/rocprof_${SOLVER}_$TAG.err
  F=$(find /tmp/prof_$SOLVER -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp $F $OUT/kernel_stats_${SOLVER}_venice_$TAG.csv && head -8 $F | cut -c1-160
  python -c "
import json; d=json.loads(open('$OUT/rocprof_bench_${SOLVER}_$TAG.json').read().strip().splitlines()[-1]); print('$SOLVER line:', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
