#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, smoke, bench, rocprofv3 kernel stats.
# Usage: tools/gpu_session.sh [tag]
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out
[ -z "$GRAFT_REPO_ROOT" ] && OUT=$(pwd)/gpurun_out
mkdir -p $OUT
cd $(dirname $0)/..
REPO=$(pwd)
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -40 | tee $OUT/pytest_gpu_$TAG.log | grep -E "passed|failed"
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as e; e.smoke()" 2>&1 | tail -5 | tee $OUT/smoke_$TAG.log
echo "== bench (default: iterative_schur)"; timeout 900 python bench.py --gpus 1 2> $OUT/bench_$TAG.err | tee $OUT/bench_$TAG.json | cut -c1-1500; tail -3 $OUT/bench_$TAG.err
echo "== bench (cgnr)"; timeout 900 python bench.py --gpus 1 --solver cgnr --both-solvers 0 2> $OUT/bench_cgnr_$TAG.err | tee $OUT/bench_cgnr_$TAG.json | cut -c1-1200
echo "== rocprofv3 kernel stats"
cd /tmp && export TMPDIR=/tmp
for SOLVER in cgnr iterative_schur; do
  rm -rf /tmp/prof_$SOLVER
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$SOLVER -o $SOLVER -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --both-solvers 0 --minimizer-iterations 0 --solver $SOLVER > $OUT/rocprof_bench_$SOLVER\_$TAG.json 2> $OUT/rocprof_$SOLVER\_$TAG.err
  F=$(find /tmp/prof_$SOLVER -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp $F $OUT/kernel_stats_${SOLVER}_$TAG.csv && head -12 $F | cut -c1-200
done
