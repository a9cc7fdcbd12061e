cd /root/repo; REPO=$(pwd); OUT=$REPO/gpurun_out; TAG=${1:-r01f}
cd /tmp && export TMPDIR=/tmp
for SOLVER in cgnr iterative_schur; do
  rm -rf /tmp/prof_$SOLVER
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$SOLVER -o $SOLVER -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --both-solvers 0 --minimizer-iterations 0 --solver $SOLVER > $OUT/rocprof_bench_$SOLVER\_$TAG.json 2> $OUT/rocprof_$SOLVER\_$TAG.err
  F=$(find /tmp/prof_$SOLVER -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp $F $OUT/kernel_stats_${SOLVER}_$TAG.csv && head -4 $F | cut -c1-200
  python -c "
import json; d=json.load(open('$OUT/rocprof_bench_${SOLVER}_$TAG.json')); print('bench under rocprof:', d['value'], d['roofline']['avg_launch_ms'])"
done
