#!/bin/bash
# Round-4 GPU-box session (run through gpurun).  usage: tools/gpu_r04.sh TAG stage [stage ...]
# new stages: bench_self2 / bench_self8 (bench.py --gpus N typed WITHOUT a launcher: it starts its own ranks; on a one-GPU box the ranks share
# device 0 = validation mode; --oracle-check compares the assembled step with the oracle at full size), every other stage is tools/gpu_r03.sh's.
TAG=$1; shift
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
for STAGE in "$@"; do
  case $STAGE in
    bench_self2|bench_self8)
      echo "===== $STAGE ($(date +%T))"
      N=${STAGE#bench_self}
      for WL in ${WLS:-ladybug1723}; do for SV in ${SVS:-iterative_schur cgnr}; do
        echo "--- $WL $SV --gpus $N (self-launched)"
        timeout 1200 python bench.py --gpus $N --workload $WL --solver $SV --steps 3 --warmup 1 --no-cpu-baseline --oracle-check 1 2> $OUT/bench_self${N}_${WL}_${SV}_$TAG.err | tee $OUT/bench_self${N}_${WL}_${SV}_$TAG.json | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print({'n_gpus': d['n_gpus'], 'value': d['value'], 'parallelism': d['config']['parallelism'][:90], 'oracle_check': d.get('oracle_check')})"
        tail -2 $OUT/bench_self${N}_${WL}_${SV}_$TAG.err
      done; done ;;
    pytest_files)   # FILES="tests/test_a.py tests/test_b.py"
      echo "===== $STAGE ($(date +%T))"
      timeout 2400 python -m pytest $FILES -m gpu -q --timeout 900 2>&1 | tail -300 | tee $OUT/pytest_files_$TAG.log | tail -30 ;;
    *) bash tools/gpu_r03.sh $TAG $STAGE ;;
  esac
done
echo "===== r04 done ($(date +%T))"
