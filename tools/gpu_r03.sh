#!/bin/bash
# Round-3 GPU-box session (run through gpurun).  usage: tools/gpu_r03.sh TAG stage [stage ...]
# new stages: dense realgraph leftover generic small_ab; bench_n8 (bench.py's N = 8 code path, eight ranks on ONE GPU: validation only), bench_n4, ab (A/B libraries:
# VARIANTS="base xent ..." REPS=3 WL=venice1778), longcg_ab, pytest_multirank; every other stage is tools/gpu_r02.sh's.
TAG=$1; shift
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
for STAGE in "$@"; do
  case $STAGE in
    bench_n8|bench_n4|bench_n3)
      echo "===== $STAGE ($(date +%T))"
      N=${STAGE#bench_n}
      for WL in ${WLS:-ladybug1723}; do for SV in iterative_schur cgnr; do
        echo "--- $WL $SV world $N"
        CERES_HIP_BENCH_ONE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $N --workload $WL --solver $SV --steps 5 --warmup 2 --no-cpu-baseline 2> $OUT/bench_n${N}_${WL}_${SV}_$TAG.err | tee $OUT/bench_n${N}_${WL}_${SV}_$TAG.json | cut -c1-700; tail -2 $OUT/bench_n${N}_${WL}_${SV}_$TAG.err
      done; done ;;
    ab)   # interleaved processes, one library per process: base = the product
      echo "===== $STAGE ($(date +%T))"
      WL=${WL:-venice1778}
      timeout 900 python tools/kernel_times.py $WL > /dev/null 2>&1   # fills the /tmp cache
      for R in $(seq 1 ${REPS:-3}); do for V in ${VARIANTS:-base}; do
        LIB=""; [ "$V" != "base" ] && LIB=$REPO/ceres-solver_amd/csrc/variants/libceres_hip_$V.so
        echo -n "$V " | tee -a $OUT/ab_$TAG.txt
        CERES_HIP_LIBRARY=$LIB timeout 300 python tools/kernel_times.py $WL 2>/dev/null | tail -1 | tee -a $OUT/ab_$TAG.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('jtjx_ms','jtjx_frac','sx_ms','sx_frac','schur_init_ms','schur_jacobi_ms','back_substitute_ms','cgnr_setup_ms')})"
      done; done ;;
    longcg_ab)
      echo "===== $STAGE ($(date +%T))"
      for R in $(seq 1 ${REPS:-2}); do for V in ${VARIANTS:-base}; do
        LIB=""; [ "$V" != "base" ] && LIB=$REPO/ceres-solver_amd/csrc/variants/libceres_hip_$V.so
        echo -n "$V " | tee -a $OUT/longcg_ab_$TAG.txt
        CERES_HIP_LIBRARY=$LIB timeout 600 python tools/gpu_long_cg.py 30 2>&1 | tail -1 | tee -a $OUT/longcg_ab_$TAG.txt
      done; done ;;
    pytest_multirank)
      echo "===== $STAGE ($(date +%T))"
      timeout 1800 python -m pytest tests/test_gpu_multirank.py -m gpu -q --timeout 900 2>&1 | tail -40 | tee $OUT/pytest_multirank_$TAG.log | tail -15 ;;
    dense)     # DENSE_SCHUR's factorisation: MFMA probe + times at 228 / 456 / 910 cameras (+ kernel stats at n = 8190)
      echo "===== $STAGE ($(date +%T))"
      timeout 120 ./tools/probes/mfma_f64_probe | tee $OUT/mfma_f64_probe_$TAG.txt
      timeout 300 python tools/dense_cholesky_times.py 2052 4104 8190 2>&1 | grep "^{" | tee $OUT/dense_cholesky_$TAG.jsonl
      cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_dc
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dc -o dc -- python $REPO/tools/dense_cholesky_times.py 8190 > /dev/null 2>&1
      F=$(find /tmp/prof_dc -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/kernel_stats_dense_cholesky_8190_$TAG.csv && head -8 $F | cut -c1-200
      cd $REPO ;;
    realgraph) # S.x / JtJx on the replicated libmv visibility graphs
      echo "===== $STAGE ($(date +%T))"
      timeout 600 python tools/real_graph_times.py 2>/dev/null | tee $OUT/real_graph_$TAG.jsonl ;;
    leftover)  # Venice shape + 1 % prior rows against the pure problem
      echo "===== $STAGE ($(date +%T))"
      timeout 600 python tools/leftover_rows_times.py 2>/dev/null | tee $OUT/leftover_rows_$TAG.json ;;
    generic)   # the generic path, timed once (Ladybug shape)
      echo "===== $STAGE ($(date +%T))"
      timeout 300 python tools/kernel_times.py ladybug1723 --force-generic 2>/dev/null | tail -1 | tee $OUT/ktimes_generic_ladybug_$TAG.json ;;
    small_ab)  # launch- / latency-bound shapes per library variant
      echo "===== $STAGE ($(date +%T))"
      bash tools/gpu_small_ab.sh 2>&1 | tee $OUT/small_ab_$TAG.txt ;;
    *) bash tools/gpu_r02.sh $TAG $STAGE ;;
  esac
done
echo "===== r03 done ($(date +%T))"
