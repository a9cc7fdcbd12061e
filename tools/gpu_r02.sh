#!/bin/bash
# Round-2 GPU-box session (run through gpurun).  usage: tools/gpu_r02.sh TAG stage [stage ...]
# stages: probe boxinfo pytest pytest_all smoke bench bench_cgnr bench10m bench10m_fp32 bench1m bench_n2 longcg small ktimes ktimes1m pmc p2plat gather ab_mo ab_z rocprof rocprof10m rocprof1m rocprof_small
TAG=$1; shift
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
for STAGE in "$@"; do
  echo "===== $STAGE ($(date +%T))"
  case $STAGE in
    probe) bash tools/probe.sh | tee $OUT/probe_$TAG.txt ;;
    pytest) timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -x 2>&1 | tail -60 | tee $OUT/pytest_gpu_$TAG.log | tail -15 ;;
    pytest_all) timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -80 | tee $OUT/pytest_gpu_$TAG.log | tail -25 ;;
    smoke) timeout 300 python -c "import __graft_entry__ as e; e.smoke()" 2>&1 | tail -5 | tee $OUT/smoke_$TAG.log ;;
    bench) timeout 900 python bench.py --gpus 1 2> $OUT/bench_$TAG.err | tee $OUT/bench_$TAG.json | cut -c1-1800; tail -3 $OUT/bench_$TAG.err ;;
    bench_cgnr) timeout 900 python bench.py --gpus 1 --solver cgnr --both-solvers 0 --minimizer-iterations 0 --host-boundary-steps 0 2> $OUT/bench_cgnr_$TAG.err | tee $OUT/bench_cgnr_$TAG.json | cut -c1-1200; tail -3 $OUT/bench_cgnr_$TAG.err ;;
    bench10m) timeout 1500 python bench.py --gpus 1 --workload synthetic10M --steps 10 --cpu-seconds 30 2> $OUT/bench_synthetic10M_fp64_$TAG.err | tee $OUT/bench_synthetic10M_fp64_$TAG.json | cut -c1-1800; tail -3 $OUT/bench_synthetic10M_fp64_$TAG.err ;;
    bench10m_fp32) timeout 1500 python bench.py --gpus 1 --workload synthetic10M --storage fp32 --steps 10 --no-cpu-baseline 2> $OUT/bench_synthetic10M_fp32_$TAG.err | tee $OUT/bench_synthetic10M_fp32_$TAG.json | cut -c1-1500; tail -3 $OUT/bench_synthetic10M_fp32_$TAG.err ;;
    bench1m) timeout 900 python bench.py --gpus 1 --workload synthetic1M --steps 20 --no-cpu-baseline 2> $OUT/bench_synthetic1M_$TAG.err | tee $OUT/bench_synthetic1M_$TAG.json | cut -c1-1500; tail -3 $OUT/bench_synthetic1M_$TAG.err ;;
    longcg)
      for F in 0 1; do CERES_HIP_CG_FUSED=$F timeout 600 python tools/gpu_long_cg.py 30 2>&1 | tail -1 | tee -a $OUT/longcg_$TAG.jsonl; done ;;
    small)
      for WL in dubrovnik16 ladybug1723; do for F in 0 1; do
        echo -n "$WL fused=$F " | tee -a $OUT/small_$TAG.txt
        CERES_HIP_CG_FUSED=$F timeout 600 python bench.py --workload $WL --steps 200 --warmup 20 --no-cpu-baseline --minimizer-iterations 0 --host-boundary-steps 0 2>/dev/null \
          | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k: d[k] for k in ('ms_per_step','value')} | {'cg_its': d['config']['cg_iterations_per_step'], 'cgnr': d['extra'].get('cgnr',{}).get('ms_per_step'), 'sx_ms': d['roofline']['avg_launch_ms'], 'phases': d['extra']['solve_phases_ms']}))" | tee -a $OUT/small_$TAG.txt
      done; done ;;
    ab_mo)   # M_o written camera-major by kInit vs gathered through cam_slot (Venice shape)
      for V in 0 1; do echo -n "MO_CAM=$V " | tee -a $OUT/ab_mo_$TAG.jsonl; CERES_HIP_MO_CAM=$V timeout 600 python tools/kernel_times.py venice1778 2>$OUT/ab_mo_$TAG.err | tail -1 | tee -a $OUT/ab_mo_$TAG.jsonl; done ;;
    ab_z)    # per-observation F^T z written camera-major vs gathered (50 k cameras, 3 M observations)
      for V in 0 1; do echo -n "Z_CAM=$V " | tee -a $OUT/ab_z_$TAG.jsonl; CERES_HIP_Z_CAM=$V CERES_HIP_MO_CAM=$V timeout 600 python tools/kernel_times.py synthetic1M 2>/dev/null | tail -1 | tee -a $OUT/ab_z_$TAG.jsonl; done ;;
    ktimes)  # per-operator times, Venice shape
      timeout 600 python tools/kernel_times.py venice1778 2>$OUT/ktimes_$TAG.err | tail -1 | tee -a $OUT/ktimes_$TAG.jsonl; tail -3 $OUT/ktimes_$TAG.err ;;
    ktimes1m)
      timeout 600 python tools/kernel_times.py synthetic1M 2>$OUT/ktimes1m_$TAG.err | tail -1 | tee -a $OUT/ktimes1m_$TAG.jsonl; tail -3 $OUT/ktimes1m_$TAG.err ;;
    gather)  # what a camera-major gather of 144-byte cells costs + what FETCH_SIZE reports for it (known useful bytes)
      timeout 300 ./tools/probes/gather_probe | tee $OUT/gather_probe_$TAG.txt
      cd /tmp && export TMPDIR=/tmp
      for T in order0_w16 order1_w16 order2_w16 order2_w8; do
        rm -rf /tmp/pmc_g
        timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_g -o g -- $REPO/tools/probes/gather_probe $T > /dev/null 2>&1
        F=$(find /tmp/pmc_g -name "*counter_collection.csv" | head -1)
        [ -n "$F" ] && python3 - "$F" $T <<'PY' | tee -a $OUT/gather_probe_$TAG.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r.get("Counter_Name") == "FETCH_SIZE" and "gather_kernel" in r.get("Kernel_Name", "")]
vals = [float(r["Counter_Value"]) for r in rows]
if vals:
    v = sum(vals) / len(vals)
    print(f"{sys.argv[2]}: FETCH_SIZE per launch = {v:.0f} (KiB => {v * 1024 / 1e6:.1f} MB as reported, {2 * v * 1024 / 1e6:.1f} MB doubled); useful = {5001946 * 144 / 1e6:.1f} MB; dispatches {len(vals)}")
PY
      done
      cd $REPO ;;
    pmc)   # HBM traffic counters, one pass per counter set (TCC slots), kernel-trace only: WLS="venice1778 synthetic1M" COUNTERS="FETCH_SIZE WRITE_SIZE"
      for WL in ${WLS:-venice1778}; do
        timeout 900 python tools/kernel_times.py $WL > /dev/null 2>&1   # fills the /tmp cache
        cd /tmp && export TMPDIR=/tmp
        for C in ${COUNTERS:-FETCH_SIZE WRITE_SIZE}; do
          CN=$(echo $C | tr ' ' '+')
          rm -rf /tmp/pmc_run
          timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_run -o pmc -- python $REPO/tools/kernel_times.py $WL > /dev/null 2> $OUT/pmc_${CN}_${WL}_$TAG.err
          F=$(find /tmp/pmc_run -name "*counter_collection.csv" | head -1)
          [ -n "$F" ] && python3 - "$F" > $OUT/pmc_${CN}_${WL}_$TAG.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Counter_Name"]][r["Kernel_Name"]].append(float(r["Counter_Value"]))
for cname, ks in agg.items():
    print("counter:", cname)
    for k, v in sorted(ks.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        print(f"{k[:110]:110s} n={len(v):5d} mean={sum(v)/len(v):14.1f} median={v2[len(v2)//2]:14.1f} max={v2[-1]:14.1f}")
PY
          head -12 $OUT/pmc_${CN}_${WL}_$TAG.txt | cut -c1-200
        done
        cd $REPO
      done ;;
    boxinfo)   # what kind of box is this (write-carrying kernels vary by box)
      { rocm-smi --showcomputepartition --showmemorypartition --showclocks --showperflevel --showpower --showtemp --showuse 2>&1 | grep -v "^$" | head -60; } | tee $OUT/boxinfo_$TAG.txt ;;
    p2plat) timeout 300 python tools/p2p_latency.py 2>$OUT/p2p_latency_$TAG.err | tail -1 | tee $OUT/p2p_latency_$TAG.json; tail -2 $OUT/p2p_latency_$TAG.err ;;
    bench_n2)  # the N > 1 code path of bench.py and of the library with two ranks on ONE GPU (validation only: timings meaningless)
      for WL in ${WLS:-ladybug1723 synthetic1M}; do for SV in iterative_schur cgnr; do
        echo "--- $WL $SV"
        CERES_HIP_BENCH_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload $WL --solver $SV --steps 5 --warmup 2 --no-cpu-baseline 2> $OUT/bench_n2_${WL}_${SV}_$TAG.err | tee $OUT/bench_n2_${WL}_${SV}_$TAG.json | cut -c1-900; tail -2 $OUT/bench_n2_${WL}_${SV}_$TAG.err
      done; done ;;
    rocprof1m)
      cd /tmp && export TMPDIR=/tmp
      for MIB in ${CHUNKS:-64}; do
        rm -rf /tmp/prof_1m
        CERES_HIP_Z_CHUNK_MIB=$MIB timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_1m -o s1m -- python $REPO/bench.py --workload synthetic1M --steps 10 --warmup 2 --no-cpu-baseline --both-solvers 0 > $OUT/rocprof_bench_synthetic1M_${MIB}_$TAG.json 2> $OUT/rocprof_synthetic1M_$TAG.err
        F=$(find /tmp/prof_1m -name "*kernel_stats.csv" | head -1)
        echo "--- chunk MiB $MIB"; python -c "import json;d=json.loads(open('$OUT/rocprof_bench_synthetic1M_${MIB}_$TAG.json').read());print('sx_ms',d['roofline']['avg_launch_ms'],'step',d['ms_per_step'])"
        [ -n "$F" ] && cp $F $OUT/kernel_stats_synthetic1M_${MIB}_$TAG.csv && head -8 $F | cut -c1-160
      done
      cd $REPO ;;
    rocprof_small)
      cd /tmp && export TMPDIR=/tmp
      for WL in ladybug1723 dubrovnik16; do
        rm -rf /tmp/prof_$WL
        timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$WL -o $WL -- python $REPO/bench.py --workload $WL --steps 200 --warmup 10 --no-cpu-baseline --both-solvers 0 --minimizer-iterations 0 --host-boundary-steps 0 > $OUT/rocprof_bench_${WL}_$TAG.json 2> $OUT/rocprof_${WL}_$TAG.err
        F=$(find /tmp/prof_$WL -name "*kernel_stats.csv" | head -1)
        [ -n "$F" ] && cp $F $OUT/kernel_stats_${WL}_$TAG.csv && head -22 $F | cut -c1-200
      done
      cd $REPO ;;
    rocprof)
      cd /tmp && export TMPDIR=/tmp
      for SOLVER in iterative_schur cgnr; do
        rm -rf /tmp/prof_$SOLVER
        timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$SOLVER -o $SOLVER -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --both-solvers 0 --minimizer-iterations 0 --host-boundary-steps 0 --solver $SOLVER > $OUT/rocprof_bench_${SOLVER}_$TAG.json 2> $OUT/rocprof_${SOLVER}_$TAG.err
        F=$(find /tmp/prof_$SOLVER -name "*kernel_stats.csv" | head -1)
        [ -n "$F" ] && cp $F $OUT/kernel_stats_${SOLVER}_venice_$TAG.csv && head -14 $F | cut -c1-220
      done
      cd $REPO ;;
    rocprof10m)
      cd /tmp && export TMPDIR=/tmp
      rm -rf /tmp/prof_10m
      timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_10m -o s10m -- python $REPO/bench.py --workload synthetic10M --steps 5 --warmup 2 --no-cpu-baseline --both-solvers 1 > $OUT/rocprof_bench_synthetic10M_$TAG.json 2> $OUT/rocprof_synthetic10M_$TAG.err
      F=$(find /tmp/prof_10m -name "*kernel_stats.csv" | head -1)
      [ -n "$F" ] && cp $F $OUT/kernel_stats_synthetic10M_$TAG.csv && head -14 $F | cut -c1-220
      cd $REPO ;;
    *) echo "unknown stage $STAGE" ;;
  esac
done
echo "===== done ($(date +%T))"
