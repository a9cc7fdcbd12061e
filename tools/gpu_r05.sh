#!/bin/bash
# Round-5 GPU-box session (run through gpurun).  usage: tools/gpu_r05.sh TAG stage [stage ...]
# stages: pytest_files (FILES=...), pytest_all, bench, shapes; anything else goes to tools/gpu_r04.sh
TAG=$1; shift
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
for STAGE in "$@"; do
  case $STAGE in
    pytest_files)
      echo "===== $STAGE ($(date +%T))"
      timeout 1500 python -m pytest $FILES -m gpu -q -x --timeout 900 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -60 | tee $OUT/pytest_files_$TAG.log | tail -30 ;;
    pytest_all)
      echo "===== $STAGE ($(date +%T))"
      timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -60 | tee $OUT/pytest_all_$TAG.log | tail -15 ;;
    bench)
      echo "===== $STAGE ($(date +%T))"
      timeout 600 python bench.py ${BENCH_ARGS} > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
      python - <<PY
import json
d = json.loads([l for l in open("$OUT/bench_$TAG.json").read().splitlines() if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step")}, "sx", d["roofline"]["frac"], "jtjx", d.get("roofline_jtjx", {}).get("frac"))
for c in d.get("extra", {}).get("other_shapes", {}).get("cases", []): print(c)
print(d.get("extra", {}).get("other_shapes", {}).get("error"))
PY
      ;;
    *) bash tools/gpu_r04.sh $TAG $STAGE ;;
  esac
done
echo "===== r05 done ($(date +%T))"
