#!/bin/bash
# Where the wave cycles of the fused kernels go: SQ counters (8 slots per pass), per kernel.
# PMC passes carry --kernel-trace only (no runtime/HIP trace domains).  Runs on the GPU box.
TAG=${1:-sq}
cd $(dirname $0)/..
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
python tools/kernel_times.py venice1778 > /dev/null 2>&1   # fills the /tmp problem cache
cd /tmp && export TMPDIR=/tmp
PASS1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
PASS2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES"
PASS3="SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
i=0
for P in "$PASS1" "$PASS2" "$PASS3"; do
  i=$((i+1)); rm -rf /tmp/pmc_sq$i
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmc_sq$i -o pmc -- python $REPO/tools/kernel_times.py venice1778 > /dev/null 2> $OUT/pmc_sq${i}_$TAG.err
  F=$(find /tmp/pmc_sq$i -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python - "$F" > $OUT/pmc_sq${i}_$TAG.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
    if "bal_" not in k: continue
    print(k[:140])
    for c, v in sorted(cs.items()):
        v2 = sorted(v)
        print(f"    {c:28s} n={len(v):4d} median={v2[len(v2)//2]:16.1f}")
PY
done
cat $OUT/pmc_sq1_$TAG.txt | head -60
