#!/bin/bash
# where a trust-region iteration's time goes: tools/minimize_trace.py plain and under rocprofv3 --kernel-trace --stats.  usage: TAG
TAG=$1
REPO=$(cd $(dirname $0)/.. && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
cd /tmp && export TMPDIR=/tmp
timeout 600 python $REPO/tools/minimize_trace.py > $OUT/minimize_trace_$TAG.jsonl 2> $OUT/minimize_trace_$TAG.err; tail -3 $OUT/minimize_trace_$TAG.jsonl
rm -rf /tmp/prof_min
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_min -o min -- python $REPO/tools/minimize_trace.py --repeat 1 > $OUT/minimize_trace_rocprof_$TAG.jsonl 2> $OUT/minimize_trace_rocprof_$TAG.err
F=$(find /tmp/prof_min -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp $F $OUT/kernel_stats_minimize_venice_$TAG.csv && head -30 $F | cut -c1-200
