#!/bin/bash
# Copies what tools/gpu_r06_final.sh <TAG> left in gpurun_out/ (…_<TAG>.<ext>) into profiles/ as <TAG>_….<ext> and regenerates
# profiles/pmc_traffic.json from the PMC summaries.  usage: tools/collect_final.sh [TAG]   (default r06_final)
TAG=${1:-r06_final}
REPO=$(cd $(dirname $0)/.. && pwd); cd $REPO
for F in gpurun_out/*_$TAG.*; do
  B=$(basename $F); EXT=${B##*.}; STEM=${B%_$TAG.*}
  case $EXT in err) continue;; esac          # (the commands' stderr: kept in gpurun_out/ only)
  case $STEM in trace_*|rocprof_*) continue;; esac
  D=$STEM; [ $STEM = long_cg ] && D=long_cg_30_iterations_venice; cp $F profiles/${TAG}_$D.$EXT
done
python tools/make_pmc_traffic.py $TAG
ls profiles/${TAG}_* | wc -l
