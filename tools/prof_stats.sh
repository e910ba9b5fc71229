#!/bin/bash
# rocprofv3 kernel stats of the (graph-mode) bench -> gpurun_out/<tag>/prof ; usage: tools/prof_stats.sh <tag>
TAG=${1:-prof}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > $REPO/$OUT/prof.log 2>&1
cd $REPO
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
tail -1 $OUT/prof.log
