#!/bin/bash
# kernel trace of a few replayed bench steps -> idle-gap / overlap analysis (tools/trace_gaps.py)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$REPO/gpurun_out/trace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --steps 3 --warmup 2 --no-roofline --no-cpu-baseline > $OUT/run.log 2>&1
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python $REPO/tools/trace_gaps.py "$f" > $OUT/gaps.txt 2>&1
cat $OUT/gaps.txt
rm -rf $OUT/prof
