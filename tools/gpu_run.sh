#!/bin/bash
# GPU pass: parity tests, smoke, bench (graph + eager), rocprof kernel stats of the bench command.
# usage: tools/gpu_run.sh <tag> [skip-tests]
TAG=${1:-run}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
nproc > $OUT/box.txt; rocminfo | grep -E "gfx9" | head -2 >> $OUT/box.txt
if [ "$2" != "skip-tests" ]; then
  timeout 900 python -m pytest tests -m gpu -q --maxfail=30 --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> $OUT/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
  echo "smoke exit $?" >> $OUT/smoke.log
fi
timeout 900 python bench.py > $OUT/bench.log 2> $OUT/bench.err
echo "bench exit $?" >> $OUT/bench.log
timeout 300 python bench.py --no-graph --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > $OUT/bench_eager.log 2> $OUT/bench_eager.err
echo "bench exit $?" >> $OUT/bench_eager.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > $REPO/$OUT/prof.log 2>&1
cd $REPO
find $OUT/prof -name "*kernel_stats*" >> $OUT/prof.log
# keep only the stats csv + a trimmed kernel trace (the full trace can be tens of MB)
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
tail -3 $OUT/pytest_gpu.log 2>/dev/null; tail -2 $OUT/smoke.log 2>/dev/null; tail -2 $OUT/bench.log; tail -2 $OUT/bench_eager.log
