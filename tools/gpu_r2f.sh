#!/bin/bash
# round-2 closing pass (second): full GPU suite, smoke, bench lines (configs 3 and 4) with roofline (+ cpu_baseline for 3),
# rocprofv3 kernel stats of the bench command, per-shape table of one eager step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD; OUT=gpurun_out/r2f; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
TG_DUMP_SHAPES=$REPO/$OUT/shapes_c3.json timeout 400 python bench.py > $OUT/bench_c3.log 2> $OUT/bench_c3.err; echo "exit $?" >> $OUT/bench_c3.log
timeout 300 python bench.py --config 4 --no-cpu-baseline > $OUT/bench_c4.log 2> $OUT/bench_c4.err; echo "exit $?" >> $OUT/bench_c4.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_c3 -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > $REPO/$OUT/prof_c3.log 2>&1
cd $REPO; find $OUT/prof_c3 -name "*kernel_trace.csv" -delete
tail -2 $OUT/pytest_gpu.log; tail -1 $OUT/smoke.log; head -c 250 $OUT/bench_c3.log; echo; head -c 250 $OUT/bench_c4.log; echo
