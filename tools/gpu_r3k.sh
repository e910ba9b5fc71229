#!/bin/bash
# round-3 closing pass (repeatable): dispatch table re-recorded, FULL GPU suite, smoke, default bench line with roofline +
# cpu_baseline + per-shape table, rocprofv3 kernel statistics of configs 3 and 4, config-4 line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD; OUT=gpurun_out/${1:-r3k}; mkdir -p $OUT; export TMPDIR=/tmp
TG_RECORD_KERNELS=$PWD/$OUT/bench_dispatch_kernels.json timeout 900 python -m pytest tests/test_gpu_bench_shapes.py -q -m gpu --tb=short -p no:cacheprovider > $OUT/pytest_record.log 2>&1
cp $OUT/bench_dispatch_kernels.json tests/golden/bench_dispatch_kernels.json
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
TG_DUMP_SHAPES=$REPO/$OUT/shapes_c3.json timeout 400 python bench.py > $OUT/bench_c3.log 2> $OUT/bench_c3.err; echo "exit $?" >> $OUT/bench_c3.log
timeout 300 python bench.py --config 4 --no-cpu-baseline > $OUT/bench_c4.log 2> $OUT/bench_c4.err; echo "exit $?" >> $OUT/bench_c4.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_c3 -o bench -- python $REPO/bench.py --steps 4 --warmup 1 --no-roofline --no-cpu-baseline > $REPO/$OUT/prof_c3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_c4 -o bench -- python $REPO/bench.py --config 4 --steps 4 --warmup 1 --no-roofline --no-cpu-baseline > $REPO/$OUT/prof_c4.log 2>&1
cd $REPO; find $OUT -name "*kernel_trace.csv" -delete
tail -2 $OUT/pytest_record.log; tail -3 $OUT/pytest_gpu.log; tail -1 $OUT/smoke.log
for f in c3 c4; do echo -n "$f: "; head -c 260 $OUT/bench_$f.log | cut -c 80-260; echo; done
