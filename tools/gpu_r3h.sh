#!/bin/bash
# round-3 closing pass: full GPU suite, smoke, the default bench line (roofline + cpu_baseline + per-shape table),
# rocprofv3 kernel statistics of configs 3 and 4, the other BASELINE configs, the N > 1 schedule on one GPU with and
# without overlap, deterministic mode's cost, PMC counters of the filter-gradient kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD; OUT=gpurun_out/r3h; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
TG_DUMP_SHAPES=$REPO/$OUT/shapes_c3.json timeout 400 python bench.py > $OUT/bench_c3.log 2> $OUT/bench_c3.err; echo "exit $?" >> $OUT/bench_c3.log
timeout 300 python bench.py --config 4 --no-cpu-baseline > $OUT/bench_c4.log 2> $OUT/bench_c4.err; echo "exit $?" >> $OUT/bench_c4.log
for c in 0 1 2; do timeout 200 python bench.py --config $c --no-cpu-baseline --no-roofline > $OUT/bench_c$c.log 2> $OUT/bench_c$c.err; done
TG_DETERMINISTIC=1 timeout 200 python bench.py --no-cpu-baseline --no-roofline > $OUT/bench_c3_det.log 2> $OUT/bench_c3_det.err
timeout 200 python bench.py --no-cpu-baseline --no-roofline > $OUT/bench_c3_b.log 2> $OUT/bench_c3_b.err
timeout 200 python bench.py --reduce-always --overlap on --no-cpu-baseline --no-roofline > $OUT/bench_c3_reduce_on.log 2> $OUT/bench_c3_reduce_on.err
timeout 200 python bench.py --reduce-always --overlap off --no-cpu-baseline --no-roofline > $OUT/bench_c3_reduce_off.log 2> $OUT/bench_c3_reduce_off.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_c3 -o bench -- python $REPO/bench.py --steps 4 --warmup 1 --no-roofline --no-cpu-baseline > $REPO/$OUT/prof_c3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_c4 -o bench -- python $REPO/bench.py --config 4 --steps 4 --warmup 1 --no-roofline --no-cpu-baseline > $REPO/$OUT/prof_c4.log 2>&1
cd $REPO; find $OUT -name "*kernel_trace.csv" -delete
bash tools/pmc_kernels.sh "E128a wgrad 64;E64a wgrad 64;E32a wgrad 64;E16 wgrad 64;G32a wgrad 64;G64a wgrad 64;E64a fwd 64;E256a fwd 64;E256b fwd 48" > $OUT/pmc.log 2>&1
cp gpurun_out/pmc2/summary.json $OUT/pmc_summary.json
tail -3 $OUT/pytest_gpu.log; tail -1 $OUT/smoke.log
for f in c3 c4 c0 c1 c2 c3_det c3_b c3_reduce_on c3_reduce_off; do echo -n "$f: "; head -c 260 $OUT/bench_$f.log | cut -c 80-260; echo; done
tail -12 $OUT/pmc.log
