#!/bin/bash
# round-2 measurement pass: PMC (traffic + MFMA utilisation) per conv family, bench lines + rocprof stats of configs 1-3
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD; OUT=gpurun_out/r2b; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/pmc_kernels.sh "E256a fwd 64;E256a dgrad 64;E256a wgrad 64;E256b fwd 48;E256b wgrad 64;E128a fwd 64;E128a wgrad 64;E128b fwd 48;E64a fwd 64;E64a wgrad 64;E64b fwd 48;E32a fwd 64;E32a wgrad 64;E32b fwd 48;E16 fwd 64;E16 wgrad 64;E8 fwd 64;G64a fwd 64;G32a fwd 64" > $OUT/pmc.log 2>&1
cp gpurun_out/pmc2/summary.json $OUT/pmc_summary.json
for c in 1 2; do
  timeout 600 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c$c.log 2> $OUT/bench_c$c.err; echo "exit $?" >> $OUT/bench_c$c.log
done
TG_DUMP_SHAPES=$REPO/$OUT/shapes_c3.json timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_c3.log 2> $OUT/bench_c3.err; echo "exit $?" >> $OUT/bench_c3.log
for c in 1 2 3; do
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_c$c -o bench -- python $REPO/bench.py --config $c --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > $REPO/$OUT/prof_c$c.log 2>&1
  cd $REPO; find $OUT/prof_c$c -name "*kernel_trace.csv" -delete
done
tail -5 $OUT/pmc.log; for c in 1 2 3; do head -c 300 $OUT/bench_c$c.log; echo; done
