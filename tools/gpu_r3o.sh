#!/bin/bash
# slab reductions of the filter gradients on an auxiliary stream (TG_WGRAD_AUX): interleaved bench A/B, then the model /
# golden / data-parallel tests with it on
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3o; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2; do
  TG_WGRAD_AUX=0 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $OUT/bench_c3_aux0_$i.log 2>> $OUT/bench.err
  TG_WGRAD_AUX=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $OUT/bench_c3_aux1_$i.log 2>> $OUT/bench.err
done
TG_WGRAD_AUX=1 timeout 300 python bench.py --config 4 --no-cpu-baseline --no-roofline > $OUT/bench_c4_aux1.log 2>> $OUT/bench.err
TG_WGRAD_AUX=1 timeout 300 python bench.py --reduce-always --overlap on --no-cpu-baseline --no-roofline > $OUT/bench_c3_aux1_reduce.log 2>> $OUT/bench.err
timeout 1200 python -m pytest tests/test_golden.py tests/test_gpu_model.py -q -m gpu -x --tb=short -p no:cacheprovider > $OUT/pytest_model.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_model.log
for f in aux0_1 aux1_1 aux0_2 aux1_2; do echo -n "$f: "; head -c 200 $OUT/bench_c3_$f.log | cut -c 90-200; echo; done
head -c 200 $OUT/bench_c4_aux1.log | cut -c 80-200; echo; head -c 200 $OUT/bench_c3_aux1_reduce.log | cut -c 90-200; echo
tail -3 $OUT/pytest_model.log; tail -3 $OUT/bench.err
