#!/bin/bash
# round-3 fifth pass: sign bits instead of the block-end conv outputs of the discriminators -- parity tests, then the
# bench step with and without them, interleaved on one box; dispatch table re-recorded
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3e; mkdir -p $OUT; export TMPDIR=/tmp
TG_RECORD_KERNELS=$PWD/$OUT/bench_dispatch_kernels.json timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bench_shapes.py tests/test_gpu_model.py -q -m gpu -x --tb=short -p no:cacheprovider \
  -k "sign or conv_variants or dispatch_table or losses_and_gradients or graph_replay or config4_half" > $OUT/pytest_signs.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_signs.log
for i in 1 2; do
  TG_POOL_SIGNS=0 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $OUT/bench_c3_nosigns_$i.log 2> $OUT/bench_c3_nosigns_$i.err
  timeout 300 python bench.py --no-cpu-baseline --no-roofline > $OUT/bench_c3_signs_$i.log 2> $OUT/bench_c3_signs_$i.err
done
tail -5 $OUT/pytest_signs.log
for f in nosigns_1 signs_1 nosigns_2 signs_2; do head -c 300 $OUT/bench_c3_$f.log | cut -c 90-260; echo; tail -2 $OUT/bench_c3_$f.err; done
