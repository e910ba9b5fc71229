#!/bin/bash
# deterministic mode with the ordered two-stage sums: its tests and its cost on the bench step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3m; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -m gpu -x --tb=short -p no:cacheprovider \
  -k "ordered or deterministic or fp16_training or bit_reproducible or bias or pointwise or loss" > $OUT/pytest_det.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_det.log
timeout 300 python bench.py --no-cpu-baseline --no-roofline > $OUT/bench_c3.log 2> $OUT/bench_c3.err
TG_DETERMINISTIC=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $OUT/bench_c3_det.log 2> $OUT/bench_c3_det.err
TG_DETERMINISTIC=1 timeout 300 python bench.py --config 4 --no-cpu-baseline --no-roofline > $OUT/bench_c4_det.log 2> $OUT/bench_c4_det.err
tail -3 $OUT/pytest_det.log
for f in c3 c3_det c4_det; do echo -n "$f: "; head -c 260 $OUT/bench_$f.log | cut -c 80-260; echo; tail -2 $OUT/bench_$f.err; done
