#!/bin/bash
# round-3 second pass: the new parity tests only (config-4 shape pins, deterministic mode, larger RGB filter)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_model.py -m gpu -q --maxfail=30 --tb=short -p no:cacheprovider -s \
  -k "config4 or deterministic or larger_filter or fp16_training or fp16_conv" > $OUT/pytest_new.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_new.log
grep -n "\[config4\|\[fp16\]\|\[grads\]\|passed\|failed\|FAILED\|pytest exit" $OUT/pytest_new.log | tail -40
