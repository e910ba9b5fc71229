#!/bin/bash
# margins of the round-3 parity bounds (printed figures of the tests that carry one)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3l; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_model.py -q -m gpu -s --tb=short -p no:cacheprovider \
  -k "flash_attention_at_config4 or config4_half or losses_and_gradients or fp16_training or sign_bit" > $OUT/pytest_margins.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_margins.log
grep -h "\[flash c4\|\[config4\|\[sensitivity\|\[fp16\]\|\[grads\]\|passed\|failed" $OUT/pytest_margins.log | head -80
