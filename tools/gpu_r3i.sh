#!/bin/bash
# round-3: dispatch table re-recorded with the quadrant kernel's heuristic, then the FULL suite; PMC of the norm backward
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3i; mkdir -p $OUT; export TMPDIR=/tmp
TG_RECORD_KERNELS=$PWD/$OUT/bench_dispatch_kernels.json timeout 900 python -m pytest tests/test_gpu_bench_shapes.py -q -m gpu --tb=short -p no:cacheprovider > $OUT/pytest_record.log 2>&1
cp $OUT/bench_dispatch_kernels.json tests/golden/bench_dispatch_kernels.json
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
bash tools/pmc_norm.sh > $OUT/pmc_norm.log 2>&1; cp gpurun_out/pmc_norm/summary.json $OUT/pmc_norm_summary.json
tail -3 $OUT/pytest_record.log; tail -3 $OUT/pytest_gpu.log; tail -6 $OUT/pmc_norm.log
