#!/bin/bash
# round-3 sixth pass: where did the sign-bit saving go?  rocprofv3 kernel statistics of the bench step with and without
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD; OUT=gpurun_out/r3f; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for m in 0 1; do
  TG_POOL_SIGNS=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_signs$m -o bench -- python $REPO/bench.py --steps 4 --warmup 1 --no-roofline --no-cpu-baseline > $REPO/$OUT/prof_signs$m.log 2>&1
done
cd $REPO; find $OUT -name "*kernel_trace.csv" -delete
timeout 300 python -m pytest tests/test_gpu_model.py -q -m gpu -x --tb=short -p no:cacheprovider -k "sign_bit" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for m in 0 1; do f=$(find $OUT/prof_signs$m -name "*kernel_stats.csv" | head -1); echo "== signs=$m $f"; head -25 $f | cut -c1-200; done
