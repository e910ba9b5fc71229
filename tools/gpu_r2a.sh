#!/bin/bash
# round-2 first GPU pass: tests (recording the dispatched kernel symbols), RCCL one-rank smoke, bench plain / segmented
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r2a; mkdir -p $OUT; export TMPDIR=/tmp
nproc > $OUT/box.txt
TG_RECORD_KERNELS=$OUT/bench_dispatch_kernels.json timeout 1200 python -m pytest tests -m gpu -q --maxfail=40 --tb=short -p no:cacheprovider -s --durations=25 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python tools/rccl_smoke.py > $OUT/rccl_smoke.log 2>&1; echo "exit $?" >> $OUT/rccl_smoke.log
timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench.log 2> $OUT/bench.err; echo "bench exit $?" >> $OUT/bench.log
timeout 300 python bench.py --steps 10 --warmup 3 --overlap on --no-roofline --no-cpu-baseline > $OUT/bench_seg.log 2> $OUT/bench_seg.err; echo "bench exit $?" >> $OUT/bench_seg.log
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3; tail -2 $OUT/rccl_smoke.log; tail -c 400 $OUT/bench_seg.log
