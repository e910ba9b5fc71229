#!/bin/bash
# round-3 first pass: (1) is a 16-bit trajectory reproducible? eager x eager, graph x eager, with and without
# TG_DETERMINISTIC; (2) the full GPU suite in the new order (primitives first), no -x; (3) bench line of config 3
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD; OUT=gpurun_out/r3a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 400 python tools/fp16_repro.py --reps 5 > $OUT/fp16_repro.log 2>&1; echo "exit $?" >> $OUT/fp16_repro.log
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python bench.py > $OUT/bench_c3.log 2> $OUT/bench_c3.err; echo "exit $?" >> $OUT/bench_c3.log
cat $OUT/fp16_repro.log; tail -3 $OUT/pytest_gpu.log; head -c 300 $OUT/bench_c3.log; echo
