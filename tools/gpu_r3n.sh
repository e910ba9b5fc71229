#!/bin/bash
# second-box repeat of the driver's own commands: full GPU suite with -x, smoke, default bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r3n}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
timeout 400 python bench.py > $OUT/bench_c3.log 2> $OUT/bench_c3.err; echo "exit $?" >> $OUT/bench_c3.log
tail -3 $OUT/pytest_gpu.log; tail -1 $OUT/smoke.log; head -c 300 $OUT/bench_c3.log | cut -c 80-300; echo
rocm-smi --showproductname 2>/dev/null | head -5; hostname
