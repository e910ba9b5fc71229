#!/bin/bash
# First GPU pass: parity tests (all, no -x), smoke, bench, rocprof kernel stats.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/box.txt 2>&1
nproc >> gpurun_out/box.txt
timeout 900 python -m pytest tests -m gpu -q --maxfail=60 --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1
echo "bench exit $?" >> gpurun_out/bench.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-roofline --no-cpu-baseline > "$OLDPWD/gpurun_out/prof.log" 2>&1
cd "$OLDPWD"
find gpurun_out/prof -name "*kernel_stats*" | head -3 >> gpurun_out/prof.log
tail -5 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; tail -2 gpurun_out/bench.log
