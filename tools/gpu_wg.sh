#!/bin/bash
# filter-gradient kernels: parity tests at the bench shapes + per-layer timings
cd "${GRAFT_REPO_ROOT:-/root/repo}"
make -C twingan_amd/csrc kbench > /dev/null 2>&1
timeout 600 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_ops.py -x -q -m gpu -k "wgrad or weight or bwd_weight or upcat" 2>&1 | tail -3
for c in E256a E256b E128a E128b E64a E64b E32a E32b E16 E8; do
  timeout 120 tools/kbench.bin $c --op wgrad --batch 64 --iters 20 2>&1 | grep -v "^case" | tail -1
done
