#!/bin/bash
# kbench on selected cases: tools/gpu_kb.sh "<cases>" <op> <batch> [env...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/kb; mkdir -p $OUT
for c in $1; do
  timeout 120 tools/kbench.bin $c --op $2 --batch $3 --iters 20 2>&1 | grep -v "^case" >> $OUT/kb.txt
done
cat $OUT/kb.txt
