#!/bin/bash
# conv statistics epilogue: op tests + kbench A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r2d; mkdir -p $OUT; rm -f $OUT/kb.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -p no:cacheprovider -k "statistics_epilogue or norm_act" > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
for b in 32 64; do
for c in E256a E256b E128a E128b; do
  timeout 120 tools/kbench.bin $c --op stats --batch $b --iters 20 --nocheck 2>&1 | grep -v "^case" >> $OUT/kb.txt
done; done
for b in 32 64; do
for c in E128b; do
  TG_STATS_WIDE_TILE=1 timeout 120 tools/kbench.bin $c --op stats --batch $b --iters 20 --nocheck 2>&1 | grep -v "^case" >> $OUT/kb.txt
done; done
cat $OUT/kb.txt
