#!/bin/bash
# round-3 seventh pass: the quadrant filter-gradient kernel (64 x 64 blocks, 8 waves) against the tile kernel, same box,
# interleaved (TG_TUNE_WG_QUAD), kbench cross-checks both against the direct kernels at batch 8
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3g; mkdir -p $OUT; export TMPDIR=/tmp
for c in E64a E64b E32a E32b E16 G16a G32a G32b G64a; do
  for q in 0 1 0 1; do
    echo -n "quad=$q " >> $OUT/kb_wgrad.txt
    TG_TUNE_WG_QUAD=$q timeout 120 tools/kbench.bin $c --op wgrad --batch 64 --iters 30 2>&1 | grep -v "^case" | tail -1 >> $OUT/kb_wgrad.txt
  done
done
for c in E64a E32a E16; do
  for q in 0 1; do
    echo -n "quad=$q " >> $OUT/kb_wgradb.txt
    TG_TUNE_WG_QUAD=$q timeout 120 tools/kbench.bin $c --op wgradb --batch 48 --iters 30 2>&1 | grep -v "^case" | tail -1 >> $OUT/kb_wgradb.txt
  done
done
for c in E64a E32a E16; do
  for q in 0 1; do
    echo -n "quad=$q n16 " >> $OUT/kb_wgrad16.txt
    TG_TUNE_WG_QUAD=$q timeout 120 tools/kbench.bin $c --op wgrad --batch 16 --iters 30 2>&1 | grep -v "^case" | tail -1 >> $OUT/kb_wgrad16.txt
  done
done
TG_TUNE_WG_QUAD=1 timeout 900 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_ops.py -q -m gpu -x --tb=short -p no:cacheprovider -k "(wgrad or weight or upcat or conv_variants or conv) and not dispatch_table" > $OUT/pytest_wgrad.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_wgrad.log
for q in 0 1 0 1; do
  TG_TUNE_WG_QUAD=$q timeout 300 python bench.py --no-cpu-baseline --no-roofline > $OUT/bench_c3_quad${q}_$RANDOM.log 2>> $OUT/bench.err
done
cat $OUT/kb_wgrad.txt $OUT/kb_wgradb.txt $OUT/kb_wgrad16.txt; tail -3 $OUT/pytest_wgrad.log
for f in $OUT/bench_c3_quad*.log; do echo $f; head -c 200 $f | cut -c 90-200; echo; done
