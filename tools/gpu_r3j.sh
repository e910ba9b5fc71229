#!/bin/bash
# round-3: conv_img (8x8 / 4x4 images staged in LDS) against conv_small, same box, interleaved (TG_TUNE_CONV_IMG)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3j; mkdir -p $OUT; export TMPDIR=/tmp
for c in E8 G8a G4; do
  for b in 16 32 64; do
    for op in fwd dgrad; do
      for q in 0 1 0 1; do
        echo -n "img=$q n=$b " >> $OUT/kb_img.txt
        TG_TUNE_CONV_IMG=$q timeout 120 tools/kbench.bin $c --op $op --batch $b --iters 50 2>&1 | grep -v "^case" | tail -1 >> $OUT/kb_img.txt
      done
    done
  done
done
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bench_shapes.py -q -m gpu -x --tb=short -p no:cacheprovider -k "not dispatch_table" > $OUT/pytest_ops.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_ops.log
for q in 0 1 0 1; do
  TG_TUNE_CONV_IMG=$q timeout 300 python bench.py --no-cpu-baseline --no-roofline > $OUT/bench_c3_img${q}_$RANDOM.log 2>> $OUT/bench.err
done
cat $OUT/kb_img.txt; tail -3 $OUT/pytest_ops.log
for f in $OUT/bench_c3_img*.log; do echo $f; head -c 200 $f | cut -c 90-200; echo; done
