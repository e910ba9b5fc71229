#!/bin/bash
# round-3 fourth pass: filter-gradient tile kernel, 4 waves x 8-row tiles vs 8 waves x 16-row tiles (TG_TUNE_WG_NW),
# same box, interleaved; then the filter-gradient parity tests and a bench line with the new default
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3d; mkdir -p $OUT; export TMPDIR=/tmp
for c in E128a E128b E64a E64b E32a E32b E16 G16a G32a G32b G64a G128a; do
  for nw in 4 8 4 8; do
    echo -n "nw=$nw " >> $OUT/kb_wgrad.txt
    TG_TUNE_WG_NW=$nw timeout 120 tools/kbench.bin $c --op wgrad --batch 64 --iters 30 2>&1 | grep -v "^case" | tail -1 >> $OUT/kb_wgrad.txt
  done
done
for c in E128a E64a E32a E16; do
  for nw in 4 8; do
    echo -n "nw=$nw " >> $OUT/kb_wgradb.txt
    TG_TUNE_WG_NW=$nw timeout 120 tools/kbench.bin $c --op wgradb --batch 48 --iters 30 2>&1 | grep -v "^case" | tail -1 >> $OUT/kb_wgradb.txt
  done
done
TG_RECORD_KERNELS=$PWD/$OUT/bench_dispatch_kernels.json timeout 900 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_ops.py -q -m gpu -x --tb=short -p no:cacheprovider -k "wgrad or weight or upcat or conv_variants or dispatch_table or conv" > $OUT/pytest_wgrad.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_wgrad.log
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_c3.log 2> $OUT/bench_c3.err; echo "exit $?" >> $OUT/bench_c3.log
TG_TUNE_WG_NW=4 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $OUT/bench_c3_nw4.log 2> $OUT/bench_c3_nw4.err
timeout 300 python bench.py --no-cpu-baseline --no-roofline > $OUT/bench_c3_nw8.log 2> $OUT/bench_c3_nw8.err
cat $OUT/kb_wgrad.txt $OUT/kb_wgradb.txt; tail -3 $OUT/pytest_wgrad.log
for f in bench_c3 bench_c3_nw4 bench_c3_nw8; do head -c 330 $OUT/$f.log; echo; done
