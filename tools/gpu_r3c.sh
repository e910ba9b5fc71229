#!/bin/bash
# round-3 third pass: full GPU suite (f16 fused epilogues, persistent w_bar packs, D streams under spectral norm),
# config-4 bench A/B (TG_SN_DOMAIN_STREAMS), config-0 preset, the N > 1 schedule + diagnostics on one GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python bench.py --config 4 --no-cpu-baseline --no-roofline > $OUT/bench_c4.log 2> $OUT/bench_c4.err; echo "exit $?" >> $OUT/bench_c4.log
TG_SN_DOMAIN_STREAMS=0 timeout 300 python bench.py --config 4 --no-cpu-baseline --no-roofline > $OUT/bench_c4_1stream.log 2> $OUT/bench_c4_1stream.err; echo "exit $?" >> $OUT/bench_c4_1stream.log
timeout 200 python bench.py --config 0 --no-cpu-baseline --no-roofline > $OUT/bench_c0.log 2> $OUT/bench_c0.err; echo "exit $?" >> $OUT/bench_c0.log
timeout 300 python bench.py --reduce-always --overlap on --no-cpu-baseline --no-roofline > $OUT/bench_c3_reduce.log 2> $OUT/bench_c3_reduce.err; echo "exit $?" >> $OUT/bench_c3_reduce.log
tail -4 $OUT/pytest_gpu.log
for f in bench_c4 bench_c4_1stream bench_c0 bench_c3_reduce; do echo "== $f"; head -c 600 $OUT/$f.log; echo; tail -3 $OUT/$f.err; done
