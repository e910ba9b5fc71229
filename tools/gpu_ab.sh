#!/bin/bash
# tools/gpu_ab.sh <outdir> <rounds> <variant> [<variant> ...]: interleaved A/B on ONE box of the in-tree library ('cur')
# and the alternative builds tools/ab/<variant>.so (tools/build_variant.sh), bench config 3, images/s and ms per step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; ROUNDS=$2; shift 2; mkdir -p $OUT; export TMPDIR=/tmp
for i in $(seq 1 $ROUNDS); do
  for which in cur "$@"; do
    if [ $which = cur ]; then unset TG_LIB_PATH; else export TG_LIB_PATH=$PWD/tools/ab/$which.so; fi
    timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps ${STEPS:-20} --warmup 3 ${BENCH_ARGS} 2>$OUT/ab_${which}_$i.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.log
  done
done
unset TG_LIB_PATH
