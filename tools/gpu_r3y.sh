#!/bin/bash
# round 3, pass y: deferred + batched slab reductions (TG_WGRAD_DEFER=0/1), configs 3 and 4, + the golden / model tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r3y}; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2; do
  for d in 0 1; do
    TG_WGRAD_DEFER=$d timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 3 2>$OUT/err_c3_$d.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 defer $d', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.log
  done
done
for d in 0 1; do
  TG_WGRAD_DEFER=$d timeout 300 python bench.py --config 4 --no-cpu-baseline --no-roofline --steps 12 --warmup 3 2>$OUT/err_c4_$d.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 defer $d', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.log
done
timeout 900 python -m pytest tests/test_golden.py tests/test_gpu_model.py -q -m gpu -x > $OUT/pytest_model.log 2>&1; tail -3 $OUT/pytest_model.log
