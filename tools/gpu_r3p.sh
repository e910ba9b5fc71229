#!/bin/bash
# round 3, pass p: the new rows' GPU tests (native normalisers, preprocessing modes, embedding dataset) + an interleaved
# A/B of the streaming stores' cache policy (tools/ab/aux16.so = sc1, aux2.so = nt, built with -DTG_STORE_AUX=...)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r3p}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_data.py -q -m gpu \
  -k "layer_norm or native or batch_renorm or preprocess or cropping or distillation_trains or loader" > $OUT/pytest_new.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_new.log
tail -5 $OUT/pytest_new.log
for i in 1 2; do
  for which in cur aux16 aux2; do
    if [ $which = cur ]; then unset TG_LIB_PATH; else export TG_LIB_PATH=$PWD/tools/ab/$which.so; fi
    timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 3 2>$OUT/ab_${which}_$i.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.log
  done
done
unset TG_LIB_PATH
hostname
