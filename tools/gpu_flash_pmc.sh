#!/bin/bash
# SQ counters of the flash kernels (own run: --pmc only)
R=$PWD; mkdir -p $R/gpurun_out/flpmc; export TMPDIR=/tmp; cd /tmp
for cs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $cs | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $cs --output-format csv -d $R/gpurun_out/flpmc/$tag -o pmc -- python $R/tools/flash_bench.py > $R/gpurun_out/flpmc/$tag.log 2>&1
  tail -2 $R/gpurun_out/flpmc/$tag.log
done
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/flpmc/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'flash' not in k: continue
    k = k.split('(')[0][-40:]
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
  print(k)
  for c, v in sorted(d.items()): print('   %-28s %14.0f (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
