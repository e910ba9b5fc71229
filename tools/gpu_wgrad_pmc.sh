#!/bin/bash
# SQ issue / wait / LDS counters of the filter-gradient tile kernel at the mid layers
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/wgpmc; rm -rf $O; mkdir -p $O
make -C twingan_amd/csrc kbench > /dev/null 2>&1
cd /tmp
for spec in "E64a wgrad 64" "E32a wgrad 64" "E128a wgrad 64"; do
  set -- $spec
  i=0
  for cs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $cs --output-format csv -d $O/$1_$i -o pmc -- $R/tools/kbench.bin $1 --op $2 --batch $3 --nocheck --iters 3 > $O/$1_$i.log 2>&1
  done
done
cd $R
python - <<'PY'
import csv, glob, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/wgpmc/**/*counter_collection.csv', recursive=True):
  case = f.split('/')[2].split('_')[0]
  for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'wgrad_tile' not in k: continue
    acc[case][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(acc.items()):
  print(k, ' '.join('%s=%.3g' % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
tail -3 $O/E64a_3.log
