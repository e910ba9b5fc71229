cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/ldspmc; rm -rf $O; mkdir -p $O
cd /tmp
for spec in "E32a fwd 64" "E16 fwd 64" "E256a fwd 64" "G64a fwd 64" "E64a wgrad 64"; do
  set -- $spec
  i=0
  for cs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $cs --output-format csv -d $O/$1_$2_$i -o pmc -- $R/tools/kbench.bin $1 --op $2 --batch $3 --nocheck --iters 3 > $O/$1_$2_$i.log 2>&1
  done
done
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/ldspmc/**/*counter_collection.csv', recursive=True):
  case = f.split('/')[2].rsplit('_',1)[0]
  for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'pack' in k or 'slab_reduce' in k: continue
    acc[case+' '+k.split('(')[0][-60:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(acc.items()):
  print(k); print('   ', ' '.join('%s=%.4g' % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
