#!/bin/bash
# HBM traffic of the normalisation backward (the launch pair behind tg_norm_act_bwd) from PMC counters, one counter per
# pass, counters only -> gpurun_out/pmc_norm/summary.json (rows in the format of tools/pmc_parse.py, kernel = the entry point)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD; export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_norm; rm -rf $OUT; mkdir -p $OUT
cd /tmp
for spec in "16 256 32" "32 256 32" "16 256 64" "64 128 32" "128 64 32"; do
  set -- $spec
  for cs in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $cs --output-format csv -d $OUT/c$1_hw$2_n$3_$cs -o pmc -- python $REPO/tools/pmc_norm.py $1 $2 $3 3 > $OUT/c$1_hw$2_n$3_$cs.log 2>&1
  done
done
cd $REPO
python - <<'PY'
import csv, glob, json, os, re
root = 'gpurun_out/pmc_norm'
rows = []
for d in sorted(glob.glob(root + '/c*_FETCH_SIZE')):
  m = re.match(r'c(\d+)_hw(\d+)_n(\d+)_', os.path.basename(d))
  c, hw, n = (int(v) for v in m.groups())
  tot = {}
  for cs in ('FETCH_SIZE', 'WRITE_SIZE'):
    files = glob.glob(os.path.join(d.replace('FETCH_SIZE', cs), '**', '*counter_collection.csv'), recursive=True)
    per = {}
    for r in csv.DictReader(open(files[0])):
      name = r['Kernel_Name']
      if 'norm_act_bwd' in name and r['Counter_Name'] == cs:
        k = 'bwd1' if 'bwd1' in name else 'bwd2'
        per.setdefault(k, []).append(float(r['Counter_Value']))
    tot[cs] = {k: sum(v) / len(v) for k, v in per.items()}      # KiB per launch of each of the two kernels
  fetch = sum(tot['FETCH_SIZE'].values()); write = sum(tot['WRITE_SIZE'].values())
  alg = 3 * n * hw * hw * c * 2      # gz and y read once, gy written once
  hbm = int((2.0 * fetch + write) * 1024)      # FETCH_SIZE counts 64 B per 128-B request on gfx950
  rows.append(dict(kernel='tg_norm_act_bwd', shape='norm_act_bwd:c%d:hw%d:n%d' % (c, hw, n), case='norm', op='bwd', n=n,
                   algorithmic_bytes_per_launch=alg, hbm_bytes_per_launch=hbm, traffic_over_algorithmic=round(hbm / alg, 3),
                   FETCH_SIZE_KiB_per_launch=fetch, WRITE_SIZE_KiB_per_launch=write, per_kernel_KiB=tot))
json.dump(dict(kernels=rows), open(root + '/summary.json', 'w'), indent=1)
for r in rows:
  print(r['shape'], 'traffic/alg', r['traffic_over_algorithmic'], r['per_kernel_KiB'])
PY
