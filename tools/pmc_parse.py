#!/usr/bin/env python
"""Parses the rocprofv3 --pmc csv outputs of tools/pmc_traffic.sh into per-launch HBM bytes.
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request, so wide coalesced
reads are doubled (MI355X_MICROARCH.md, HBM section).  WRITE_SIZE is used as reported (uncalibrated)."""
import csv
import glob
import json
import os
import re
import sys

CASES = {'G256a': (256, 64, 16), 'E256a': (256, 16, 16), 'E128a': (128, 32, 32), 'G32a': (32, 512, 128), 'E32b': (32, 128, 256)}


def main(root):
  res = {}
  for d in sorted(glob.glob(os.path.join(root, '*_*_n*_*'))):
    if not os.path.isdir(d):
      continue
    m = re.match(r'(\w+?)_(fwd|dgrad|wgrad)_n(\d+)_(\w+)$', os.path.basename(d))
    if not m:
      continue
    case, op, n, ctr = m.group(1), m.group(2), int(m.group(3)), m.group(4)
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not files:
      continue
    per_kernel = {}
    for row in csv.DictReader(open(files[0])):
      name = row.get('Kernel_Name', '')
      if row.get('Counter_Name') != ctr:
        continue
      if not any(t in name for t in ('conv_tile', 'conv_wgrad_tile', 'conv_small', 'conv_fwd_mfma', 'conv_wgrad_mfma')):
        continue
      per_kernel.setdefault(name, []).append(float(row['Counter_Value']))
    if not per_kernel:
      continue
    name, vals = max(per_kernel.items(), key=lambda kv: sum(kv[1]))
    hw, cin, cout = CASES[case]
    key = (case, op, n)
    ent = res.setdefault(key, dict(case=case, op=op, n=n, kernel=name[:80],
                                   shape='%s:mfma:k3:c%d>%d:hw%d:n%d' % (op, cin, cout, hw, n)))
    ent[ctr + '_KiB_per_launch'] = sum(vals) / len(vals)
  out = []
  for ent in res.values():
    f, w = ent.get('FETCH_SIZE_KiB_per_launch'), ent.get('WRITE_SIZE_KiB_per_launch')
    if f is None or w is None:
      continue
    hw, cin, cout = CASES[ent['case']]
    px = ent['n'] * hw * hw
    ent['hbm_bytes_per_launch'] = int((2.0 * f + w) * 1024)
    ent['algorithmic_bytes_per_launch'] = int(2 * px * (cin + cout) + 2 * 9 * cin * cout)
    ent['traffic_over_algorithmic'] = round(ent['hbm_bytes_per_launch'] / ent['algorithmic_bytes_per_launch'], 3)
    out.append(ent)
  print(json.dumps(dict(note='rocprofv3 --pmc, one counter per pass; FETCH_SIZE x2 (gfx950), WRITE_SIZE as reported',
                        kernels=out), indent=1))


if __name__ == '__main__':
  main(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc')
