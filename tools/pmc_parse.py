#!/usr/bin/env python
"""Parses the rocprofv3 --pmc csv outputs of tools/pmc_kernels.sh into, per (kbench case, op, batch): HBM bytes per
launch and MFMA utilisation of the conv kernel that ran.

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request, so wide coalesced
reads are doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is used as reported (uncalibrated).
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES (summed over all SIMDs; measured: exactly 32 cycles per
v_mfma_f32_32x32x16_bf16, i.e. MFMA flops / 1024) / (kernel cycles x 1024 SIMDs) -- rocprofiler's MfmaUtil expression for
gfx950 (counter_defs.yaml: reduce(SQ_VALU_MFMA_BUSY_CYCLES,sum) / (reduce(GRBM_GUI_ACTIVE,max) * SIMD_NUM)).  The csv
reports GRBM_GUI_ACTIVE summed over the 8 XCDs (checked against the kernels' wall time at ~2.1 GHz), so kernel cycles =
GRBM_GUI_ACTIVE / 8.  mfma_flops_per_launch = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 (padding included)."""
import csv
import glob
import json
import os
import re
import sys

CASES = {'E256a': (256, 16, 16), 'E256b': (256, 16, 32), 'E128a': (128, 32, 32), 'E128b': (128, 32, 64),
         'E64a': (64, 64, 64), 'E64b': (64, 64, 128), 'E32a': (32, 128, 128), 'E32b': (32, 128, 256),
         'E16': (16, 256, 256), 'E8': (8, 256, 256), 'D4': (4, 264, 256), 'G4': (4, 256, 256), 'G8a': (8, 512, 256),
         'G16a': (16, 512, 256), 'G32a': (32, 512, 128), 'G32b': (32, 128, 128), 'G64a': (64, 256, 64),
         'G128a': (128, 128, 32), 'G256a': (256, 64, 16)}
CONV = ('conv_tile', 'conv_thin16', 'conv_img', 'conv_wgrad_tile', 'conv_wgrad_quad', 'conv_wgrad_thin', 'conv_small', 'conv_fwd_mfma',
        'conv_wgrad_mfma')
N_SIMD = 256 * 4
N_XCD = 8


def short(name):
  m = re.search(r'(conv_\w+?_kernel|conv_\w+_mfma)', name)
  if not m:
    return name[:60]
  targs = re.findall(r'Li(\d+)E', name) or re.findall(r'<([^>]*)>', name)
  return m.group(1) + ('<%s>' % ','.join(targs) if targs else '')


def main(root):
  res = {}
  for d in sorted(glob.glob(os.path.join(root, '*_*_n*_*'))):
    if not os.path.isdir(d):
      continue
    m = re.match(r'(\w+?)_(fwd|dgrad|wgrad)_n(\d+)_(\w+)$', os.path.basename(d))
    if not m:
      continue
    case, op, n = m.group(1), m.group(2), int(m.group(3))
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not files:
      continue
    vals = {}      # kernel -> counter -> [values per dispatch]
    for row in csv.DictReader(open(files[0])):
      name = row.get('Kernel_Name', '')
      if any(t in name for t in CONV):
        vals.setdefault(name, {}).setdefault(row.get('Counter_Name'), []).append(float(row['Counter_Value']))
    if not vals:
      continue
    # the conv kernel of the case (the slab reduction and the pack kernels are filtered by name above)
    name, ctrs = max(vals.items(), key=lambda kv: sum(sum(v) for v in kv[1].values()))
    hw, cin, cout = CASES[case]
    ent = res.setdefault((case, op, n), dict(case=case, op=op, n=n, kernel=short(name),
                                             shape='%s:mfma:k3:c%d>%d:hw%d:n%d' % (op, cin, cout, hw, n)))
    for c, v in ctrs.items():
      ent[c] = sum(v) / len(v)
  out = []
  for ent in res.values():
    hw, cin, cout = CASES[ent['case']]
    px = ent['n'] * hw * hw
    ent['algorithmic_bytes_per_launch'] = int(2 * px * (cin + cout) + 2 * 9 * cin * cout)
    ent['algorithmic_flops_per_launch'] = int(2 * px * 9 * cin * cout)
    f, w = ent.get('FETCH_SIZE'), ent.get('WRITE_SIZE')
    if f is not None and w is not None:
      ent['FETCH_SIZE_KiB_per_launch'], ent['WRITE_SIZE_KiB_per_launch'] = f, w
      ent['hbm_bytes_per_launch'] = int((2.0 * f + w) * 1024)
      ent['traffic_over_algorithmic'] = round(ent['hbm_bytes_per_launch'] / ent['algorithmic_bytes_per_launch'], 3)
    busy, act = ent.get('SQ_VALU_MFMA_BUSY_CYCLES'), ent.get('GRBM_GUI_ACTIVE')
    if busy is not None and act:
      ent['mfma_util'] = round(busy / (act / N_XCD * N_SIMD), 4)
      ent['kernel_cycles'] = int(act / N_XCD)
      mops = ent.get('SQ_INSTS_VALU_MFMA_MOPS_BF16')
      if mops is not None:
        ent['mfma_flops_per_launch'] = int(mops * 512)
        ent['mfma_flops_over_algorithmic'] = round(mops * 512 / ent['algorithmic_flops_per_launch'], 3)
        # the part of the MFMA time that is not channel padding
        ent['mfma_util_useful'] = round(ent['mfma_util'] / max(ent['mfma_flops_over_algorithmic'], 1e-9), 4)
    out.append(ent)
  print(json.dumps(dict(note='rocprofv3 --pmc, one counter set per pass (tools/pmc_kernels.sh); FETCH_SIZE x2 (gfx950), '
                             'WRITE_SIZE as reported; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)',
                        kernels=out), indent=1))


if __name__ == '__main__':
  main(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc2')
