"""Lists every conv entry point + layer shape one bench step (256x256, batch 16, bf16) dispatches, from the per-shape
timing table bench.py's roofline pass dumps (TG_DUMP_SHAPES) -> tests/golden/bench_dispatch_shapes.json, the case list
of tests/test_gpu_bench_shapes.py.  Usage: python tools/make_dispatch_shapes.py profiles/rNN_shapes_eager_step.json"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(src):
  rows = json.load(open(src))
  out = {}
  for r in rows:
    m = re.match(r'(tg_conv2d\w+)\[(\w+):(\w+):k(\d):c([\d+]+)>(\d+):hw(\d+):n([\d+]+)\]', r['kernel'])
    if not m:
      continue
    ep, _, _, k, cin, cout, hw, n = m.groups()
    key = 'k%s:c%s>%s:hw%s' % (k, cin, cout, hw)
    out.setdefault(key, {}).setdefault(ep, [])
    if n not in out[key][ep]:
      out[key][ep].append(n)
  for key in out:
    for ep in out[key]:
      out[key][ep].sort(key=lambda s: [int(v) for v in s.split('+')])
  dst = os.path.join(ROOT, 'tests', 'golden', 'bench_dispatch_shapes.json')
  with open(dst, 'w') as fh:
    json.dump(dict(source=os.path.relpath(src, ROOT), layers=out), fh, indent=1, sort_keys=True)
  print(dst, len(out), 'layer shapes,', sum(len(v) for e in out.values() for v in e.values()), 'dispatches')


if __name__ == '__main__':
  main(sys.argv[1])
