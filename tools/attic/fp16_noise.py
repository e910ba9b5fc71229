"""Run-to-run noise of the default (atomics) 16-bit mode: N eager + N graph fp16 trajectories of the test's size against
eager #0 -- update rel-L2 and tensors not bit-identical.  One line per environment; run once per switch setting:
  TG_TUNE_SLAB_PLAIN=0 python tools/fp16_noise.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.fp16_repro import err, trajectory      # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
p0, e0 = trajectory('fp16', False)
ee = [err(trajectory('fp16', False)[1], e0, p0) for _ in range(reps)]
gg = [err(trajectory('fp16', True)[1], e0, p0) for _ in range(reps)]
tag = ' '.join('%s=%s' % (k, v) for k, v in sorted(os.environ.items()) if k.startswith('TG_'))
print('[%s] eager-vs-eager0 %s   graph-vs-eager0 %s' % (tag or 'default', ['%.2e/%d' % e for e in ee], ['%.2e/%d' % e for e in gg]))
