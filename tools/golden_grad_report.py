"""Per-variable gradient errors of the fp32 HIP path against a reference-generated fixture (diagnostic for the
bounds of tests/test_golden.py).  usage: python tools/golden_grad_report.py twingan_hw16_c8_style [more fixtures]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_golden as TG      # noqa: E402
from twingan_amd import Config      # noqa: E402
from twingan_amd import twingan as T      # noqa: E402

for name in sys.argv[1:]:
  g = TG.load(name)
  cfg = Config(precision='fp32', **TG.product_kw(name))
  tr = T.Trainer(cfg, device='cuda:0', seed=0)
  noise = TG._dev(g['in/style_noise']) if 'in/style_noise' in g else None
  tr.store.load_state_dict({k[len('param/'):]: torch.from_numpy(v).float() for k, v in g.items() if k.startswith('param/')})
  s, t = TG._dev(g['in/sources']), TG._dev(g['in/targets'])
  a_s, a_t = TG._dev(g['in/gp_alpha_s']), TG._dev(g['in/gp_alpha_t'])
  n_s = TG._dev(g['in/dragan_noise_s']) if 'in/dragan_noise_s' in g else None
  n_t = TG._dev(g['in/dragan_noise_t']) if 'in/dragan_noise_t' in g else None
  for group, fn, args in (('g', T.generator_loss, (s, t, cfg, noise)),
                          ('d', T.discriminator_loss, (s, t, cfg, a_s, a_t, n_s, n_t, noise))):
    tr.store.zero_grad(group)
    tr._set_requires_grad(g=group == 'g', d=group == 'd')
    tr.P.__dict__.get('sn_cache', {}).clear()
    loss, terms = fn(tr.P, *args)
    loss.backward()
    gd = tr.store.grad_dict()
    rows = []
    for k in tr.store.names(group):
      ref = g['grad/' + k]
      rows.append((float(np.linalg.norm(gd[k].double().cpu().numpy() - ref)), float(np.linalg.norm(ref)), k))
    tot = (sum(r[0] ** 2 for r in rows) / sum(r[1] ** 2 for r in rows)) ** 0.5
    print('%s %s: aggregate %.3e, batch %d' % (name, group, tot, s.shape[0]))
    for e, n, k in sorted(rows, key=lambda r: -r[0])[:8]:
      print('    abs err %.3e  ref norm %.3e  rel %.3e  %s' % (e, n, e / (n + 1e-30), k))
