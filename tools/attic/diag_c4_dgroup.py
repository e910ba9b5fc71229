"""Where does the config-4 (fp16, spectral norm + attention) discriminator-group gradient deviation from the float64 oracle
sit?  Per variable: |hip - ref| and |ref|, sorted by contribution to the group's squared error; per loss term likewise."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from oracle import torch_ref as R                  # noqa: E402
from test_gpu_model import make                    # noqa: E402
from twingan_amd import pggan, twingan as T        # noqa: E402

kw = dict(hw=32, max_ch=64, spectral_norm=True, do_self_attention=True, self_attention_hw=16, loss_architecture='wgan_gp', loss_scale=128.0)
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg, rcfg, tr, Pref, dev, ref = make(kw, sys.argv[1] if len(sys.argv) > 1 else 'fp16', seed=SEED, batch=2)
rcfg.sn_state = R.init_sn_state(Pref, seed=3)
sn0 = {k: v.float() for k, v in rcfg.sn_state.items()}


def reset():
  if rcfg.sn_cache:
    rcfg.sn_cache.clear()
  for k, v in sn0.items():
    rcfg.sn_state[k] = v.double()
    tr.store.state[k].copy_(v)


names = tr.store.names('d')
for which in ('no_gp',):
  reset()
  Q = {k: v.detach().clone().requires_grad_(True) for k, v in Pref.items()}
  rl, rterms = R.discriminator_loss(Q, ref['s'], ref['t'], rcfg, ref['a_s'], ref['a_t'])
  pick = lambda terms: sum(v for k, v in terms.items() if which == 'all' or (('gradient_penalty' in k) == (which == 'gp_only')))
  rg = R.grads_of(pick(rterms), Q, [k for k in names])
  reset()
  tr.store.zero_grad('d')
  tr._set_requires_grad(g=False, d=True)
  loss, terms = T.discriminator_loss(tr.P, dev['s'], dev['t'], cfg, dev['a_s'], dev['a_t'])
  (pick(terms) * cfg.loss_scale).backward()
  pggan.end_run(tr.P)
  torch.cuda.synchronize()
  hg = {k: v.double().cpu() / cfg.loss_scale for k, v in tr.store.grad_dict().items() if k in names}
  num = {k: float(((hg[k] - rg[k]) ** 2).sum()) for k in names}
  den = sum(float((rg[k] ** 2).sum()) for k in names)
  print('== %s: group rel-L2 %.4f' % (which, (sum(num.values()) / den) ** 0.5))
  for k in sorted(names, key=lambda k: -num[k])[:8]:
    print('   %-70s share %.3f  rel %.3f  |ref| %.3e' % (k, num[k] / sum(num.values()), (num[k] / max(float((rg[k] ** 2).sum()), 1e-300)) ** 0.5,
                                                       float(rg[k].norm())))

# the oracle's own deviation under the same storage rounding, per domain (non-GP terms)
from oracle import rounding      # noqa: E402
P0 = {k: v.detach() for k, v in Pref.items()}
sdt = torch.float16 if (len(sys.argv) < 2 or sys.argv[1] == 'fp16') else torch.bfloat16


def no_gp(Q):
  _, tt = R.discriminator_loss(Q, ref['s'], ref['t'], rcfg, ref['a_s'], ref['a_t'])
  return sum(v for k, v in tt.items() if 'gradient_penalty' not in k)


e, rnd, ex = rounding.gradient_sensitivity(P0, list(names), no_gp, sdt, reset=reset)
for dom in ('discriminator_s', 'discriminator_t'):
  ks = [k for k in names if k.startswith(dom)]
  m = (sum(float(((rnd[k] - ex[k]) ** 2).sum()) for k in ks) / sum(float((ex[k] ** 2).sum()) for k in ks)) ** 0.5
  h = (sum(num[k] for k in ks) / sum(float((rg[k] ** 2).sum()) for k in ks)) ** 0.5
  print('seed %d %s: kernels %.4f   rounding model %.4f' % (SEED, dom, h, m))
