"""Prints how far the HIP path is from the reference-computed full-size fixture (tests/golden/full_hw256_c256.json):
worst loss-term deviation, worst image-probe deviation, and the distribution of per-variable gradient-norm ratios,
for the fp32 and the bf16 path.  Same comparison as tests/test_gpu_model.py::test_full_width_stage_hits_the_reference."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import torch_ref as R      # noqa: E402  (weights are re-created from the fixture's seed)
from twingan_amd import Config         # noqa: E402
from twingan_amd import twingan as T   # noqa: E402

fix = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'full_hw256_c256.json')))
hw, batch = fix['config']['hw'], fix['batch']
P = R.init_params(R.Config(**fix['config']), seed=fix['param_seed'], dtype=torch.float64, std='he')
P = {k: v.float() for k, v in P.items()}
for precision in ('fp32', 'bf16'):
  cfg = Config(precision=precision, **fix['config'])
  tr = T.Trainer(cfg, device='cuda:0', seed=0)
  tr.store.load_state_dict(P)
  g = torch.Generator().manual_seed(fix['input_seed'])
  adt = torch.bfloat16 if precision == 'bf16' else torch.float32
  s = torch.rand(batch, hw, hw, 3, generator=g).to('cuda:0').to(adt)
  t = torch.rand(batch, hw, hw, 3, generator=g).to('cuda:0').to(adt)
  a_s = torch.tensor(fix['gp_alpha_s'], dtype=torch.float32, device='cuda:0')
  a_t = torch.tensor(fix['gp_alpha_t'], dtype=torch.float32, device='cuda:0')
  step = hw // 4
  with torch.no_grad():
    o = T.forward_generators(tr.P, s, t, cfg)
  probe = max(float((o[k][:, ::step, ::step, :].float().cpu() - torch.tensor(fix['probe'][k])).abs().max())
              for k in ('s_prime', 't_prime', 's_cycle', 't_cycle'))
  del o
  worst_term, ratios = 0.0, []
  for group, fn, args, want in (('g', T.generator_loss, (s, t, cfg), fix['g_terms']),
                                ('d', T.discriminator_loss, (s, t, cfg, a_s, a_t), fix['d_terms'])):
    tr.store.zero_grad(group)
    tr._set_requires_grad(g=group == 'g', d=group == 'd')
    loss, terms = fn(tr.P, *args)
    worst_term = max(worst_term, max(abs(v.item() - want[k]) / max(1.0, abs(want[k])) for k, v in terms.items()))
    loss.backward()
    gd = tr.store.grad_dict()
    names = tr.store.names(group)
    top = max(fix['grad_norm'][k] for k in names)
    ratios += [float(gd[k].double().norm()) / fix['grad_norm'][k] for k in names if fix['grad_norm'][k] > 1e-3 * top]
    del loss, terms, gd
  r = np.array(ratios)
  print('%s: worst loss term %.2e (rel. to max(1,|x|)), worst image probe %.2e, gradient-norm ratio over %d variables: '
        'median %.4f, min %.4f, max %.4f' % (precision, worst_term, probe, len(r), np.median(r), r.min(), r.max()),
        flush=True)
  del tr
  torch.cuda.empty_cache()
