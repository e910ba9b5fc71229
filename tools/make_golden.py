#!/usr/bin/env python
"""Generates tests/golden/*.npz: seeded input/output vectors of the TwinGAN hot path computed by the
float64 oracle (oracle/np_ops.py for the primitives, oracle/torch_ref.py in float64 for whole
networks, losses and gradients).

The reference itself (Python-2 / TF-1.8) cannot be imported here and ships no golden
vectors for this path, so these fixtures are *oracle-generated*: they freeze the restatement so that
(a) the oracle cannot drift silently and (b) the GPU parity tests have committed vectors to hit.
Parity stays "unpinned" in the sense of oracle/__init__.py.

Run:  python tools/make_golden.py      (rewrites tests/golden/; deterministic)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import np_ops as N          # noqa: E402
from oracle import torch_ref as R       # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def primitives():
  r = np.random.RandomState(7)
  d = {}
  # 3x3 SAME / 1x1 / 4x4 VALID convs + both gradients (TF layouts: NHWC, HWIO)
  for tag, (n, h, cin, cout, k, pad) in dict(c3=(2, 8, 16, 24, 3, 'SAME'), c1=(2, 8, 16, 8, 1, 'SAME'),
                                            c4=(3, 4, 8, 16, 4, 'VALID'), rgb=(2, 8, 3, 16, 1, 'SAME')).items():
    x = r.randn(n, h, h, cin)
    w = r.randn(k, k, cin, cout) * (2.0 / (k * k * cin)) ** 0.5
    y = N.conv2d(x, w, pad)
    gy = r.randn(*y.shape)
    d[tag + '_x'], d[tag + '_w'], d[tag + '_y'], d[tag + '_gy'] = x, w, y, gy
    d[tag + '_gx'] = N.conv2d_bwd_data(gy, w, (h, h), pad)
    d[tag + '_gw'] = N.conv2d_bwd_weight(x, gy, (k, k), pad)
  # instance norm -> lrelu -> pixel norm
  x = r.randn(2, 8, 8, 16) * 2.0 + 0.5
  gamma, beta = 1.0 + 0.1 * r.randn(16), 0.1 * r.randn(16)
  d['na_x'], d['na_gamma'], d['na_beta'] = x, gamma, beta
  d['na_z'] = N.pixel_norm(N.leaky_relu(N.instance_norm(x, gamma, beta)))
  d['na_z_nopn'] = N.leaky_relu(N.instance_norm(x, gamma, beta))
  d['na_z_rgb'] = N.instance_norm(x, gamma, beta)
  # resampling
  x = r.randn(2, 4, 4, 8)
  d['rs_x'], d['rs_up'], d['rs_pool'] = x, N.upsample2x(x), N.avg_pool2(N.upsample2x(x) + r.randn(2, 8, 8, 8) * 0 + 1.0)
  # minibatch stddev
  x = r.randn(4, 4, 4, 16)
  d['mb_x'], d['mb_y'] = x, N.minibatch_state_concat(x)
  # losses
  a, b = r.rand(2, 8, 8, 3), r.rand(2, 8, 8, 3)
  d['l_a'], d['l_b'] = a, b
  d['l_abs'] = np.array(N.absolute_difference(a, b, 0.7))
  g = r.randn(3, 8, 8, 3) * 0.1
  d['gp_g'], d['gp'] = g, np.array(N.gradient_penalty(g, 10.0))
  # Adam, TF form, three steps with a shared counter
  th, m, v = r.randn(64), np.zeros(64), np.zeros(64)
  d['adam_theta0'] = th.copy()
  gs = r.randn(3, 64)
  d['adam_g'] = gs
  for t in range(3):
    th, m, v = N.adam_step(th, gs[t], m, v, t + 1)
  d['adam_theta3'], d['adam_m3'], d['adam_v3'] = th, m, v
  return d


def model(hw, max_ch, batch, growing=False, alpha=0.0, seed=0, **extra):
  cfg = R.Config(hw=hw, max_ch=max_ch, is_growing=growing, alpha_grow=alpha, **extra)
  P = R.init_params(cfg, seed=seed, dtype=torch.float64, std='he')
  # round the parameters to fp32 so the GPU fp32 path starts from identical bits
  P = {k: v.float().double() for k, v in P.items()}
  g = torch.Generator().manual_seed(1234)
  s = torch.rand(batch, hw, hw, 3, generator=g).double()
  t = torch.rand(batch, hw, hw, 3, generator=g).double()
  a_s = torch.rand(batch, generator=g).double()
  a_t = torch.rand(batch, generator=g).double()
  d = {'in/sources': s.numpy(), 'in/targets': t.numpy(), 'in/gp_alpha_s': a_s.numpy(), 'in/gp_alpha_t': a_t.numpy()}
  if cfg.use_style_embedding:      # the random_style_embed draw of twingan.py:232-235 is an input of the fixture
    cfg.style_noise = torch.randn(batch, cfg.style_embed_size, generator=g).double()
    d['in/style_noise'] = cfg.style_noise.numpy()
  for k, v in P.items():
    d['param/' + k] = v.numpy()
  with torch.no_grad():
    o = R.forward_generators(P, s, t, cfg)
    for k in ('es', 's_prime', 't_prime', 's_cycle', 't_cycle'):
      d['fwd/' + k] = o[k].numpy()
    d['fwd/d_s_real'] = R.discriminator(P, s, cfg, 'discriminator_s')[0].numpy()
    d['fwd/d_t_prime'] = R.discriminator(P, o['t_prime'], cfg, 'discriminator_t')[0].numpy()
  for v in P.values():
    v.requires_grad_(True)
  gl, gterms = R.generator_loss(P, s, t, cfg)
  gg = R.grads_of(gl, P, R.generator_var_names(P))
  dl, dterms = R.discriminator_loss(P, s, t, cfg, a_s.reshape(-1, 1, 1, 1), a_t.reshape(-1, 1, 1, 1))
  dg = R.grads_of(dl, P, R.discriminator_var_names(P))
  d['loss/g_total'] = np.array(float(gl))
  d['loss/d_total'] = np.array(float(dl))
  for k, v in gterms.items():
    d['loss/g/' + k] = np.array(float(v))
  for k, v in dterms.items():
    d['loss/d/' + k] = np.array(float(v))
  for k, v in gg.items():
    d['grad/' + k] = v.detach().numpy()
  for k, v in dg.items():
    d['grad/' + k] = v.detach().numpy()
  return d


def main():
  os.makedirs(OUT, exist_ok=True)
  np.savez_compressed(os.path.join(OUT, 'primitives.npz'), **primitives())
  # 16x16 (no cycle-GAN term, twingan.py:466) and 64x64 at 8 channels (cycle-GAN term on), plus a growing stage
  np.savez_compressed(os.path.join(OUT, 'twingan_hw16_c8.npz'), **model(16, 8, 2))
  np.savez_compressed(os.path.join(OUT, 'twingan_hw64_c8.npz'), **model(64, 8, 2))
  np.savez_compressed(os.path.join(OUT, 'twingan_hw16_c8_growing.npz'), **model(16, 8, 2, growing=True, alpha=0.3))
  # option rows of SURVEY 8(a): hinge loss + equalized lr + residual shortcuts; batch norm; style embedding
  np.savez_compressed(os.path.join(OUT, 'twingan_hw16_c8_hinge_eqlr_res.npz'),
                      **model(16, 8, 2, loss='hinge', equalized=True, res_block=True))
  np.savez_compressed(os.path.join(OUT, 'twingan_hw16_c8_batch_norm.npz'), **model(16, 8, 2, norm='batch_norm'))
  np.savez_compressed(os.path.join(OUT, 'twingan_hw16_c8_style.npz'),
                      **model(16, 8, 2, use_style_embedding=True, style_embed_size=8))
  for f in sorted(os.listdir(OUT)):
    print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
  main()
