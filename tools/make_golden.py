#!/usr/bin/env python
"""Generates tests/golden/*.npz.

  * twingan_*.npz -- REFERENCE-generated: weights, inputs, random draws, generated images, every loss term and every
    gradient of one G+D step, computed by the reference's own graph-building code (twingan.py / image_generation.py /
    nets/pggan*.py / libs/* under /root/reference) executed on the TensorFlow-1.8 API stand-in of oracle/tf_shim
    (TensorFlow itself is not installed here; see oracle/tf_shim/core.py for exactly what that does and does not
    pin).  The float64 oracle must agree with those numbers to 1e-9 or nothing is written.
  * primitives.npz -- oracle-generated (oracle/np_ops.py): single ops (conv + both gradients, norm, resampling,
    minibatch stddev, losses, Adam) whose semantics are TensorFlow's, not the reference's.

Needs /root/reference, i.e. runs in the build container only; the tests read the committed files.
Run:  python tools/make_golden.py      (rewrites tests/golden/; deterministic)
"""
import os
import re
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
  """One model fixture, computed by the REFERENCE's own graph code (oracle/ref_runner.py: twingan.GanModel._clone_fn
  and everything it calls, executed on the TF stand-in of oracle/tf_shim) from seeded weights and inputs.  The random
  draws the reference makes (WGAN-GP / DRAGAN alphas, style noise) become inputs of the fixture.  Before anything is
  written the float64 oracle is checked against the same numbers (1e-9): a fixture is never frozen from a
  restatement that disagrees with the reference."""
  from oracle import ref_runner
  cfg = R.Config(hw=hw, max_ch=max_ch, is_growing=growing, alpha_grow=alpha, **extra)
  P = R.init_params(cfg, seed=seed, dtype=torch.float64, std='he')
  # round the parameters to fp32 so the GPU fp32 path starts from identical bits
  P = {k: v.float().double() for k, v in P.items()}
  state = {}
  if cfg.spectral_norm:
    state = {k: v.float().double() for k, v in R.init_sn_state(P, seed=seed + 1).items()}
  g = torch.Generator().manual_seed(1234)
  s = torch.rand(batch, hw, hw, 3, generator=g).double()
  t = torch.rand(batch, hw, hw, 3, generator=g).double()
  preset = {k: v.numpy() for k, v in list(P.items()) + list(state.items())}
  ref = ref_runner.run(ref_runner.flags_of(cfg), s.numpy(), t.numpy(), global_step=ref_runner.global_step_of(cfg),
                       seed=seed, preset=preset)
  created = set(ref['variables']) - {'global_step'}
  moving = {k for k in created if re.search(r'/(moving_mean|moving_variance)_[st]$', k)}      # oracle: cfg.bn_state
  assert created - moving == set(preset), (sorted(created - moving - set(preset)), sorted(set(preset) - created))
  draws = {}
  for n, v in ref['random']:
    draws.setdefault(n, []).append(v)
  d = {'in/sources': s.numpy(), 'in/targets': t.numpy()}
  a_s = a_t = noise_s = noise_t = None
  if 'alpha' in draws:
    d['in/gp_alpha_s'], d['in/gp_alpha_t'] = draws['alpha'][0].reshape(-1), draws['alpha'][1].reshape(-1)
    a_s, a_t = (torch.from_numpy(x).reshape(-1, 1, 1, 1) for x in (d['in/gp_alpha_s'], d['in/gp_alpha_t']))
  else:
    d['in/gp_alpha_s'] = d['in/gp_alpha_t'] = np.zeros(batch)
  if 'uniform' in draws:      # get_perturbed_batch, image_generation.py:441-449
    d['in/dragan_noise_s'], d['in/dragan_noise_t'] = draws['uniform']
    noise_s, noise_t = torch.from_numpy(draws['uniform'][0]), torch.from_numpy(draws['uniform'][1])
  if cfg.use_style_embedding:      # the random_style_embed draw of twingan.py:232-235
    d['in/style_noise'] = draws['random_style_embed'][0]
    cfg.style_noise = torch.from_numpy(d['in/style_noise'])
  for k, v in list(P.items()) + list(state.items()):
    d['param/' + k] = v.numpy()
  ep = ref['end_points']
  d['fwd/es'] = ep['encoded_source_content_before_classification']
  for k in ('s_prime', 't_prime', 's_cycle', 't_cycle'):
    d['fwd/' + k] = ep[k + '_output']
  d['fwd/d_s_real'] = ep['discriminator_real_s_prediction']
  d['fwd/d_t_prime'] = ep['discriminator_t_prime_prediction']
  d['loss/g_total'], d['loss/d_total'] = np.array(ref['g_loss']), np.array(ref['d_loss'])
  for grp in 'gd':
    for k, v in ref[grp + '_terms'].items():
      d['loss/%s/%s' % (grp, ref_runner.term_name(k))] = np.array(v)
  gnames, dnames = R.generator_var_names(P), R.discriminator_var_names(P)
  for k in gnames:
    d['grad/' + k] = ref['g_grads'].get(k, np.zeros(tuple(P[k].shape)))
  for k in dnames:
    d['grad/' + k] = ref['d_grads'].get(k, np.zeros(tuple(P[k].shape)))
  for k in state:
    d['state_after/' + k] = ref['state_after'][k]

  # ---- the pin: the float64 oracle against the reference's numbers ------------------------------------------
  if state:
    cfg.sn_state, cfg.sn_cache = {k: v.clone() for k, v in state.items()}, {}
  for v in P.values():
    v.requires_grad_(True)
  gl, gterms = R.generator_loss(P, s, t, cfg)
  gg = R.grads_of(gl, P, gnames)
  dl, dterms = R.discriminator_loss(P, s, t, cfg, a_s, a_t, noise_s, noise_t)
  dg = R.grads_of(dl, P, dnames)
  worst = max(abs(float(gl) - ref['g_loss']), abs(float(dl) - ref['d_loss']))
  for grp, terms in (('g', gterms), ('d', dterms)):
    assert {'loss/%s/%s' % (grp, k) for k in terms} == {k for k in d if k.startswith('loss/%s/' % grp)}, grp
    for k, v in terms.items():
      worst = max(worst, abs(float(v) - float(d['loss/%s/%s' % (grp, k)])))
  scale = max(float(np.abs(d['grad/' + k]).max()) for k in gnames + dnames)
  for k, v in list(gg.items()) + list(dg.items()):
    worst = max(worst, float(np.abs(v.detach().numpy() - d['grad/' + k]).max()) / scale)
  if state:
    R.end_run(cfg)
    for k in state:
      worst = max(worst, float(np.abs(cfg.sn_state[k].numpy() - d['state_after/' + k]).max()))
  assert worst < 1e-9, 'oracle disagrees with the reference: %g' % worst
  print('  oracle vs reference: max deviation %.1e (losses, loss terms, all gradients%s)'
        % (worst, ', spectral-norm u' if state else ''))
  return d


def clones(hw, max_ch, batch, n, seed=0):
  """Data-parallel fixture: n clones built by deployment/model_deploy.create_clones around GanModel._clone_fn, losses
  divided by n and gradients summed by optimize_clones (oracle/ref_runner.run_clones)."""
  from oracle import ref_runner
  cfg = R.Config(hw=hw, max_ch=max_ch)
  P = {k: v.float().double() for k, v in R.init_params(cfg, seed=seed, dtype=torch.float64, std='he').items()}
  g = torch.Generator().manual_seed(4321)
  batches = [(torch.rand(batch, hw, hw, 3, generator=g).double(), torch.rand(batch, hw, hw, 3, generator=g).double())
             for _ in range(n)]
  ref = ref_runner.run_clones(ref_runner.flags_of(cfg), [(s.numpy(), t.numpy()) for s, t in batches], seed=seed,
                              preset={k: v.numpy() for k, v in P.items()})
  d = {'param/' + k: v.numpy() for k, v in P.items()}
  d['loss/g_total'], d['loss/d_total'] = np.array(ref['g_loss']), np.array(ref['d_loss'])
  gnames, dnames = R.generator_var_names(P), R.discriminator_var_names(P)
  assert set(ref['g_grads']) == set(gnames) and set(ref['d_grads']) == set(dnames)      # image_generation.py:487-501
  for k in gnames:
    d['grad_g/' + k] = ref['g_grads'][k]
  for k in dnames:
    d['grad_d/' + k] = ref['d_grads'][k]
  for v in P.values():
    v.requires_grad_(True)
  worst, tot = 0.0, dict(g=0.0, d=0.0)
  acc = {k: 0.0 for k in P}
  for i, ((s, t), c) in enumerate(zip(batches, ref['clones'])):
    a = [x for nme, x in c['random'] if nme == 'alpha']
    d['clone%d/sources' % i], d['clone%d/targets' % i] = s.numpy(), t.numpy()
    d['clone%d/gp_alpha_s' % i], d['clone%d/gp_alpha_t' % i] = a[0].reshape(-1), a[1].reshape(-1)
    gl, _ = R.generator_loss(P, s, t, cfg)
    dl, _ = R.discriminator_loss(P, s, t, cfg, torch.from_numpy(a[0]), torch.from_numpy(a[1]))
    tot['g'] += float(gl) / n
    tot['d'] += float(dl) / n
    for k, v in list(R.grads_of(gl / n, P, gnames).items()) + list(R.grads_of(dl / n, P, dnames).items()):
      acc[k] = acc[k] + v
  worst = max(abs(tot['g'] - ref['g_loss']), abs(tot['d'] - ref['d_loss']))
  for k in gnames:
    worst = max(worst, float(np.abs(acc[k].numpy() - d['grad_g/' + k]).max()))
  for k in dnames:
    worst = max(worst, float(np.abs(acc[k].numpy() - d['grad_d/' + k]).max()))
  assert worst < 1e-9, worst
  print('  %d clones, oracle vs reference (model_deploy): max deviation %.1e' % (n, worst))
  return d


CASES = {      # fixture name -> (batch, oracle Config fields); tests/test_golden.py::MODELS mirrors the Config fields
  # 16x16 (no cycle-GAN term, twingan.py:466) and 64x64 at 8 channels (cycle-GAN term on), plus a growing stage
  'twingan_hw16_c8': (2, dict(hw=16, max_ch=8)),
  'twingan_hw64_c8': (2, dict(hw=64, max_ch=8)),
  'twingan_hw16_c8_growing': (2, dict(hw=16, max_ch=8, growing=True, alpha=0.3)),
  # option rows of SURVEY 8(a)
  'twingan_hw16_c8_hinge_eqlr_res': (2, dict(hw=16, max_ch=8, loss='hinge', equalized=True, res_block=True)),
  'twingan_hw16_c8_batch_norm': (2, dict(hw=16, max_ch=8, norm='batch_norm')),
  # batch 1: the reference's conditional instance norm multiplies [B,1,1,C] statistics by a [B,C] gamma without
  # reshaping it (libs/instance_norm.py:100-135), which only broadcasts as intended for one image
  'twingan_hw16_c8_style': (1, dict(hw=16, max_ch=8, use_style_embedding=True, style_embed_size=8)),
  # ... while its conditional BATCH norm (libs/batch_norm.py:403-424 reshapes the rows) works for any batch
  'twingan_hw16_c8_style_bn': (2, dict(hw=16, max_ch=8, use_style_embedding=True, style_embed_size=8, norm='batch_norm')),
  'twingan_hw16_c8_dragan': (2, dict(hw=16, max_ch=8, loss='dragan')),
  'twingan_hw16_c16_sn_att': (2, dict(hw=16, max_ch=16, spectral_norm=True, do_self_attention=True,
                                      self_attention_hw=8)),
}


def training(hw=16, max_ch=8, batch=2, n_runs=4, seed=0):
  """n_runs consecutive session.run(train_op) of the reference's training graph (oracle/ref_runner.run_training:
  clones, Adam from the flags, GanModel._add_optimization with its n_critic alternation) from seeded weights: the
  inputs / alphas of every run, the counters and losses it saw, and every variable afterwards."""
  from oracle import ref_runner
  cfg = R.Config(hw=hw, max_ch=max_ch, lr=1e-3)
  P = {k: v.float().double() for k, v in R.init_params(cfg, seed=seed, dtype=torch.float64, std='he').items()}
  g = torch.Generator().manual_seed(777)
  runs = [(torch.rand(batch, hw, hw, 3, generator=g).double(), torch.rand(batch, hw, hw, 3, generator=g).double())
          for _ in range(n_runs)]
  flags = dict(ref_runner.flags_of(cfg), learning_rate=cfg.lr, learning_rate_decay_type='fixed', optimizer='adam',
               adam_beta1=cfg.beta1, adam_beta2=cfg.beta2, opt_epsilon=cfg.adam_eps, n_critic=2)
  ref = ref_runner.run_training(flags, [(s.numpy(), t.numpy()) for s, t in runs], seed=seed,
                                preset={k: v.numpy() for k, v in P.items()})
  d = {'param/' + k: v.numpy().copy() for k, v in P.items()}      # train_step below updates P in place
  d['meta/lr'], d['meta/beta1'], d['meta/beta2'], d['meta/eps'] = (np.array(x) for x in (cfg.lr, cfg.beta1, cfg.beta2, cfg.adam_eps))
  for k in P:
    d['after/' + k] = ref['variables'][k]
  for k in ('beta1_power', 'beta2_power'):
    d['after_opt/' + k] = ref['variables'][k]
  opt = R.AdamState(P, cfg)
  for i, ((s, t), h) in enumerate(zip(runs, ref['history'])):
    a = [v for n, v in h['random'] if n == 'alpha']
    d['run%d/sources' % i], d['run%d/targets' % i] = s.numpy(), t.numpy()
    d['run%d/gp_alpha_s' % i], d['run%d/gp_alpha_t' % i] = a[0].reshape(-1), a[1].reshape(-1)
    d['run%d/counters' % i] = np.array([h['n_critic_counter'], h['global_step'], h['n_critic_counter_after'],
                                        h['global_step_after']])
    d['run%d/d_loss' % i] = np.array(h['train_tensor'])      # train_op = identity(discriminator_loss) for wgan
    out = R.train_step(P, opt, s, t, cfg, torch.from_numpy(a[0]), torch.from_numpy(a[1]), i)
    assert 'd_loss' not in out or abs(out['d_loss'] - h['train_tensor']) < 1e-9
  worst = max(float(np.abs(P[k].detach().numpy() - d['after/' + k]).max()) for k in P)
  assert worst < 1e-9, worst
  print('  %d training runs, oracle vs reference: max parameter deviation %.1e' % (n_runs, worst))
  return d


def full_size(hw=256, max_ch=256, batch=2, seed=0, **extra):
  """BASELINE.json's headline configuration (256x256, 256 channels) through the reference's own code, at full size.
  The weights (71 MB) are not stored: they are `R.init_params(cfg, seed, float64, 'he')` rounded to fp32 and the inputs
  come from a seeded generator, both re-created by the test; what is stored is what the reference computed -- every
  loss term, the L2 norm of every variable's gradient, a probe of every generated image.  (Takes a few minutes and
  ~20 GB: `python tools/make_golden.py --full`.)"""
  from oracle import ref_runner
  cfg = R.Config(hw=hw, max_ch=max_ch, **extra)
  P = {k: v.float().double() for k, v in R.init_params(cfg, seed=seed, dtype=torch.float64, std='he').items()}
  # ``extra`` (BASELINE configs[4]: spectral_norm, do_self_attention, self_attention_hw): the power-iteration vectors u are
  # R.init_sn_state(P, seed + 1) rounded to fp32, re-created by the test like the weights
  state = {k: v.float().double() for k, v in R.init_sn_state(P, seed=seed + 1).items()} if cfg.spectral_norm else {}
  g = torch.Generator().manual_seed(1234)
  s = torch.rand(batch, hw, hw, 3, generator=g).double()
  t = torch.rand(batch, hw, hw, 3, generator=g).double()
  ref = ref_runner.run(ref_runner.flags_of(cfg), s.numpy(), t.numpy(), seed=seed,
                       preset={k: v.numpy() for k, v in list(P.items()) + list(state.items())})
  assert set(ref['variables']) - {'global_step'} == set(P) | set(state)
  alphas = [v.reshape(-1).tolist() for n, v in ref['random'] if n == 'alpha']
  out = dict(config=dict(hw=hw, max_ch=max_ch, **extra), batch=batch, param_seed=seed, input_seed=1234,
             gp_alpha_s=alphas[0], gp_alpha_t=alphas[1], g_total=ref['g_loss'], d_total=ref['d_loss'],
             g_terms={ref_runner.term_name(k): v for k, v in ref['g_terms'].items()},
             d_terms={ref_runner.term_name(k): v for k, v in ref['d_terms'].items()})
  gnames, dnames = R.generator_var_names(P), R.discriminator_var_names(P)
  out['grad_norm'] = {k: float(np.linalg.norm(ref['g_grads'][k])) for k in gnames}
  out['grad_norm'].update({k: float(np.linalg.norm(ref['d_grads'][k])) for k in dnames})
  step = hw // 4
  out['probe'] = {k: ref['end_points'][k + '_output'][:, ::step, ::step, :].tolist()
                  for k in ('s_prime', 't_prime', 's_cycle', 't_cycle')}
  out['probe']['d_real_s'] = ref['end_points']['discriminator_real_s_prediction'].reshape(-1).tolist()
  out['probe']['d_t_prime'] = ref['end_points']['discriminator_t_prime_prediction'].reshape(-1).tolist()
  out['probe']['es_abs_mean'] = float(np.abs(ref['end_points']['encoded_source_content_before_classification']).mean())
  return out


SCHEMAS = {      # full-width configurations of BASELINE.json, run once through the reference just for its variables
  'hw256_c256': dict(hw=256, max_ch=256),
  'hw128_c256_growing': dict(hw=128, max_ch=256, is_growing=True, alpha_grow=0.5),
  'hw64_c256_sn_att_bn': dict(hw=64, max_ch=256, norm='batch_renorm', spectral_norm=True, do_self_attention=True,
                              self_attention_hw=32),
  'hw32_c128_eqlr_res': dict(hw=32, max_ch=128, equalized=True, res_block=True),
  'hw32_c64_style_bn': dict(hw=32, max_ch=64, use_style_embedding=True, style_embed_size=16, norm='batch_norm'),
  'hw64_c64_unet_max16': dict(hw=64, max_ch=64, unet_max_concat_hw=16),
  'hw64_c64_dis32': dict(hw=64, max_ch=64, max_ch_dis=32, res_block=True),
  'hw32_c32_sn_everywhere_res': dict(hw=32, max_ch=16, spectral_norm=True, sn_non_disc=True, res_block=True),
}


PGGAN_CASES = {      # BASELINE configs[0] (4x4 stage 0, batch 16) and an 8x8 stage; name -> (batch, oracle Config kwargs)
    'pggan_hw4_c16': (16, dict(hw=4, max_ch=16, norm='batch_norm')),
    'pggan_hw8_c16_in': (4, dict(hw=8, max_ch=16, norm='instance_norm')),
    'pggan_hw8_c16_hinge_grow': (4, dict(hw=8, max_ch=16, norm='batch_norm', loss='hinge', is_growing=True, alpha_grow=0.3)),
}


def pggan_model(batch, seed=0, **kw):
  """One fixture of the plain PGGAN trainer (image_generation.GanModel._clone_fn executed by
  oracle/ref_runner.run_pggan): seeded weights and targets, the reference's own noise and GP-alpha draws, every loss
  term, every gradient, the generated images.  Refuses to write unless the float64 oracle agrees to 1e-9."""
  from oracle import ref_runner
  cfg = R.Config(use_unet=False, **kw)
  P = {k: v.float().double() for k, v in R.init_pggan_params(cfg, seed=seed, dtype=torch.float64, std='he').items()}
  g = torch.Generator().manual_seed(4321)
  t = torch.rand(batch, cfg.hw, cfg.hw, 3, generator=g).double()
  flags = dict(train_image_size=cfg.hw, pggan_max_num_channels=cfg.max_ch, generator_norm_type=cfg.norm,
               loss_architecture=cfg.loss, is_growing=cfg.is_growing, max_number_of_steps=ref_runner.GROW_STEPS,
               grow_start_number_of_steps=0)
  ref = ref_runner.run_pggan(flags, t.numpy(), global_step=ref_runner.global_step_of(cfg), seed=seed,
                             preset={k: v.numpy() for k, v in P.items()})
  assert set(ref['trainable']) == set(P), set(ref['trainable']) ^ set(P)
  draws = {}
  for n, v in ref['random']:
    draws.setdefault(n, []).append(v)
  noise = torch.from_numpy(draws['normal'][0])
  alpha = torch.from_numpy(draws['alpha'][0]) if 'alpha' in draws else torch.zeros(batch, 1, 1, 1, dtype=torch.float64)
  d = {'in/targets': t.numpy(), 'in/noise': noise.numpy(), 'in/gp_alpha': alpha.numpy().reshape(-1)}
  for k, v in P.items():
    d['param/' + k] = v.numpy()
  d['fwd/generator_output'] = ref['end_points']['generator_output']
  d['fwd/d_real'] = ref['end_points']['discriminator_real_prediction']
  d['loss/g_total'], d['loss/d_total'] = np.array(ref['g_loss']), np.array(ref['d_loss'])
  for grp in 'gd':
    for k, v in ref[grp + '_terms'].items():
      d['loss/%s/%s' % (grp, k)] = np.array(v)
    for k, v in ref[grp + '_grads'].items():
      if (grp == 'd') == k.startswith('discriminator'):
        d['grad/' + k] = v
  # the oracle must reproduce all of it
  for v in P.values():
    v.requires_grad_(True)
  gl, gt = R.pggan_generator_loss(P, t, cfg, noise)
  dl, dt = R.pggan_discriminator_loss(P, t, cfg, noise, alpha)
  for k, v in list(gt.items()) + list(dt.items()):
    grp = 'g' if k in gt else 'd'
    assert abs(float(v) - float(d['loss/%s/%s' % (grp, k)])) < 1e-9, k
  grads = dict(R.grads_of(gl, P, [k for k in P if k.startswith('generator')]))
  grads.update(R.grads_of(dl, P, [k for k in P if k.startswith('discriminator')]))
  scale = max(float(np.abs(d['grad/' + k]).max()) for k in grads)
  for k, v in grads.items():
    assert np.abs(v.numpy() - d['grad/' + k]).max() < 1e-9 * scale, k
  return d


def infer_model(norm, hw=16, max_ch=8, batch=2, seed=5):
  """The inference branch (twingan.py:300-363: is_training=False, `sources_ph` / `targets_ph` ->
  custom_generated_t_style_source / custom_generated_s_style_target) computed by the reference's own code with fed
  placeholders and NON-trivial BatchNorm moving statistics preset into its variables."""
  from oracle import ref_runner
  cfg = R.Config(hw=hw, max_ch=max_ch, norm=norm)
  P = {k: v.float().double() for k, v in R.init_params(cfg, seed=seed, dtype=torch.float64, std='he').items()}
  rng = np.random.RandomState(seed)
  state = {}
  if norm != 'instance_norm':
    for k in list(P):
      if k.endswith(('/gamma_s', '/gamma_t')):
        base, d, c = k.rsplit('/', 1)[0], k[-2:], P[k].shape[0]
        state[base + '/moving_mean' + d] = torch.from_numpy(np.float32(rng.randn(c) * 0.3)).double()
        state[base + '/moving_variance' + d] = torch.from_numpy(np.float32(0.5 + rng.rand(c))).double()
  preset = {k: v.numpy() for k, v in list(P.items()) + list(state.items())}
  s, t = rng.rand(batch, hw, hw, 3), rng.rand(batch, hw, hw, 3)
  sp, tp = np.float32(rng.rand(batch, hw, hw, 3)).astype(np.float64), np.float32(rng.rand(batch, hw, hw, 3)).astype(np.float64)
  ref = ref_runner.run(ref_runner.flags_of(cfg), s, t, want_grads=False, preset=preset, feed={'sources_ph': sp, 'targets_ph': tp})
  d = {'in/sources_ph': sp, 'in/targets_ph': tp}
  for k, v in preset.items():
    d['param/' + k] = v
  for k in ('custom_generated_t_style_source', 'custom_generated_s_style_target'):
    d['out/' + k] = ref['custom'][k]
  cfg.bn_state = state
  assert np.abs(R.translate(P, torch.from_numpy(sp), cfg, 't').numpy() - d['out/custom_generated_t_style_source']).max() < 1e-9
  assert np.abs(R.translate(P, torch.from_numpy(tp), cfg, 's').numpy() - d['out/custom_generated_s_style_target']).max() < 1e-9
  return d


def main():
  os.makedirs(OUT, exist_ok=True)
  np.savez_compressed(os.path.join(OUT, 'primitives.npz'), **primitives())
  for name, (batch, kw) in CASES.items():
    print(name)
    kw = dict(kw)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **model(kw.pop('hw'), kw.pop('max_ch'), batch, **kw))
  np.savez_compressed(os.path.join(OUT, 'clones2_hw16_c8.npz'), **clones(16, 8, 2, 2))
  np.savez_compressed(os.path.join(OUT, 'train4_hw16_c8.npz'), **training())
  for name, (batch, kw) in PGGAN_CASES.items():
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **pggan_model(batch, **kw))
  for norm in ('instance_norm', 'batch_norm', 'batch_renorm'):
    np.savez_compressed(os.path.join(OUT, 'infer_hw16_c8_%s.npz' % norm), **infer_model(norm))
  # the variables the reference creates at full width, with what its initialisers drew (names, shapes, statistics)
  import json
  from oracle import ref_runner
  schema = {}
  for name, kw in SCHEMAS.items():
    cfg = R.Config(**kw)
    r = np.random.RandomState(0)
    ref = ref_runner.run(ref_runner.flags_of(cfg), r.rand(1, cfg.hw, cfg.hw, 3), r.rand(1, cfg.hw, cfg.hw, 3),
                         global_step=ref_runner.global_step_of(cfg), want_grads=False)
    schema[name] = dict(config=kw, variables={
      k: dict(shape=list(v.shape), trainable=k in ref['trainable'], mean=float(v.mean()), std=float(v.std()))
      for k, v in ref['variables'].items() if k != 'global_step'})
    print('schema', name, len(schema[name]['variables']), 'variables')
  with open(os.path.join(OUT, 'variable_schema.json'), 'w') as fh:
    json.dump(schema, fh, indent=0, sort_keys=True)
  # the progressive-growing stage driver (pggan_runner.py:82-160), executed the same way
  drv = []
  for args in ((4, 32, {4: 16, 8: 16, 16: 8, 32: 8}, 300000),
               (4, 256, {4: 16, 8: 16, 16: 16, 32: 16, 64: 12, 128: 12, 256: 12, 512: 6}, 300000),
               (8, 64, {8: 8, 16: 8, 32: 8, 64: 3}, 1000)):
    drv.append(dict(start_hw=args[0], max_hw=args[1], hw_to_batch_size={str(k): v for k, v in args[2].items()},
                    num_images_per_resolution=args[3], stages=ref_runner.run_stage_driver(*args)))
  with open(os.path.join(OUT, 'stage_driver.json'), 'w') as fh:
    json.dump(drv, fh, indent=1)
  for f in sorted(os.listdir(OUT)):
    print(f, os.path.getsize(os.path.join(OUT, f)))


PREPROCESS_CASES = [      # (h, w, resize_mode, is_training, seed): the trainer's image preprocessing on one decoded image
    (37, 53, 'PAD', True, 0), (64, 40, 'PAD', True, 1), (50, 50, 'PAD', True, 2), (37, 53, 'CROP', True, 3),
    (20, 33, 'RESHAPE', True, 4), (120, 90, 'PAD', True, 5), (31, 17, 'CROP', True, 6), (48, 48, 'PAD', False, 7),
    (9, 30, 'PAD', True, 8), (70, 64, 'PAD', True, 9)]


def preprocess_fixture(hw=32):
  """preprocessing/danbooru_preprocessing.preprocess_image executed on the TF stand-in (oracle/ref_runner.run_preprocess)
  for small random images: input, the random draws of the live branch, output."""
  from oracle import ref_runner
  rng = np.random.RandomState(17)
  out = {'hw': np.int64(hw)}
  for i, (h, w, mode, training, seed) in enumerate(PREPROCESS_CASES):
    img = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
    if i == 2:
      img[:, :, 1] = img[:, :, 0]      # some grey-ish and saturated pixels
      img[::3, ::2] = 255
    res, dr = ref_runner.run_preprocess(img, hw, mode, training, seed)
    applied = dict(dr['applied'])
    flip = bool(training and dr['flip_uniform'] < 0.5)
    sat_first = bool(training and dr['applied'][0][0] == 'saturation')
    mine = N.preprocess_image(img, hw, mode, training, flip=flip, saturation_first=sat_first,
                              delta=applied.get('brightness', 0.0), factor=applied.get('saturation', 1.0))
    assert np.abs(mine - res).max() < 1e-12, (i, np.abs(mine - res).max())
    out['img%d' % i] = img
    out['out%d' % i] = res
    out['par%d' % i] = np.array([float(flip), float(sat_first), applied.get('brightness', 0.0), applied.get('saturation', 1.0),
                                 float(training), float(dr['sel'] if training else -1)])
    out['mode%d' % i] = np.array(mode)
  return out


PREPROCESS_MODE_CASES = [      # (h, w, resize_mode, is_training, seed, do_random_cropping, color_space)
    (37, 53, 'RESHAPE', True, 0, True, 'rgb'),       # the reference's training recipe (docs/training.md:22-23)
    (64, 40, 'PAD', True, 1, True, 'yiq'), (50, 50, 'CROP', True, 2, True, 'bgr'), (31, 45, 'RESHAPE', True, 3, True, 'gray'),
    (90, 120, 'PAD', True, 4, True, 'rgb'), (48, 36, 'PAD', False, 5, True, 'rgb'),      # evaluation: the flag is ignored
    (37, 53, 'RANDOM_CROP', True, 6, False, 'rgb'), (64, 70, 'RANDOM_CROP', True, 7, True, 'rgb'),
    (20, 45, 'RANDOM_CROP', True, 8, True, 'rgb'), (50, 50, 'RANDOM_CROP', False, 9, False, 'bgr'),
    (32, 32, 'NONE', True, 10, False, 'rgb'), (33, 47, 'PAD', True, 11, False, 'yiq'), (40, 40, 'RESHAPE', False, 12, False, 'gray')]


def preprocess_modes_fixture(hw=32):
  """The reference's preprocess_image executed on the TF stand-in with --do_random_cropping, the RANDOM_CROP / NONE resize
  modes and the colour spaces: input, the draws of the live branch (flip, ordering, distortions, both crop rectangles),
  output; the float64 restatement must reproduce each before it is written."""
  from oracle import ref_runner
  rng = np.random.RandomState(23)
  out = {'hw': np.int64(hw), 'n': np.int64(len(PREPROCESS_MODE_CASES))}
  for i, (h, w, mode, training, seed, cropping, cs) in enumerate(PREPROCESS_MODE_CASES):
    img = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
    res, dr = ref_runner.run_preprocess(img, hw, mode, training, seed, do_random_cropping=cropping, color_space=cs)
    applied = dict(dr['applied'])
    flip = bool(training and dr['flip_uniform'] < 0.5)
    sat_first = bool(dr['applied'] and dr['applied'][0][0] == 'saturation')
    moff = None if dr['mode_crop'] is None else dr['mode_crop'][:2]
    mine = N.preprocess_image(img, hw, mode, training, flip=flip, saturation_first=sat_first,
                              delta=applied.get('brightness', 0.0), factor=applied.get('saturation', 1.0), crop=dr['crop'],
                              color_space=cs, mode_offset=moff)
    assert np.abs(mine - res).max() < 1e-12, (i, np.abs(mine - res).max())
    assert (dr['crop'] is not None) == (cropping and training)
    out['img%d' % i] = img
    out['out%d' % i] = res
    out['par%d' % i] = np.array([float(flip), float(sat_first), applied.get('brightness', 0.0), applied.get('saturation', 1.0),
                                 float(training), float(cropping)])
    out['crop%d' % i] = np.array(dr['crop'] if dr['crop'] is not None else (-1, -1, -1, -1), np.int64)
    out['moff%d' % i] = np.array(moff if moff is not None else (-1, -1), np.int64)
    out['mode%d' % i] = np.array(mode)
    out['cs%d' % i] = np.array(cs)
  return out


if __name__ == '__main__':
  if '--preprocess-modes' in sys.argv:      # only the fixture of the cropping / resize-mode / colour-space cases
    np.savez_compressed(os.path.join(OUT, 'preprocess_modes_hw32.npz'), **preprocess_modes_fixture())
    print('preprocess_modes_hw32.npz', os.path.getsize(os.path.join(OUT, 'preprocess_modes_hw32.npz')))
    sys.exit(0)
  if '--preprocess' in sys.argv:      # only the input-preprocessing fixture
    np.savez_compressed(os.path.join(OUT, 'preprocess_hw32.npz'), **preprocess_fixture())
    print('preprocess_hw32.npz', os.path.getsize(os.path.join(OUT, 'preprocess_hw32.npz')))
  elif '--infer' in sys.argv:      # only the inference-branch fixtures
    for norm in ('instance_norm', 'batch_norm', 'batch_renorm'):
      np.savez_compressed(os.path.join(OUT, 'infer_hw16_c8_%s.npz' % norm), **infer_model(norm))
      print('infer', norm)
  elif '--pggan' in sys.argv:      # only the plain-PGGAN fixtures
    for name, (batch, kw) in PGGAN_CASES.items():
      np.savez_compressed(os.path.join(OUT, name + '.npz'), **pggan_model(batch, **kw))
      print(name, os.path.getsize(os.path.join(OUT, name + '.npz')))
  elif '--full' in sys.argv:      # only a full-width fixture (slow); the default run leaves them untouched
    # --full [--hw 64|128|256]: BASELINE.json configs[1] / configs[2] / configs[3] at 256 channels, batch 2
    import json
    hw = int(sys.argv[sys.argv.index('--hw') + 1]) if '--hw' in sys.argv else 256
    # --sn --attention [--batch N]: BASELINE configs[4] (256 x 256 + self-attention at 64 x 64 + spectral-norm discriminators)
    extra = {}
    if '--sn' in sys.argv:
      extra['spectral_norm'] = True
    if '--attention' in sys.argv:
      extra.update(do_self_attention=True, self_attention_hw=64)
    batch = int(sys.argv[sys.argv.index('--batch') + 1]) if '--batch' in sys.argv else 2
    name = 'full_hw%d_c256%s%s.json' % (hw, '_sn' if '--sn' in sys.argv else '', '_att' if '--attention' in sys.argv else '')
    with open(os.path.join(OUT, name), 'w') as fh:
      json.dump(full_size(hw=hw, batch=batch, **extra), fh, indent=0, sort_keys=True)
    print(name, os.path.getsize(os.path.join(OUT, name)))
  else:
    main()
