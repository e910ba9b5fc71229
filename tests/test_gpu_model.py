"""GPU parity of the full networks, losses, gradients (incl. the WGAN-GP double backward) and the
alternating Adam step against the torch-CPU oracle on identical seeded inputs and weights.

fp32 path (direct kernels): network outputs rel-L2 <= 1e-5 / max|d| <= 1e-4, whole-model gradients rel-L2 <= 1e-2.
bf16 path (MFMA kernels, fp32 master weights): outputs rel-L2 <= 3e-2; whole-model gradients are only
checked directionally (cosine >= 0.9) because the graph is chaotic under bf16 storage rounding -- see
test_losses_and_gradients; per-primitive bf16 bounds are in test_gpu_ops.py.
"""
import dataclasses
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref as R       # noqa: E402  (checker only)


def rel_l2(a, b):
  a = a.detach().double().cpu().numpy()
  b = b.detach().double().cpu().numpy()
  assert a.shape == b.shape, (a.shape, b.shape)
  return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def make(cfg_kw, precision, seed=0, batch=3):
  from twingan_amd import Config
  from twingan_amd.twingan import Trainer
  cfg = Config(precision=precision, **cfg_kw)
  rcfg = R.Config(hw=cfg.hw, max_ch=cfg.max_ch, is_growing=cfg.is_growing, alpha_grow=cfg.alpha_grow,
                  use_unet=cfg.use_unet, equalized=cfg.equalized_learning_rate, res_block=cfg.use_res_block,
                  spectral_norm=cfg.spectral_norm, do_self_attention=cfg.do_self_attention,
                  self_attention_hw=cfg.self_attention_hw, loss=cfg.loss_architecture,
                  use_style_embedding=cfg.use_style_embedding, style_embed_size=cfg.style_embed_size,
                  unet_max_concat_hw=cfg.unet_max_concat_hw, sn_non_disc=cfg.spectral_norm_in_non_discriminator,
                  max_ch_dis=cfg.max_ch_dis, larger_rgb=cfg.use_larger_filter_at_rgb_layer)
  Pref = R.init_params(rcfg, seed=seed, dtype=torch.float64, std='he')
  tr = Trainer(cfg, device='cuda:0', seed=seed)
  tr.store.load_state_dict({k: v.float() for k, v in Pref.items()})
  Pref = {k: v.float().double() for k, v in Pref.items()}          # identical fp32-representable weights
  g = torch.Generator().manual_seed(1234 + seed)
  s = torch.rand(batch, cfg.hw, cfg.hw, 3, generator=g)
  t = torch.rand(batch, cfg.hw, cfg.hw, 3, generator=g)
  a_s, a_t = torch.rand(batch, generator=g), torch.rand(batch, generator=g)
  adt = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}[precision]
  if precision != 'fp32':
    s, t = s.to(adt).float(), t.to(adt).float()
  dev = dict(s=s.to('cuda:0').to(adt).contiguous(), t=t.to('cuda:0').to(adt).contiguous(), a_s=a_s.to('cuda:0'),
             a_t=a_t.to('cuda:0'))
  ref = dict(s=s.double(), t=t.double(), a_s=a_s.double().reshape(-1, 1, 1, 1), a_t=a_t.double().reshape(-1, 1, 1, 1))
  return cfg, rcfg, tr, Pref, dev, ref


def test_param_schema_matches_oracle():
  cfg, rcfg, tr, Pref, _, _ = make(dict(hw=16, max_ch=32), 'fp32')
  sd = tr.store.state_dict()
  assert set(sd) == set(Pref)
  for k in sd:
    assert tuple(sd[k].shape) == tuple(Pref[k].shape), k
  assert tr.P['discriminator_s/before_fc_1x1x32/Conv/weights'].shape == (3, 3, 40, 32)     # physical (padded) view


@pytest.mark.parametrize('growing', [False, True])
def test_networks_fp32_forward(growing):
  from twingan_amd import pggan
  kw = dict(hw=32, max_ch=32, is_growing=growing, alpha_grow=0.3 if growing else 0.0)
  cfg, rcfg, tr, Pref, dev, ref = make(kw, 'fp32')
  with torch.no_grad():
    net, ep = pggan.encoder_before_classification(tr.P, dev['s'], 's', cfg)
    rnet, rep = R.encoder(Pref, ref['s'], 's', rcfg)
    assert set(ep) == set(rep)
    for k in rep:
      e = rel_l2(ep[k], rep[k])
      assert e < 2e-5, ("encoder", k, e)
    out, gep = pggan.generator(tr.P, net, 't', cfg, ep)
    rout, rgep = R.generator(Pref, rnet, 't', rcfg, rep)
    assert set(k for k in gep if k != 'alpha_grow') == set(rgep)
    assert rel_l2(out, rout) < 2e-5
    assert float((out.double().cpu() - rout).abs().max()) < 1e-4
    pred, dep = pggan.discriminator(tr.P, out, cfg, 'discriminator_t')
    rpred, _ = R.discriminator(Pref, rout, rcfg, 'discriminator_t')
    assert rel_l2(pred, rpred) < 2e-5
    assert pred.shape == (3, 1)


FP32_GRAD_TOL = 5e-3         # whole-model fp32 gradients, aggregate rel-L2 over a group (measured 5e-7 .. 1.6e-3)
FP32_VAR_GRAD_TOL = 2e-2     # ... and every single variable of non-negligible norm (measured <= 5.7e-3)


def _grads_close(tr, Pref, names, tol, what, min_cos=None, var_tol=None):
  """Aggregate rel-L2 over the group <= tol, and -- ``var_tol`` -- every single variable whose reference gradient is
  not negligible (norm >= 1e-3 of the group's largest) within var_tol of it: a wrong gradient of one small tensor
  (a bias, a gamma, one operand of a paired filter gradient) cannot hide in the aggregate."""
  gd = tr.store.grad_dict()
  num = den = dot = na = 0.0
  worst = (0.0, None)
  ref = {k: (Pref[k].grad.numpy() if Pref[k].grad is not None else None) for k in names}
  top = max(np.linalg.norm(b) for b in ref.values() if b is not None)
  for k in names:
    a = gd[k].double().cpu().numpy()
    b = ref[k] if ref[k] is not None else np.zeros_like(a)
    num += np.sum((a - b) ** 2)
    den += np.sum(b ** 2)
    dot += np.sum(a * b)
    na += np.sum(a ** 2)
    e = np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-12)
    if e > worst[0] and np.linalg.norm(b) >= 1e-3 * top:
      worst = (e, k)
  tot = np.sqrt(num / (den + 1e-30))
  print('[grads] %s: aggregate rel-L2 %.3e, worst variable %s %.3e' % (what, tot, worst[1], worst[0]))
  assert tot < tol, '%s grads rel-L2 %.3e (worst %s %.3e)' % (what, tot, worst[1], worst[0])
  if var_tol is not None:
    assert worst[0] < var_tol, '%s: gradient of %s off by rel-L2 %.3e' % (what, worst[1], worst[0])
  if min_cos is not None:
    cos = dot / (np.sqrt(na * den) + 1e-30)
    assert cos > min_cos, '%s grads cosine %.4f' % (what, cos)


@pytest.mark.parametrize('precision,hw,max_ch', [('fp32', 16, 16), ('fp32', 64, 8), ('bf16', 32, 32), ('fp16', 32, 32)])
def test_losses_and_gradients(precision, hw, max_ch):
  from twingan_amd import twingan as T
  cfg, rcfg, tr, Pref, dev, ref = make(dict(hw=hw, max_ch=max_ch), precision, seed=2, batch=2)
  # fp32: the fp32 torch-CPU oracle itself sits 1e-3 from the fp64 one at 64x64 (GP double backward, IN
  # cancellations); the fp32 kernels (other summation orders, one-pass shifted statistics) measure 1e-6 at 16x16 and
  # 1.6e-3 (worst single variable 5.7e-3) at 64x64.  Every statistic of the forward pass is summed in a fixed order
  # (norm.hip wave_channel_accumulate), so these figures are reproducible run to run -- round 1 needed 8e-2 because
  # atomics-ordered statistics flipped a LeakyReLU unit near zero now and then.  bf16: this graph is chaotic under 2^-8 storage rounding -- the fp64 oracle
  # with bf16 rounding inserted at the same storage points (tools/bf16_sensitivity.py) moves the G
  # gradients by rel-L2 0.31 on this very case (0.21 from rounding the weights alone, 0.10 in fp16), and
  # the kernels reproduce that figure (0.32); per-primitive bf16 bounds are tight (test_gpu_ops.py).
  # So the whole-model bf16 check is directional: rel-L2 <= 0.5 and cosine >= 0.9.  fp16 (TG_F16, 11 significand bits):
  # the same kernels with the f16 MFMA instructions; the same directional bound, measured tighter (the sensitivity
  # tool's 0.10).
  ftol, gtol = (1e-4, FP32_GRAD_TOL) if precision == 'fp32' else (3e-2, 0.5)
  vtol = FP32_VAR_GRAD_TOL if precision == 'fp32' else None
  min_cos = None if precision == 'fp32' else 0.9
  for v in Pref.values():
    v.requires_grad_(True)
  # ---- generator loss
  tr.store.zero_grad('g')
  tr._set_requires_grad(g=True, d=False)
  gl, gterms = T.generator_loss(tr.P, dev['s'], dev['t'], cfg)
  rgl, rterms = R.generator_loss(Pref, ref['s'], ref['t'], rcfg)
  assert set(gterms) == set(rterms)
  for k in rterms:
    assert abs(gterms[k].item() - rterms[k].item()) < ftol * max(1.0, abs(rterms[k].item())) + (
        0 if precision == 'fp32' else 2e-2), (k, gterms[k].item(), rterms[k].item())
  gl.backward()
  rgl.backward()
  gptr = tr.store.grad['g'].data_ptr()
  assert tr.P['generator/block_4x4x%d/Conv/weights' % max_ch].grad.data_ptr() >= gptr     # flat buffer still in place
  _grads_close(tr, Pref, tr.store.names('g'), gtol, 'generator', min_cos, vtol)
  if precision != 'fp32':
    # ... and quantitatively: no further from the float64 gradients than the storage format itself puts a correct
    # implementation (oracle/rounding.py: the float64 oracle with this format's rounding at the kernels' storage points;
    # 0.31 for bf16, 0.10 for fp16 on these inputs) -- measured 0.32 / 0.09; bound 1.5 x the prediction + 0.02
    from oracle import rounding
    sdt = torch.bfloat16 if precision == 'bf16' else torch.float16
    pred, _, exact = rounding.generator_gradient_sensitivity({k: v.detach() for k, v in Pref.items()}, ref['s'], ref['t'], rcfg, sdt)
    gd = tr.store.grad_dict()
    num = sum(float(((gd[k].double().cpu() - exact[k]) ** 2).sum()) for k in exact)
    den = sum(float((exact[k] ** 2).sum()) for k in exact)
    err = (num / den) ** 0.5
    print('[sensitivity] %s generator gradients: kernels %.3f from the float64 oracle, storage rounding alone %.3f' % (precision, err, pred))
    assert err < 1.5 * pred + 0.02, (precision, err, pred)
  for v in Pref.values():
    v.grad = None
  # ---- discriminator loss (WGAN-GP double backward)
  tr.store.zero_grad('d')
  tr._set_requires_grad(g=False, d=True)
  dl, dterms = T.discriminator_loss(tr.P, dev['s'], dev['t'], cfg, dev['a_s'], dev['a_t'])
  rdl, rdterms = R.discriminator_loss(Pref, ref['s'], ref['t'], rcfg, ref['a_s'], ref['a_t'])
  assert set(dterms) == set(rdterms)
  for k in rdterms:
    assert abs(dterms[k].item() - rdterms[k].item()) < ftol * max(1.0, abs(rdterms[k].item())) + (
        0 if precision == 'fp32' else 5e-2 * abs(rdterms[k].item()) + 2e-2), (k, dterms[k].item(), rdterms[k].item())
  dl.backward()
  rdl.backward()
  _grads_close(tr, Pref, tr.store.names('d'), gtol, 'discriminator', min_cos, vtol)


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_gdrop_layer_in_the_discriminator_matches_oracle(precision, monkeypatch):
  """libs/gdrop.py:20-36 behind nets/pggan.py's do_dgrop argument (which no trainer of the reference sets: off by default
  here as there): with the SAME noise draws the discriminator's prediction and its input gradient land on the oracle's,
  whose gdrop is pinned to the reference's (tests/test_reference_live.py::test_gdrop_layer_matches_live_reference); the
  node's second derivative (the gradient penalty differentiates it twice): tests/test_gpu_ops.py::test_gdrop_op."""
  from twingan_amd import ops, pggan
  kw = dict(hw=16, max_ch=8, do_dgrop=True, gdrop_strength=0.3)
  cfg, rcfg, tr, Pref, dev, ref = make(kw, precision, seed=9, batch=3)
  rcfg.do_dgrop, rcfg.gdrop_strength = True, 0.3
  g = torch.Generator().manual_seed(77)
  noises = [torch.randn(3, c, generator=g) for c in (8, 8, 8, 8, 9, 8)]      # one discriminator call: 2 + 2 block convs, 2 tail convs
  feed = {'i': 0}
  real = ops.gdrop

  def fed(x, strength, noise=None, c_logical=None):
    nz = noises[feed['i']]
    feed['i'] += 1
    assert nz.shape[1] == (c_logical or x.shape[-1]), (nz.shape, x.shape, c_logical)
    pad = torch.zeros(nz.shape[0], x.shape[-1])      # the minibatch-stddev tensor is channel-padded: 9 -> 16
    pad[:, :nz.shape[1]] = nz
    return real(x, strength, noise=pad.to(x.device), c_logical=c_logical)
  monkeypatch.setattr(ops, 'gdrop', fed)
  tol = 2e-5 if precision == 'fp32' else 3e-2
  xin = dev['s'].clone().requires_grad_(True)
  pred, _ = pggan.discriminator(tr.P, xin, cfg, 'discriminator_s')
  assert feed['i'] == 6
  rcfg.gdrop_noise = [n.double() for n in noises[:6]]
  rx = ref['s'].clone().requires_grad_(True)
  rpred, _ = R.discriminator(Pref, rx, rcfg, 'discriminator_s')
  assert rel_l2(pred, rpred) < tol
  pred.sum().backward()
  rpred.sum().backward()
  assert rel_l2(xin.grad, rx.grad) < (5e-5 if precision == 'fp32' else 0.12)      # bf16: one rounding per layer through 8 convs and 6 gdrop layers, both ways


def test_use_gdrop_controller_and_identity(monkeypatch):
  """--use_gdrop: the `gdrop_strength` variable exists (checkpoint schema), training is unchanged (do_dgrop is never set by
  the trainers: twingan.py:861-867), and after a generator run past global step 100 the variable holds
  gdrop_coef * max(clip(generator_loss, 0, 1) - gdrop_lim, 0) ** gdrop_exp (image_generation.py:563-585; pinned to the
  reference by tests/test_reference_live.py)."""
  from twingan_amd import Config
  from twingan_amd.twingan import Trainer
  g = torch.Generator().manual_seed(3)
  s = torch.rand(2, 16, 16, 3, generator=g).to('cuda:0')
  t = torch.rand(2, 16, 16, 3, generator=g).to('cuda:0')
  al = torch.rand(2, generator=g).to('cuda:0')
  flats = {}
  for flag in (False, True):
    cfg = Config(hw=16, max_ch=8, precision='fp32', use_gdrop=flag, gdrop_lim=0.25)
    tr = Trainer(cfg, device='cuda:0', seed=4)
    tr.global_step = 150
    assert ('gdrop_strength' in tr.store.state_dict(include_state=True)) == flag
    out = tr.run(s, t, al, al)      # generator run
    if flag:
      want = 0.2 * max(min(max(float(out[0]), 0.0), 1.0) - 0.25, 0.0) ** 2.0
      got = float(tr.store.state['gdrop_strength'])
      assert want > 0 and abs(got - want) < 1e-6 * max(1.0, want), (got, want)
      assert tuple(tr.store.state_dict(include_state=True)['gdrop_strength'].shape) == ()
    tr.run(s, t, al, al)            # discriminator run
    torch.cuda.synchronize()
    flats[flag] = (tr.store.flat['g'].clone(), tr.store.flat['d'].clone())
  assert torch.equal(flats[True][0], flats[False][0]) and torch.equal(flats[True][1], flats[False][1])


@pytest.mark.parametrize('equalized,res_block,growing', [(True, False, False), (False, True, False), (True, True, True)])
def test_equalized_lr_and_res_block(equalized, res_block, growing):
  """--equalized_learning_rate (input scaling, N(0,1) weights; nets/pggan_utils.py:82-84,236-254) and --use_res_block
  (identity / 1x1 'shortcut' residuals; :257-264,334-342): schema, forward, losses and gradients vs the oracle,
  including the WGAN-GP double backward through the scaled / residual layers."""
  from twingan_amd import pggan
  from twingan_amd import twingan as T
  kw = dict(hw=32, max_ch=16, equalized_learning_rate=equalized, use_res_block=res_block, is_growing=growing,
            alpha_grow=0.4 if growing else 0.0)
  cfg, rcfg, tr, Pref, dev, ref = make(kw, 'fp32', seed=5, batch=2)
  assert set(tr.store.state_dict()) == set(Pref)
  if res_block:
    assert 'generator/block_8x8x16/shortcut/weights' in Pref and 'discriminator_s/from_rgb_32x32/shortcut/biases' in Pref
  with torch.no_grad():
    net, ep = pggan.encoder_before_classification(tr.P, dev['s'], 's', cfg)
    rnet, rep = R.encoder(Pref, ref['s'], 's', rcfg)
    for k in rep:
      assert rel_l2(ep[k], rep[k]) < 2e-5, k
    out, _ = pggan.generator(tr.P, net, 't', cfg, ep)
    rout, _ = R.generator(Pref, rnet, 't', rcfg, rep)
    assert rel_l2(out, rout) < 2e-5
    pred, _ = pggan.discriminator(tr.P, out, cfg, 'discriminator_t')
    rpred, _ = R.discriminator(Pref, rout, rcfg, 'discriminator_t')
    assert rel_l2(pred, rpred) < 2e-5
  for v in Pref.values():
    v.requires_grad_(True)
  tr.store.zero_grad('g')
  tr._set_requires_grad(g=True, d=False)
  gl, gterms = T.generator_loss(tr.P, dev['s'], dev['t'], cfg)
  rgl, rterms = R.generator_loss(Pref, ref['s'], ref['t'], rcfg)
  for k in rterms:
    assert abs(gterms[k].item() - rterms[k].item()) < 1e-4 * max(1.0, abs(rterms[k].item())), k
  gl.backward()
  rgl.backward()
  _grads_close(tr, Pref, tr.store.names('g'), FP32_GRAD_TOL, 'generator', var_tol=FP32_VAR_GRAD_TOL)
  for v in Pref.values():
    v.grad = None
  tr.store.zero_grad('d')
  tr._set_requires_grad(g=False, d=True)
  dl, dterms = T.discriminator_loss(tr.P, dev['s'], dev['t'], cfg, dev['a_s'], dev['a_t'])
  rdl, rdterms = R.discriminator_loss(Pref, ref['s'], ref['t'], rcfg, ref['a_s'], ref['a_t'])
  for k in rdterms:
    assert abs(dterms[k].item() - rdterms[k].item()) < 1e-4 * max(1.0, abs(rdterms[k].item())), k
  dl.backward()
  rdl.backward()
  _grads_close(tr, Pref, tr.store.names('d'), FP32_GRAD_TOL, 'discriminator', var_tol=FP32_VAR_GRAD_TOL)


def test_equalized_res_block_bf16_graph_step_runs():
  """The same options on the production path (bf16 MFMA kernels, hipGraph): finite losses, parameters move."""
  from twingan_amd import Config
  from twingan_amd.twingan import Trainer
  cfg = Config(hw=32, max_ch=32, equalized_learning_rate=True, use_res_block=True)
  tr = Trainer(cfg, device='cuda:0', seed=1, use_graph=True)
  g = torch.Generator().manual_seed(3)
  s = torch.rand(4, 32, 32, 3, generator=g).to('cuda:0').bfloat16()
  t = torch.rand(4, 32, 32, 3, generator=g).to('cuda:0').bfloat16()
  before = tr.store.flat['g'].clone()
  for _ in range(6):
    loss, terms = tr.run(s, t)
    assert np.isfinite(float(loss)) and all(np.isfinite(float(v)) for v in terms.values())
  assert float((tr.store.flat['g'] - before).abs().max()) > 0


def test_alternating_train_steps_match_oracle():
  """Four session.run equivalents (G, D, G, D) with the shared Adam counter (image_generation.py:640-652)."""
  cfg, rcfg, tr, Pref, dev, ref = make(dict(hw=16, max_ch=16), 'fp32', seed=3, batch=2)
  opt = R.AdamState(Pref, rcfg)
  for i in range(4):
    tr.run(dev['s'], dev['t'], dev['a_s'], dev['a_t'])
    R.train_step(Pref, opt, ref['s'], ref['t'], rcfg, ref['a_s'], ref['a_t'], counter=i)
  assert tr.adam_t == opt.t == 4 and tr.global_step == 2 and tr.n_critic_counter == 4
  sd = tr.store.state_dict()
  # Adam's first steps move every weight by ~lr regardless of gradient scale, so compare the UPDATE
  upd_err = []
  num = den = 0.0
  P0 = R.init_params(rcfg, seed=3, dtype=torch.float64, std='he')
  for k in sd:
    d_dev = sd[k].double().cpu() - P0[k].float().double()
    d_ref = Pref[k].detach() - P0[k].float().double()
    num += float(((d_dev - d_ref) ** 2).sum())
    den += float((d_ref ** 2).sum())
    if float(d_ref.abs().max()) > 0:
      upd_err.append(float((d_dev - d_ref).norm() / (d_ref.norm() + 1e-30)))
  # Adam's first steps are sign-like, so a handful of near-zero gradients may flip; bound the aggregate
  assert np.sqrt(num / den) < 5e-2, np.sqrt(num / den)
  assert np.median(upd_err) < 3e-2, np.median(upd_err)


@pytest.mark.parametrize('norm', ['instance_norm', 'batch_norm'])
def test_graph_replay_matches_eager_steps(norm):
  """hipGraph-captured steps (Trainer(use_graph=True)) against eagerly launched ones, run for run: same kernels,
  same order, so parameters agree to atomics-reordering noise -- and the capture's eager warm-up leaves no trace:
  counters, Adam step and moments, BatchNorm moving statistics are those of the eager trajectory.  'wgan' so no
  device RNG is involved."""
  from twingan_amd import Config
  from twingan_amd.twingan import Trainer
  cfg = Config(hw=32, max_ch=16, precision='fp32', loss_architecture='wgan', generator_norm_type=norm)
  g = torch.Generator().manual_seed(9)
  s = torch.rand(2, 32, 32, 3, generator=g).to('cuda:0')
  t = torch.rand(2, 32, 32, 3, generator=g).to('cuda:0')
  a = Trainer(cfg, device='cuda:0', seed=4)
  b = Trainer(cfg, device='cuda:0', seed=4, use_graph=True)
  la = lb = None
  for i in range(5):               # graph trainer: first call = warm-up (undone) + capture + 1 replay
    la, _ = a.run(s, t)
    lb, _ = b.run(s, t)
    if i == 0:
      assert b.use_graph and b.graph_fallback_reason is None, b.graph_fallback_reason
      assert abs(la.item() - lb.item()) < 1e-5 * max(1.0, abs(la.item())), 'the first captured step saw warmed-up weights'
  torch.cuda.synchronize()
  assert (a.adam_t, a.n_critic_counter, a.global_step) == (b.adam_t, b.n_critic_counter, b.global_step) == (5, 5, 2)
  assert int(b._adam_step_dev.item()) == int(a._adam_step_dev.item()) == 5
  sa, sb = a.store.state_dict(include_state=True), b.store.state_dict(include_state=True)
  num = sum(float(((sa[k] - sb[k]).double() ** 2).sum()) for k in sa)
  den = sum(float((sa[k].double() ** 2).sum()) for k in sa)
  assert (num / den) ** 0.5 < 5e-3, (num / den) ** 0.5      # Adam's sign-like first steps amplify atomics-order noise
  for grp in ('g', 'd'):           # Adam moments: the warm-up steps must not have entered them
    ma, mb = a.store.m[grp], b.store.m[grp]
    assert float((ma - mb).norm() / (ma.norm() + 1e-30)) < 5e-2
  if norm == 'batch_norm':
    k = 'generator/block_4x4x16/Conv/BatchNorm/moving_mean_s'
    assert float((a.store.state[k] - b.store.state[k]).abs().max()) < 1e-4 * max(1.0, float(a.store.state[k].abs().max()))
  assert abs(la.item() - lb.item()) < 1e-2 * max(1.0, abs(la.item()))


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_segmented_backward_leaves_the_same_gradients(precision):
  """The data-parallel schedule (Trainer(overlap=True)): the backward is cut at cfg.overlap_cut_hw, each segment
  completes one contiguous range of the flat gradient buffer (whose all-reduce then overlaps the next segment).  Every
  variable's gradient must be what the plain loss.backward() leaves -- same kernels on the same tensors; only the
  order in which contributions reach a gradient sink differs."""
  from twingan_amd import Config
  from twingan_amd.twingan import Trainer
  cfg = Config(hw=64, max_ch=16, precision=precision, overlap_cut_hw=16)
  adt = torch.bfloat16 if precision == 'bf16' else torch.float32
  g = torch.Generator().manual_seed(21)
  s = torch.rand(3, 64, 64, 3, generator=g).to('cuda:0').to(adt)
  t = torch.rand(3, 64, 64, 3, generator=g).to('cuda:0').to(adt)
  al = torch.rand(3, generator=g).to('cuda:0')
  plain = Trainer(cfg, device='cuda:0', seed=11, overlap=False)
  cut = Trainer(cfg, device='cuda:0', seed=11, overlap=True)
  assert (plain._nseg('g'), plain._nseg('d')) == (1, 1) and (cut._nseg('g'), cut._nseg('d')) == (3, 2)
  for grp in ('g', 'd'):
    seen = []
    for tr in (plain, cut):
      segs = [seg for seg, _ in tr._grad_segments(grp, s, t, al, al)]
      seen.append(segs)
      torch.cuda.synchronize()
    assert seen == [[0], list(range(cut._nseg(grp)))]
    ga, gb = plain.store.grad_dict(), cut.store.grad_dict()
    top = max(float(ga[k].norm()) for k in plain.store.names(grp))
    for k in plain.store.names(grp):
      na = float(ga[k].norm())
      if na < 1e-4 * top:
        continue
      e = float((ga[k] - gb[k]).norm()) / na
      assert e < (1e-4 if precision == 'fp32' else 2e-3), (grp, k, e)
    # the ranges are what the segments completed: phase p lives in [lo, hi) of the flat buffer
    for k in cut.store.names(grp):
      lo, hi = cut.store.phase_bounds[grp][cut.store.phase[k]]
      assert lo <= cut.store.offsets[k] < hi


def test_unpool_backward_data_leaves_the_same_gradients(monkeypatch):
  """A schedule change of the backward that must not change a gradient: in a generator step the backward-data of every
  discriminator block end reads the pooled gradient + sign bytes itself (tg_conv2d_bwd_data_unpool) instead of a
  full-resolution gradient tensor written by tg_lrelu_pool_bwd_signs -- bit for bit per launch."""
  from twingan_amd import Config, ops
  from twingan_amd.twingan import Trainer
  cfg = Config(hw=64, max_ch=64, precision='bf16')
  g = torch.Generator().manual_seed(29)
  s = torch.rand(2, 64, 64, 3, generator=g).to('cuda:0').bfloat16()
  t = torch.rand(2, 64, 64, 3, generator=g).to('cuda:0').bfloat16()
  al = torch.rand(2, generator=g).to('cuda:0')
  calls = []
  real = ops.conv_bwd_data_unpool_raw

  def counted(*a, **k):
    out = real(*a, **k)
    calls.append(out is not None)
    return out
  monkeypatch.setattr(ops, 'conv_bwd_data_unpool_raw', counted)
  grads = {}
  for name, unpool in (('base', False), ('new', True)):
    monkeypatch.setattr(ops, 'USE_DGRAD_UNPOOL', unpool)
    tr = Trainer(cfg, device='cuda:0', seed=13, overlap=False)
    grads[name] = {}
    for grp in ('g', 'd'):
      for _ in tr._grad_segments(grp, s, t, al, al):
        pass
      torch.cuda.synchronize()
      gd = tr.store.grad_dict()
      grads[name][grp] = {k: gd[k].clone() for k in tr.store.names(grp)}
    if unpool:      # the 64 x 64, 32 x 32 and 16 x 16 block ends of both discriminators, in the generator step only
      assert sum(calls) >= 6, calls
    else:
      assert not calls
  for grp in ('g', 'd'):
    top = max(float(v.norm()) for v in grads['base'][grp].values())
    for k, a in grads['base'][grp].items():
      na = float(a.norm())
      if na < 1e-4 * top:
        continue
      e = float((a - grads['new'][grp][k]).norm()) / na
      assert e < 2e-3, (grp, k, e)


SEGMENT_VARIANTS = {
    'sn': dict(hw=64, max_ch=16, spectral_norm=True, overlap_cut_hw=16),
    'sn_everywhere_att': dict(hw=64, max_ch=16, spectral_norm=True, spectral_norm_in_non_discriminator=True, do_self_attention=True,
                              self_attention_hw=32, overlap_cut_hw=16),
    'growing': dict(hw=64, max_ch=16, is_growing=True, alpha_grow=0.3, overlap_cut_hw=16),
    'growing_cut_at_half': dict(hw=32, max_ch=16, is_growing=True, alpha_grow=0.6, overlap_cut_hw=16),
    'style': dict(hw=64, max_ch=16, use_style_embedding=True, style_embed_size=8, overlap_cut_hw=16),
    'batch_norm_hinge': dict(hw=64, max_ch=16, generator_norm_type='batch_norm', loss_architecture='hinge', overlap_cut_hw=16),
}


@pytest.mark.parametrize('variant', sorted(SEGMENT_VARIANTS))
def test_segmented_backward_for_every_configuration(variant):
  """The overlapped clone all-reduce (deployment/model_deploy.py:473-503 sums the clones' gradients of EVERY configuration)
  needs the segmented backward everywhere: spectral norm (each normalised kernel read through a per-run leaf, sent through
  the power iteration's backward once, at the end of the segment that completes it), growing stages (the interpolated
  skip end-point a leaf of the high segment, the shrink path's variables in the high phase), the style encoder (whole in
  segment 0).  For each: the segmented backward leaves the plain backward's gradients, AND at the end of segment p the
  flat-buffer range of phase p already holds its final values (what the all-reduce started there would send)."""
  from twingan_amd import Config
  from twingan_amd.twingan import Trainer
  cfg = Config(precision='fp32', **SEGMENT_VARIANTS[variant])
  g = torch.Generator().manual_seed(23)
  hw = cfg.hw
  s = torch.rand(3, hw, hw, 3, generator=g).to('cuda:0')
  t = torch.rand(3, hw, hw, 3, generator=g).to('cuda:0')
  al = torch.rand(3, generator=g).to('cuda:0')
  plain = Trainer(cfg, device='cuda:0', seed=11, overlap=False)
  cut = Trainer(cfg, device='cuda:0', seed=11, overlap=True)
  assert cut.split and (cut._nseg('g'), cut._nseg('d')) == (3, 2)
  for grp in ('g', 'd'):
    torch.manual_seed(5)      # the style noise / device draws of both trainers
    for _ in plain._grad_segments(grp, s, t, al, al):
      pass
    torch.cuda.synchronize()
    ga = plain.store.grad_dict()
    torch.manual_seed(5)
    at_end = {}
    for seg, _ in cut._grad_segments(grp, s, t, al, al):
      torch.cuda.synchronize()
      lo, hi = cut.store.phase_bounds[grp][seg]
      at_end[seg] = cut.store.grad[grp][lo:hi].clone()
    torch.cuda.synchronize()
    gb = cut.store.grad_dict()
    top = max(float(ga[k].norm()) for k in plain.store.names(grp))
    for k in plain.store.names(grp):
      na = float(ga[k].norm())
      if na < 1e-4 * top:
        continue
      e = float((ga[k] - gb[k]).norm()) / na
      assert e < 2e-4, (variant, grp, k, e)
    for seg, snap in at_end.items():      # nothing arrived in a range after its segment ended
      lo, hi = cut.store.phase_bounds[grp][seg]
      assert torch.equal(snap, cut.store.grad[grp][lo:hi]), (variant, grp, seg)
  plain.close()
  cut.close()


def test_distillation_extras_under_graph_replay():
  """--do_encoder_distillation (twingan.py:162-177,507-521) with hipGraph replay: the datasets' embeddings live in static
  buffers the captured generator step reads; a graph trajectory with CHANGING embeddings equals the eager one (deterministic
  mode: bit for bit), and a run that brings other fields than the capture is refused."""
  from twingan_amd import Config
  from twingan_amd.twingan import Trainer
  cfg = Config(hw=16, max_ch=16, precision='bf16', do_encoder_distillation=True, distill_embed_dim=6, distillation_weight=0.5)
  g = torch.Generator().manual_seed(31)
  s = torch.rand(4, 16, 16, 3, generator=g).to('cuda:0').bfloat16()
  t = torch.rand(4, 16, 16, 3, generator=g).to('cuda:0').bfloat16()
  embs = [torch.randn(4, 6, generator=g).to('cuda:0') for _ in range(6)]
  out = {}
  with _Deterministic():
    for graph in (False, True):
      tr = Trainer(cfg, device='cuda:0', seed=3, use_graph=graph)
      torch.manual_seed(9)
      losses = []
      for i in range(6):
        loss, terms = tr.run(s, t, distill_embed_s=embs[i])
        if i % 2 == 0:
          assert 'l_source_distillation' in terms
          losses.append(float(terms['l_source_distillation']))
      assert not graph or tr.graph_fallback_reason is None, tr.graph_fallback_reason
      out[graph] = (losses, {k: v.clone() for k, v in tr.store.state_dict().items()})
      if graph:
        with pytest.raises(ValueError, match='dataset fields'):
          tr.run(s, t, distill_embed_t=embs[0])
      tr.close()
  assert len(set(out[True][0])) > 1      # the replayed graph saw the new embeddings
  assert out[True][0] == out[False][0]
  bad = [k for k in out[True][1] if not torch.equal(out[True][1][k], out[False][1][k])]
  assert not bad, bad[:5]


def test_graph_replay_wgan_gp_bf16_runs():
  """The north-star configuration's shape (bf16, WGAN-GP, device-drawn alphas) under graph replay."""
  from twingan_amd import Config
  from twingan_amd.twingan import Trainer
  cfg = Config(hw=64, max_ch=32, precision='bf16')
  g = torch.Generator().manual_seed(10)
  s = torch.rand(2, 64, 64, 3, generator=g).to('cuda:0').to(torch.bfloat16)
  t = torch.rand(2, 64, 64, 3, generator=g).to('cuda:0').to(torch.bfloat16)
  tr = Trainer(cfg, device='cuda:0', seed=5, use_graph=True)
  p0 = tr.store.flat['d'].clone()
  losses = [tr.run(s, t)[0].item() for _ in range(6)]
  assert all(np.isfinite(losses)), losses
  assert tr.use_graph and tr.adam_t == 6 and int(tr._adam_step_dev.item()) == 6      # the capture warm-up is undone
  assert float((tr.store.flat['d'] - p0).abs().max()) > 0
  assert bool(torch.isfinite(tr.store.flat['g']).all()) and bool(torch.isfinite(tr.store.flat['d']).all())


def test_unet_max_concat_hw_matches_oracle():
  """--pggan_unet_max_concat_hw (nets/pggan_utils.py:287-289): generator blocks above that resolution get no encoder
  skip (and their first conv no skip channels); outputs and generator gradients against the oracle, fp32."""
  cfg, rcfg, tr, Pref, dev, ref = make(dict(hw=32, max_ch=16, unet_max_concat_hw=8), 'fp32', seed=5, batch=2)
  from twingan_amd import twingan as T
  assert tr.P['generator/block_8x8x16/Conv/weights'].shape[2] == 32      # 16 + 16 skip channels
  assert tr.P['generator/block_16x16x16/Conv/weights'].shape[2] == 16     # no skip above 8x8
  with torch.no_grad():
    o = T.forward_generators(tr.P, dev['s'], dev['t'], cfg)
    oref = R.forward_generators(Pref, ref['s'], ref['t'], rcfg)
  for k in ('s_prime', 't_cycle'):
    assert rel_l2(o[k], oref[k]) < 2e-5, k
  tr.store.zero_grad('g')
  tr._set_requires_grad(g=True, d=False)
  loss, _ = T.generator_loss(tr.P, dev['s'], dev['t'], cfg)
  loss.backward()
  for v in Pref.values():
    v.requires_grad_(True)
  lref, _ = R.generator_loss(Pref, ref['s'], ref['t'], rcfg)
  assert abs(loss.item() - float(lref)) < 1e-4 * max(1.0, abs(float(lref)))
  gref = R.grads_of(lref, Pref, R.generator_var_names(Pref))
  gd = tr.store.grad_dict()
  num = sum(float(((gd[k].double().cpu() - gref[k]) ** 2).sum()) for k in gref)
  den = sum(float((gref[k] ** 2).sum()) for k in gref)
  assert (num / den) ** 0.5 < 2e-2, (num / den) ** 0.5


def test_discriminator_max_channels_matches_oracle():
  """--pggan_max_num_channels_dis (nets/pggan.py:54-56; pggan_utils.py:375-380): the discriminators get their own channel
  cap (schedule, before_fc scope name, minibatch-stddev pad, prediction FC); D loss terms and gradients incl. the
  gradient penalty against the oracle (pinned against the reference for this flag: tests/test_reference_live.py)."""
  cfg, rcfg, tr, Pref, dev, ref = make(dict(hw=32, max_ch=32, max_ch_dis=16), 'fp32', seed=9, batch=2)
  from twingan_amd import twingan as T
  assert 'discriminator_s/before_fc_1x1x16/Conv/weights' in tr.store.specs
  assert tr.store.specs['discriminator_t/prediction/fully_connected/weights']['shape'] == (16, 1)
  tr.store.zero_grad('d')
  tr._set_requires_grad(g=False, d=True)
  dl, dterms = T.discriminator_loss(tr.P, dev['s'], dev['t'], cfg, dev['a_s'], dev['a_t'])
  for v in Pref.values():
    v.requires_grad_(True)
  rdl, rterms = R.discriminator_loss(Pref, ref['s'], ref['t'], rcfg, ref['a_s'], ref['a_t'])
  for k in rterms:
    assert abs(dterms[k].item() - rterms[k].item()) < 1e-4 * max(1.0, abs(rterms[k].item())), k
  dl.backward()
  rdl.backward()
  _grads_close(tr, Pref, tr.store.names('d'), FP32_GRAD_TOL, 'discriminator', var_tol=FP32_VAR_GRAD_TOL)


def test_full_size_properties_256_bf16():
  """BASELINE config (256x256, max_ch 256) through size-independent properties: shapes, finiteness,
  instance-norm statistics of a generator block, pixel-norm unit RMS, D linearity of the GP ones-vector."""
  from twingan_amd import Config, pggan
  from twingan_amd.twingan import Trainer
  cfg = Config(hw=256, max_ch=256, precision='bf16')
  tr = Trainer(cfg, device='cuda:0', seed=0)
  g = torch.Generator().manual_seed(5)
  s = torch.rand(2, 256, 256, 3, generator=g).to('cuda:0').to(torch.bfloat16)
  with torch.no_grad():
    net, ep = pggan.encoder_before_classification(tr.P, s, 's', cfg)
    assert net.shape == (2, 4, 4, 256)
    assert ep['encoder_block_256x256x32'].shape == (2, 256, 256, 32)
    rms = ep['encoder_block_128x128x64'].float().pow(2).mean(dim=3)
    assert float((rms - 1).abs().max()) < 0.05                                  # pixel norm => unit RMS over C
    out, _ = pggan.generator(tr.P, net, 't', cfg, ep)
    assert out.shape == (2, 256, 256, 3) and bool(torch.isfinite(out.float()).all())
    m = out.float().mean(dim=(1, 2))
    v = out.float().var(dim=(1, 2), unbiased=False)
    assert float(m.abs().max()) < 0.05 and float((v - 1).abs().max()) < 0.1     # to_rgb is instance-normalised
    pred, _ = pggan.discriminator(tr.P, out, cfg, 'discriminator_s')
    assert pred.shape == (2, 1) and bool(torch.isfinite(pred).all())


@pytest.mark.parametrize('hw,variant,precision', [(64, '', 'fp32'), (64, '', 'bf16'), (128, '', 'fp32'), (128, '', 'bf16'),
                                                 (256, '', 'fp32'), (256, '', 'bf16'),
                                                 # BASELINE configs[4]: + self-attention at 64 x 64 + spectral-norm discriminators, in
                                                 # fp32 and in its own storage type, fp16 (static loss scale 128, model_inheritor.py:568-570)
                                                 (256, '_sn_att', 'fp32'), (256, '_sn_att', 'fp16')])
def test_full_width_stage_hits_the_reference(hw, variant, precision):
  """BASELINE.json's configurations at FULL width -- configs[1] (64x64), configs[2] (128x128) and the headline configs[3]
  (256x256), 256 channels, batch 2 -- against what the
  reference's own code computed for each (tests/golden/full_hw<hw>_c256.json, tools/make_golden.py --full [--hw N]: the graph of
  twingan.GanModel._clone_fn executed on the TF stand-in): every loss term, probes of every generated image, and --
  fp32 path -- the norm of every variable's gradient.  Weights and inputs are re-created from the fixture's seeds.
  Measured (tools/full_size_report.py): fp32 path -- worst loss term 2.3e-6, worst probe 2.3e-5, gradient-norm ratios
  0.9988..1.0015; bf16 path -- 7.5e-3, 0.17, ratios 0.75..1.18 with median 0.996.  Bounds: fp32 losses 2e-4 relative
  (to max(1, |x|)), probes 1e-3, gradient norms 1 % (of max(norm, 1e-3 of the group's largest)); bf16 losses 3e-2,
  probes 0.3, gradient-norm ratios 0.5..1.6 with the median within 3 %, and -- the bound that carries the weight -- the
  aggregate deviation of each group's gradients from the float64 ones <= 1.5 x the deviation bf16 storage rounding alone
  causes in the float64 oracle on the same weights and inputs (tests/golden/full_hw<hw>_c256_rounding.json,
  tools/make_rounding_sketch.py [--hw N]).  The figures quoted above are the 256 x 256 ones."""
  import json
  import os
  from twingan_amd import Config
  from twingan_amd import twingan as T
  with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'full_hw%d_c256%s.json' % (hw, variant))) as fh:
    fix = json.load(fh)
  assert fix['config']['hw'] == hw
  batch = fix['batch']
  rcfg = R.Config(**fix['config'])
  P = R.init_params(rcfg, seed=fix['param_seed'], dtype=torch.float64, std='he')
  cfg = Config(precision=precision, **fix['config'])
  tr = T.Trainer(cfg, device='cuda:0', seed=0)
  sd = {k: v.float() for k, v in P.items()}
  if rcfg.spectral_norm:      # the fixture's power-iteration vectors: seeded like the weights (tools/make_golden.py full_size)
    sd.update({k: v.float() for k, v in R.init_sn_state({k: v.float().double() for k, v in P.items()}, seed=fix['param_seed'] + 1).items()})
    assert all(k in tr.store.state for k in sd if k.endswith('/u'))
  tr.store.load_state_dict(sd)
  del P, sd
  g = torch.Generator().manual_seed(fix['input_seed'])
  adt = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}[precision]
  # fp16: the reference's static loss scale (model_inheritor.py:568-570) around the backward, taken out of the gradients again
  scale = 128.0 if precision == 'fp16' else 1.0
  s = torch.rand(batch, hw, hw, 3, generator=g).to('cuda:0').to(adt)
  t = torch.rand(batch, hw, hw, 3, generator=g).to('cuda:0').to(adt)
  a_s = torch.tensor(fix['gp_alpha_s'], dtype=torch.float32, device='cuda:0')
  a_t = torch.tensor(fix['gp_alpha_t'], dtype=torch.float32, device='cuda:0')
  ltol, ptol = (2e-4, 1e-3) if precision == 'fp32' else (3e-2, 0.3)
  step = hw // 4
  with torch.no_grad():
    o = T.forward_generators(tr.P, s, t, cfg)
  for k in ('s_prime', 't_prime', 's_cycle', 't_cycle'):
    want = torch.tensor(fix['probe'][k])
    got = o[k][:, ::step, ::step, :].float().cpu()
    assert float((got - want).abs().max()) < ptol, (k, float((got - want).abs().max()))
  del o
  for group, fn, args, terms_want, total in (
      ('g', T.generator_loss, (s, t, cfg), fix['g_terms'], fix['g_total']),
      ('d', T.discriminator_loss, (s, t, cfg, a_s, a_t), fix['d_terms'], fix['d_total'])):
    tr.store.zero_grad(group)
    tr._set_requires_grad(g=group == 'g', d=group == 'd')
    loss, terms = fn(tr.P, *args)
    assert set(terms) == set(terms_want), group
    for k, v in terms.items():
      assert abs(v.item() - terms_want[k]) < ltol * max(1.0, abs(terms_want[k])), (k, v.item(), terms_want[k])
    assert abs(loss.item() - total) < ltol * max(1.0, abs(total)) * 2, (group, loss.item(), total)
    (loss if scale == 1.0 else loss * scale).backward()
    # spectral norm: the fixture's two losses belong to ONE reference run (the same pre-run u): drop this pass's normalised
    # kernels without assigning its power-iteration vectors
    from twingan_amd import pggan as _pg
    tr.P.__dict__.get('sn_pending', {}).clear()
    _pg.end_run(tr.P)
    gd = {k: (v if scale == 1.0 else v / scale) for k, v in tr.store.grad_dict().items()}
    names = tr.store.names(group)
    floor = 1e-3 * max(fix['grad_norm'][k] for k in names)
    pairs = [(k, float(gd[k].double().norm()), fix['grad_norm'][k]) for k in names]
    if precision == 'fp32':
      # the attention gate sa_gamma (a scalar): its gradient is ONE dot product over a whole feature map that nearly cancels
      # (2.4e-4 from terms summing to ~1e2 in magnitude) -- fp32 leaves it 3 % from the float64 value
      bad = [p for p in pairs if abs(p[1] - p[2]) > (0.05 if p[0].endswith('/sa_gamma') else 0.01) * max(p[2], floor)]
    else:
      ratios = [p[1] / p[2] for p in pairs if p[2] > floor]
      bad = [p for p in pairs if p[2] > floor and not 0.5 < p[1] / p[2] < 1.6]
      assert abs(float(np.median(ratios)) - 1.0) < 0.03, float(np.median(ratios))
      # The statement proper (round 4): the kernels' gradients are no further from the float64 gradients than 1.5 x what
      # bf16 STORAGE ROUNDING ALONE does to this graph on these weights and inputs.  tests/golden/full_hw256_c256_rounding.json
      # (tools/make_rounding_sketch.py, float64 oracle at full size) holds that figure per optimiser group -- the oracle
      # with bf16 rounding at the kernels' storage points moves the generator group by rel-L2 0.417, the discriminator group
      # by 0.080 -- and K = 16 seeded +-1 projections of every variable's float64 gradient, from which |g_hip - g_64|^2 of a
      # group is estimated without the 71 MB of gradients (E[(r . d)^2] = |d|^2; the same estimator reads 0.378 / 0.092 on
      # the rounded oracle itself).
      with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'full_hw%d_c256%s_rounding.json' % (hw, variant))) as fh:
        rs = json.load(fh)
      assert rs['dtype'] == precision
      num = den = 0.0
      for k in names:
        idx = rs['order'].index(k)
        gen = torch.Generator().manual_seed(rs['sketch_seed'] + idx)
        r = (torch.randint(0, 2, (rs['K'], gd[k].numel()), generator=gen, dtype=torch.int8).to('cuda:0').double() * 2.0 - 1.0)
        h = (r @ gd[k].reshape(-1).double().to('cuda:0')).cpu()
        e = torch.tensor(rs['exact_sketch'][k], dtype=torch.float64)
        num += float(((h - e) ** 2).sum())
        den += float((e ** 2).sum())
      rel = (num / den) ** 0.5
      print('[full width %d%s %s] %s group: kernels %.3f from the float64 gradients (sketch estimate); storage rounding alone %.3f'
            % (hw, variant, precision, group, rel, rs['rounded_rel_l2'][group]))
      assert rel < 1.5 * rs['rounded_rel_l2'][group] + 0.02, (group, rel, rs['rounded_rel_l2'][group])
    assert not bad, bad[:5]
    del loss, terms, gd


def _dp_worker(rank, world, port, q, use_graph):
  import os
  import torch.distributed as dist
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  torch.cuda.set_device(0)
  dist.init_process_group('gloo', rank=rank, world_size=world)      # both ranks share the box's single GPU
  try:
    from twingan_amd import Config
    from twingan_amd.twingan import Trainer
    cfg = Config(hw=32, max_ch=16, precision='fp32', loss_architecture='wgan', overlap_cut_hw=8)
    g = torch.Generator().manual_seed(50 + rank)                     # every clone draws its own batch
    s = torch.rand(2, 32, 32, 3, generator=g).to('cuda:0')
    t = torch.rand(2, 32, 32, 3, generator=g).to('cuda:0')
    tr = Trainer(cfg, device='cuda:0', seed=7, world_size=world, use_graph=use_graph)
    assert tr.split and (tr._nseg('g'), tr._nseg('d')) == (3, 2)     # more than one clone: segmented backward
    for _ in range(6):
      tr.run(s, t)
    torch.cuda.synchronize()
    q.put((rank, tr.store.flat['g'].cpu().numpy(), tr.store.flat['d'].cpu().numpy(), tr.use_graph))
    dist.barrier()
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('use_graph', [False, True])
def test_data_parallel_two_clones_one_gpu(use_graph):
  """Two clones (processes) on the one GPU of the test box, gloo instead of RCCL: exercises the Trainer's DP
  path -- loss / num_clones, the segmented backward with one all-reduce per segment range of the flat gradient
  buffers (enqueued between the segment graphs, waited for before the apply graph), graph capture with a process
  group alive.  Both clones must end with identical parameters, different from a
  single clone's (deployment/model_deploy.py:242-315,473-503)."""
  import socket
  import torch.multiprocessing as mp
  sk = socket.socket()
  sk.bind(('127.0.0.1', 0))
  port = sk.getsockname()[1]
  sk.close()
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, use_graph)) for r in range(2)]
  for p in procs:
    p.start()
  got = dict()
  for _ in range(2):
    r, fg, fd, graphed = q.get(timeout=600)
    got[r] = (fg, fd, graphed)
  for p in procs:
    p.join(timeout=600)
    assert p.exitcode == 0
  for i in (0, 1):
    a, b = got[0][i], got[1][i]
    assert np.abs(a - b).max() <= 1e-6 * max(1.0, np.abs(a).max()), 'clones diverged'
  assert got[0][2] == use_graph, 'graph capture fell back to eager'
  # single clone on clone 0's batch moves differently
  from twingan_amd import Config
  from twingan_amd.twingan import Trainer
  cfg = Config(hw=32, max_ch=16, precision='fp32', loss_architecture='wgan', overlap_cut_hw=8)
  g = torch.Generator().manual_seed(50)
  s = torch.rand(2, 32, 32, 3, generator=g).to('cuda:0')
  t = torch.rand(2, 32, 32, 3, generator=g).to('cuda:0')
  tr = Trainer(cfg, device='cuda:0', seed=7)
  for _ in range(6):
    tr.run(s, t)
  assert np.abs(tr.store.flat['g'].cpu().numpy() - got[0][0]).max() > 1e-6


@pytest.mark.parametrize('loss,drift', [('hinge', 0.0), ('gan', 0.0), ('dragan', 0.0), ('wgan', 0.001)])
def test_loss_architectures_match_oracle(loss, drift):
  """--loss_architecture hinge / gan / dragan and the WGAN drift term (image_generation.py:331-400,441-476)."""
  from twingan_amd import Config
  from twingan_amd import twingan as T
  from twingan_amd.twingan import Trainer
  cfg = Config(hw=16, max_ch=16, precision='fp32', loss_architecture=loss, wgan_drift_loss_weight=drift,
               gradient_penalty_lambda=0.25 if loss == 'dragan' else 10.0)
  rcfg = R.Config(hw=16, max_ch=16, loss=loss, drift=drift, gp_lambda=cfg.gradient_penalty_lambda)
  Pref = R.init_params(rcfg, seed=6, dtype=torch.float64, std='he')
  tr = Trainer(cfg, device='cuda:0', seed=6)
  tr.store.load_state_dict({k: v.float() for k, v in Pref.items()})
  Pref = {k: v.float().double().requires_grad_(True) for k, v in Pref.items()}
  g = torch.Generator().manual_seed(77)
  s, t = torch.rand(2, 16, 16, 3, generator=g), torch.rand(2, 16, 16, 3, generator=g)
  a_s, a_t = torch.rand(2, generator=g), torch.rand(2, generator=g)
  n_s, n_t = torch.rand(2, 16, 16, 3, generator=g) * 2 - 1, torch.rand(2, 16, 16, 3, generator=g) * 2 - 1
  dev = lambda x: x.to('cuda:0').contiguous()
  # generator side
  tr.store.zero_grad('g')
  tr._set_requires_grad(g=True, d=False)
  gl, gterms = T.generator_loss(tr.P, dev(s), dev(t), cfg)
  rgl, rgterms = R.generator_loss(Pref, s.double(), t.double(), rcfg)
  assert set(gterms) == set(rgterms)
  for k in rgterms:
    assert abs(gterms[k].item() - rgterms[k].item()) < 1e-4 * max(1.0, abs(rgterms[k].item())), k
  gl.backward()
  rgl.backward()
  _grads_close(tr, Pref, tr.store.names('g'), FP32_GRAD_TOL, 'generator', var_tol=FP32_VAR_GRAD_TOL)
  for v in Pref.values():
    v.grad = None
  # discriminator side
  tr.store.zero_grad('d')
  tr._set_requires_grad(g=False, d=True)
  dl, dterms = T.discriminator_loss(tr.P, dev(s), dev(t), cfg, dev(a_s), dev(a_t), dev(n_s), dev(n_t))
  rdl, rdterms = R.discriminator_loss(Pref, s.double(), t.double(), rcfg, a_s.double().reshape(-1, 1, 1, 1),
                                      a_t.double().reshape(-1, 1, 1, 1), n_s.double(), n_t.double())
  assert set(dterms) == set(rdterms), (sorted(dterms), sorted(rdterms))
  for k in rdterms:
    assert abs(dterms[k].item() - rdterms[k].item()) < 1e-4 * max(1.0, abs(rdterms[k].item())), (k, dterms[k].item(), rdterms[k].item())
  dl.backward()
  rdl.backward()
  _grads_close(tr, Pref, tr.store.names('d'), FP32_GRAD_TOL, 'discriminator', var_tol=FP32_VAR_GRAD_TOL)


def test_batch_norm_generator_matches_oracle():
  """generator_norm_type=batch_norm -- the reference's default (nets/pggan.py:24; libs/batch_norm.py): per-pass batch
  statistics even though the passes are batched along N here, per-domain gamma/beta and moving statistics."""
  from twingan_amd import Config
  from twingan_amd import twingan as T
  from twingan_amd.twingan import Trainer
  cfg = Config(hw=16, max_ch=16, precision='fp32', generator_norm_type='batch_norm')
  state = {}
  rcfg = R.Config(hw=16, max_ch=16, norm='batch_norm', bn_state=state)
  Pref = R.init_params(rcfg, seed=8, dtype=torch.float64, std='he')
  tr = Trainer(cfg, device='cuda:0', seed=8)
  assert set(tr.store.state_dict()) == set(Pref)
  tr.store.load_state_dict({k: v.float() for k, v in Pref.items()})
  Pref = {k: v.float().double().requires_grad_(True) for k, v in Pref.items()}
  g = torch.Generator().manual_seed(78)
  s, t = torch.rand(3, 16, 16, 3, generator=g), torch.rand(3, 16, 16, 3, generator=g)
  dev = lambda x: x.to('cuda:0').contiguous()
  tr.store.zero_grad('g')
  tr._set_requires_grad(g=True, d=False)
  gl, gterms = T.generator_loss(tr.P, dev(s), dev(t), cfg)
  rgl, rgterms = R.generator_loss(Pref, s.double(), t.double(), rcfg)
  for k in rgterms:
    assert abs(gterms[k].item() - rgterms[k].item()) < 1e-4 * max(1.0, abs(rgterms[k].item())), k
  gl.backward()
  rgl.backward()
  _grads_close(tr, Pref, tr.store.names('g'), FP32_GRAD_TOL, 'generator(batch_norm)', var_tol=FP32_VAR_GRAD_TOL)
  # moving statistics: same set of variables, same values (each pass = one assign_moving_average)
  assert set(tr.store.state) == set(state)
  for k, v in state.items():
    a = tr.store.state[k].double().cpu().numpy()
    assert np.abs(a - v.numpy()).max() < 1e-5 * max(1.0, np.abs(v.numpy()).max()), k


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_no_normaliser_generator_matches_oracle(precision):
  """generator_norm_type=none (nets/pggan_utils.py:198-200): the encoder / generator convs own a bias, LeakyReLU runs in
  the conv epilogue and the pixel norm is the fused kernel with constant statistics (ops.pixel_norm, NF_NOSTATS in
  its backward).  Losses and gradients against the oracle (pinned live: test_reference_live 'no_normaliser')."""
  from twingan_amd import Config
  from twingan_amd import twingan as T
  from twingan_amd.twingan import Trainer
  cfg = Config(hw=16, max_ch=16, precision=precision, generator_norm_type='none')
  rcfg = R.Config(hw=16, max_ch=16, norm='none')
  Pref = R.init_params(rcfg, seed=9, dtype=torch.float64, std='he')
  tr = Trainer(cfg, device='cuda:0', seed=9)
  assert set(tr.store.state_dict()) == set(Pref) and 'generator/block_8x8x16/Conv/biases' in Pref
  tr.store.load_state_dict({k: v.float() for k, v in Pref.items()})
  Pref = {k: v.float().double().requires_grad_(True) for k, v in Pref.items()}
  g = torch.Generator().manual_seed(79)
  adt = torch.bfloat16 if precision == 'bf16' else torch.float32
  s, t = (torch.rand(3, 16, 16, 3, generator=g).to(adt).float() for _ in range(2))
  dev = lambda x: x.to('cuda:0').to(adt).contiguous()
  tr.store.zero_grad('g')
  tr._set_requires_grad(g=True, d=False)
  gl, gterms = T.generator_loss(tr.P, dev(s), dev(t), cfg)
  rgl, rgterms = R.generator_loss(Pref, s.double(), t.double(), rcfg)
  tol = 1e-4 if precision == 'fp32' else 5e-2
  for k in rgterms:
    assert abs(gterms[k].item() - rgterms[k].item()) < tol * max(1.0, abs(rgterms[k].item())), k
  gl.backward()
  rgl.backward()
  if precision == 'fp32':
    _grads_close(tr, Pref, tr.store.names('g'), FP32_GRAD_TOL, 'generator(no normaliser)', var_tol=FP32_VAR_GRAD_TOL)
  else:
    _grads_close(tr, Pref, tr.store.names('g'), 0.5, 'generator(no normaliser, bf16)', min_cos=0.9)


@pytest.mark.parametrize('norm,global_step', [('batch_renorm', 0), ('batch_renorm', 25000), ('batch_renorm_native', 0),
                                              ('batch_renorm_native', 25000)])
def test_batch_renorm_generator_matches_oracle(norm, global_step):
  """generator_norm_type=batch_renorm -- the configuration the reference's training guide uses (docs/training.md:17;
  libs/batch_norm.py:209-246,329-470): r / d corrections from the renorm statistics (stop-gradient, clipped by the
  global-step schedule), renorm + moving statistics updated per pass in call order.  Two consecutive generator-loss
  evaluations so the second one sees non-trivial renorm state.  'batch_renorm_native' (nets/pggan_utils.py:175-188:
  tf.contrib's layers.batch_norm(renorm=True, scope=<postfix>)) is the same arithmetic on variables named
  '<conv>/_s/{gamma, beta, moving_*, renorm_*}' (pinned live: test_reference_live 'batch_renorm_native_*')."""
  from twingan_amd import Config
  from twingan_amd import twingan as T
  from twingan_amd.twingan import Trainer
  cfg = Config(hw=16, max_ch=16, precision='fp32', generator_norm_type=norm)
  state = {}
  rcfg = R.Config(hw=16, max_ch=16, norm=norm, bn_state=state, global_step=global_step)
  Pref = R.init_params(rcfg, seed=8, dtype=torch.float64, std='he')
  tr = Trainer(cfg, device='cuda:0', seed=8)
  tr.global_step = global_step
  tr._set_renorm_clipping()
  assert set(tr.store.state_dict()) == set(Pref)
  tr.store.load_state_dict({k: v.float() for k, v in Pref.items()})
  Pref = {k: v.float().double().requires_grad_(True) for k, v in Pref.items()}
  g = torch.Generator().manual_seed(78)
  dev = lambda x: x.to('cuda:0').contiguous()
  for it in range(2):
    s, t = torch.rand(3, 16, 16, 3, generator=g), torch.rand(3, 16, 16, 3, generator=g)
    for v in Pref.values():
      v.grad = None
    tr.store.zero_grad('g')
    tr._set_requires_grad(g=True, d=False)
    gl, gterms = T.generator_loss(tr.P, dev(s), dev(t), cfg)
    rgl, rgterms = R.generator_loss(Pref, s.double(), t.double(), rcfg)
    for k in rgterms:
      assert abs(gterms[k].item() - rgterms[k].item()) < 2e-4 * max(1.0, abs(rgterms[k].item())), (it, k)
    gl.backward()
    rgl.backward()
    _grads_close(tr, Pref, tr.store.names('g'), FP32_GRAD_TOL, 'generator(%s, run %d)' % (norm, it), var_tol=FP32_VAR_GRAD_TOL)
  renorm_state = {k: v for k, v in tr.store.state.items() if not k.startswith('renorm/')}
  assert set(renorm_state) == set(state)
  assert any(k.endswith('/_t/renorm_stddev_weight' if norm.endswith('native') else '/BatchNorm/renorm_stddev_weight_t') for k in state)
  for k, v in state.items():
    a = renorm_state[k].double().cpu().numpy()
    assert np.abs(a - v.numpy()).max() < 2e-5 * max(1.0, np.abs(v.numpy()).max()), k
  if norm.endswith('native'):      # the inference branch on the moving statistics the two runs left (twingan.py:300-363)
    rcfg.bn_state = {k: v.clone() for k, v in state.items()}
    x = torch.rand(2, 16, 16, 3, generator=g)
    with torch.no_grad():
      want = R.translate({k: v.detach() for k, v in Pref.items()}, x.double(), rcfg, 't')
    got = T.translate(tr.P, dev(x), cfg, 't')
    assert rel_l2(got, want) < 2e-5, rel_l2(got, want)


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_layer_norm_native_generator_matches_oracle(precision):
  """generator_norm_type=layer_norm_native (nets/pggan_utils.py:189-197: tf.contrib.layers.layer_norm(center, scale,
  scope=<postfix>)): every encoder / generator conv is normalised with the statistics of one IMAGE over (H, W, C), gamma /
  beta per channel and per domain ('<conv>/_s/gamma'), epsilon 1e-12 -- ops.layer_norm_act (the statistics kernel as a
  differentiable node + the fused per-image-row kernel).  Losses, gradients and the inference branch (the layer keeps no
  state: the same arithmetic) against the oracle (pinned live: test_reference_live 'layer_norm_native*')."""
  from twingan_amd import Config
  from twingan_amd import twingan as T
  from twingan_amd.twingan import Trainer
  cfg = Config(hw=16, max_ch=16, precision=precision, generator_norm_type='layer_norm_native')
  rcfg = R.Config(hw=16, max_ch=16, norm='layer_norm_native')
  Pref = R.init_params(rcfg, seed=10, dtype=torch.float64, std='he')
  rng = torch.Generator().manual_seed(80)
  for k in Pref:      # non-trivial per-channel parameters, different per domain
    if k.endswith('/gamma'):
      Pref[k] = (1.0 + 0.3 * torch.randn(Pref[k].shape, generator=rng)).double()
    elif k.endswith('/beta'):
      Pref[k] = (0.2 * torch.randn(Pref[k].shape, generator=rng)).double()
  tr = Trainer(cfg, device='cuda:0', seed=10)
  assert set(tr.store.state_dict()) == set(Pref) and 'generator/block_8x8x16/Conv/_t/gamma' in Pref
  assert not [k for k in tr.store.state if not k.startswith('renorm/')]      # no moving statistics
  tr.store.load_state_dict({k: v.float() for k, v in Pref.items()})
  Pref = {k: v.float().double().requires_grad_(True) for k, v in Pref.items()}
  adt = torch.bfloat16 if precision == 'bf16' else torch.float32
  s, t = (torch.rand(3, 16, 16, 3, generator=rng).to(adt).float() for _ in range(2))
  dev = lambda x: x.to('cuda:0').to(adt).contiguous()
  tr.store.zero_grad('g')
  tr._set_requires_grad(g=True, d=False)
  gl, gterms = T.generator_loss(tr.P, dev(s), dev(t), cfg)
  rgl, rgterms = R.generator_loss(Pref, s.double(), t.double(), rcfg)
  tol = 1e-4 if precision == 'fp32' else 5e-2
  assert set(gterms) == set(rgterms)
  for k in rgterms:
    assert abs(gterms[k].item() - rgterms[k].item()) < tol * max(1.0, abs(rgterms[k].item())), (k, gterms[k].item(), rgterms[k].item())
  gl.backward()
  rgl.backward()
  if precision == 'fp32':
    _grads_close(tr, Pref, tr.store.names('g'), FP32_GRAD_TOL, 'generator(layer_norm_native)', var_tol=FP32_VAR_GRAD_TOL)
  else:
    _grads_close(tr, Pref, tr.store.names('g'), 0.5, 'generator(layer_norm_native, bf16)', min_cos=0.9)
  x = torch.rand(2, 16, 16, 3, generator=rng).to(adt).float()
  with torch.no_grad():
    want = R.translate({k: v.detach() for k, v in Pref.items()}, x.double(), rcfg, 's')
  got = T.translate(tr.P, dev(x), cfg, 's')
  assert rel_l2(got, want) < (2e-5 if precision == 'fp32' else 0.1), rel_l2(got, want)
  # and as a training step through the trainer (eager and as a captured graph: the row arithmetic is capturable)
  for use_graph in (False, True):
    tr2 = Trainer(cfg, device='cuda:0', seed=10, use_graph=use_graph)
    for _ in range(2):
      loss, _ = tr2.run(dev(s), dev(t))
    assert np.isfinite(float(loss))
    tr2.close()


@pytest.mark.parametrize('norm,both', [('instance_norm', True), ('batch_norm', False)])
def test_encoder_distillation_matches_oracle(norm, both):
  """--do_encoder_distillation (twingan.py:207-230,290-298,507-521): the two encoder_classification heads under
  encoder_content on the original and the re-encoded content, cosine distance to the dataset's embeddings
  (tg_cosine_distance_*); `both` False: only the source dataset carries embeddings (two of the four terms exist, but all
  four head applications still move the BatchNorm statistics).  Oracle pinned live (test_reference_live distillation*)."""
  from twingan_amd import Config
  from twingan_amd import twingan as T
  from twingan_amd.twingan import Trainer
  cfg = Config(hw=16, max_ch=16, precision='fp32', generator_norm_type=norm, do_encoder_distillation=True,
               distill_embed_dim=6, distillation_weight=0.7)
  state = {}
  g = torch.Generator().manual_seed(98)
  emb_s, emb_t = torch.randn(3, 6, generator=g), (torch.randn(3, 6, generator=g) if both else None)
  rcfg = R.Config(hw=16, max_ch=16, norm=norm, bn_state=state if norm != 'instance_norm' else None,
                  do_encoder_distillation=True, distill_embed_dim=6, distillation_weight=0.7,
                  distill_embed_s=emb_s.double(), distill_embed_t=None if emb_t is None else emb_t.double())
  Pref = R.init_params(rcfg, seed=19, dtype=torch.float64, std='he')
  tr = Trainer(cfg, device='cuda:0', seed=19)
  assert set(tr.store.state_dict()) == set(Pref)
  tr.store.load_state_dict({k: v.float() for k, v in Pref.items()})
  Pref = {k: v.float().double().requires_grad_(True) for k, v in Pref.items()}
  s, t = torch.rand(3, 16, 16, 3, generator=g), torch.rand(3, 16, 16, 3, generator=g)
  dev = lambda x: None if x is None else x.to('cuda:0').contiguous()
  tr.store.zero_grad('g')
  tr._set_requires_grad(g=True, d=False)
  gl, gterms = T.generator_loss(tr.P, dev(s), dev(t), cfg, distill_embed_s=dev(emb_s), distill_embed_t=dev(emb_t))
  rgl, rgterms = R.generator_loss(Pref, s.double(), t.double(), rcfg)
  assert set(gterms) == set(rgterms) and ('l_target_distillation' in gterms) == both and 'l_t_prime_distillation' in gterms
  for k in rgterms:
    assert abs(gterms[k].item() - rgterms[k].item()) < 2e-4 * max(1.0, abs(rgterms[k].item())), k
  gl.backward()
  rgl.backward()
  _grads_close(tr, Pref, tr.store.names('g'), FP32_GRAD_TOL, 'generator(distillation, %s)' % norm, var_tol=FP32_VAR_GRAD_TOL)
  if norm == 'batch_norm':
    for k, v in state.items():
      a = tr.store.state[k].double().cpu().numpy()
      assert np.abs(a - v.numpy()).max() < 1e-5 * max(1.0, np.abs(v.numpy()).max()), k
  # and through Trainer.run (eager: the embeddings are per-batch dataset fields)
  tr2 = Trainer(cfg, device='cuda:0', seed=19)
  l0, terms0 = tr2.run(dev(s), dev(t), distill_embed_s=dev(emb_s), distill_embed_t=dev(emb_t))
  assert 'l_source_distillation' in terms0 and bool(torch.isfinite(l0).all())


@pytest.mark.parametrize('norm', ['batch_norm', 'batch_renorm'])
def test_style_embedding_on_batch_norms_matches_oracle(norm):
  """--use_style_embedding on the batch-norm family (libs/batch_norm.py:82-85,152-159,209-259,403-470): pass statistics
  from the HIP normaliser, then the per-image rows gamma = 1 + FC(l2n(e)), beta = FC(l2n(e)) -- composed with the
  renorm r / d of the pass for batch_renorm -- LeakyReLU and pixel norm in the fused kernel's per-image-row mode with
  constant statistics (ops.affine_act).  Oracle pinned live: test_reference_live style_batch_norm / style_batch_renorm."""
  from twingan_amd import Config
  from twingan_amd import twingan as T
  from twingan_amd.twingan import Trainer
  cfg = Config(hw=16, max_ch=16, precision='fp32', generator_norm_type=norm, use_style_embedding=True, style_embed_size=6)
  state = {}
  rcfg = R.Config(hw=16, max_ch=16, norm=norm, bn_state=state, use_style_embedding=True, style_embed_size=6)
  Pref = R.init_params(rcfg, seed=18, dtype=torch.float64, std='he')
  tr = Trainer(cfg, device='cuda:0', seed=18)
  if norm == 'batch_renorm':
    tr._set_renorm_clipping()
  assert set(tr.store.state_dict()) == set(Pref)
  tr.store.load_state_dict({k: v.float() for k, v in Pref.items()})
  Pref = {k: v.float().double().requires_grad_(True) for k, v in Pref.items()}
  g = torch.Generator().manual_seed(88)
  s, t = torch.rand(3, 16, 16, 3, generator=g), torch.rand(3, 16, 16, 3, generator=g)
  noise = torch.randn(3, 6, generator=g)
  rcfg.style_noise = noise.double()
  dev = lambda x: x.to('cuda:0').contiguous()
  tr.store.zero_grad('g')
  tr._set_requires_grad(g=True, d=False)
  gl, gterms = T.generator_loss(tr.P, dev(s), dev(t), cfg, dev(noise))
  rgl, rgterms = R.generator_loss(Pref, s.double(), t.double(), rcfg)
  assert set(gterms) == set(rgterms)
  for k in rgterms:
    assert abs(gterms[k].item() - rgterms[k].item()) < 2e-4 * max(1.0, abs(rgterms[k].item())), k
  gl.backward()
  rgl.backward()
  _grads_close(tr, Pref, tr.store.names('g'), FP32_GRAD_TOL, 'generator(style on %s)' % norm, var_tol=FP32_VAR_GRAD_TOL)


@pytest.mark.parametrize('loss', ['hinge', 'wgan_gp'])
def test_spectral_norm_and_self_attention(loss):
  """SURVEY config 4: --spectral_norm on the discriminator convs (libs/sn.py:38-101, gradient through sigma, u assigned
  once per run) and --do_self_attention in E / G / D (libs/self_attention.py:24-70), against the oracle: losses,
  gradients (under wgan_gp also the double backward through the normalised kernels and the attention), and the
  power-iteration state after each run."""
  from twingan_amd import pggan
  from twingan_amd import twingan as T
  kw = dict(hw=32, max_ch=32, spectral_norm=True, do_self_attention=True, self_attention_hw=16, loss_architecture=loss)
  cfg, rcfg, tr, Pref, dev, ref = make(kw, 'fp32', seed=6, batch=2)
  assert set(tr.store.state_dict()) == set(Pref)
  rcfg.sn_state = R.init_sn_state(Pref, seed=3)
  assert set(rcfg.sn_state) == set(tr.store.state)
  for k, v in rcfg.sn_state.items():
    tr.store.state[k].copy_(v.float())
    rcfg.sn_state[k] = v.float().double()
  for v in Pref.values():
    v.requires_grad_(True)
  for it in range(2):
    for v in Pref.values():
      v.grad = None
    tr.store.zero_grad('g')
    tr._set_requires_grad(g=True, d=False)
    gl, gterms = T.generator_loss(tr.P, dev['s'], dev['t'], cfg)
    rgl, rterms = R.generator_loss(Pref, ref['s'], ref['t'], rcfg)
    for k in rterms:
      assert abs(gterms[k].item() - rterms[k].item()) < 1e-4 * max(1.0, abs(rterms[k].item())), (it, k)
    gl.backward()
    rgl.backward()
    pggan.end_run(tr.P)
    R.end_run(rcfg)
    _grads_close(tr, Pref, tr.store.names('g'), FP32_GRAD_TOL, 'generator run %d' % it, var_tol=FP32_VAR_GRAD_TOL)
    for v in Pref.values():
      v.grad = None
    tr.store.zero_grad('d')
    tr._set_requires_grad(g=False, d=True)
    dl, dterms = T.discriminator_loss(tr.P, dev['s'], dev['t'], cfg, dev['a_s'], dev['a_t'])
    rdl, rdterms = R.discriminator_loss(Pref, ref['s'], ref['t'], rcfg, ref['a_s'], ref['a_t'])
    for k in rdterms:
      assert abs(dterms[k].item() - rdterms[k].item()) < 1e-4 * max(1.0, abs(rdterms[k].item())), (it, k)
    dl.backward()
    rdl.backward()
    pggan.end_run(tr.P)
    R.end_run(rcfg)
    _grads_close(tr, Pref, tr.store.names('d'), FP32_GRAD_TOL, 'discriminator run %d' % it, var_tol=FP32_VAR_GRAD_TOL)
    for k, v in rcfg.sn_state.items():
      assert rel_l2(tr.store.state[k], v) < 1e-4, (it, k)


def test_spectral_norm_in_encoder_and_generator():
  """--spectral_norm_in_non_discriminator (nets/pggan.py:31-33): the encoder / generator conv kernels (incl. from-RGB,
  to-RGB and the residual shortcuts) are spectrally normalised too; generator loss terms, gradients through sigma and
  the power-iteration vectors after the run, against the oracle (which is pinned against the reference for this flag:
  tests/test_reference_live.py::sn_everywhere)."""
  from twingan_amd import pggan
  from twingan_amd import twingan as T
  kw = dict(hw=16, max_ch=16, spectral_norm=True, spectral_norm_in_non_discriminator=True, use_res_block=True)
  cfg, rcfg, tr, Pref, dev, ref = make(kw, 'fp32', seed=8, batch=2)
  rcfg.sn_state = R.init_sn_state(Pref, seed=4, non_disc=True)
  assert set(rcfg.sn_state) == set(tr.store.state), sorted(set(rcfg.sn_state) ^ set(tr.store.state))[:6]
  assert any(k.startswith('generator/') for k in rcfg.sn_state) and any('/shortcut/u' in k for k in rcfg.sn_state)
  for k, v in rcfg.sn_state.items():
    tr.store.state[k].copy_(v.float())
    rcfg.sn_state[k] = v.float().double()
  for v in Pref.values():
    v.requires_grad_(True)
  tr.store.zero_grad('g')
  tr._set_requires_grad(g=True, d=False)
  gl, gterms = T.generator_loss(tr.P, dev['s'], dev['t'], cfg)
  rgl, rterms = R.generator_loss(Pref, ref['s'], ref['t'], rcfg)
  for k in rterms:
    assert abs(gterms[k].item() - rterms[k].item()) < 1e-4 * max(1.0, abs(rterms[k].item())), k
  gl.backward()
  rgl.backward()
  pggan.end_run(tr.P)
  R.end_run(rcfg)
  _grads_close(tr, Pref, tr.store.names('g'), FP32_GRAD_TOL, 'generator run', var_tol=FP32_VAR_GRAD_TOL)
  for k, v in rcfg.sn_state.items():
    assert rel_l2(tr.store.state[k], v) < 1e-4, k


@pytest.mark.parametrize('precision,loss,scale', [('bf16', 'hinge', 1.0), ('fp16', 'wgan_gp', 128.0)])
def test_spectral_norm_attention_bf16_graph_step_runs(precision, loss, scale):
  """configs[4]'s ingredients as captured graphs: spectral norm + self-attention on bf16, and on fp16 storage with the
  static loss scale 128 under WGAN-GP (the double backward through the discriminator's attention in half precision)."""
  from twingan_amd import Config
  from twingan_amd.twingan import Trainer
  cfg = Config(hw=32, max_ch=64, spectral_norm=True, do_self_attention=True, self_attention_hw=16, loss_architecture=loss,
               precision=precision, loss_scale=scale)
  tr = Trainer(cfg, device='cuda:0', seed=1, use_graph=True)
  g = torch.Generator().manual_seed(3)
  dt = torch.bfloat16 if precision == 'bf16' else torch.float16
  s = torch.rand(4, 32, 32, 3, generator=g).to('cuda:0').to(dt)
  t = torch.rand(4, 32, 32, 3, generator=g).to('cuda:0').to(dt)
  u0 = tr.store.state['discriminator_s/from_rgb_32x32/Conv/u'].clone()
  for _ in range(6):
    loss, terms = tr.run(s, t)
    assert np.isfinite(float(loss)) and all(np.isfinite(float(v)) for v in terms.values())
  assert float((tr.store.state['discriminator_s/from_rgb_32x32/Conv/u'] - u0).abs().max()) > 0


@pytest.mark.parametrize('attention', [False, True])
def test_style_embedding_matches_oracle(attention):
  """--use_style_embedding (twingan.py:47-51,201-288,495-505): the style encoder (pggan.encoder: classification head on
  the encoder body), generator normalisers conditioned on the embedding (gamma = 1 + FC(e), beta = FC(e), one row per
  image: random embedding for s' / t', the encoded style for the cycle images), and the style read-back losses."""
  from twingan_amd import twingan as T
  kw = dict(hw=16, max_ch=16, use_style_embedding=True, style_embed_size=8, do_self_attention=attention,
            self_attention_hw=8)
  cfg, rcfg, tr, Pref, dev, ref = make(kw, 'fp32', seed=9, batch=2)
  assert set(tr.store.state_dict()) == set(Pref)
  g = torch.Generator().manual_seed(5)
  noise = torch.randn(2, 8, generator=g)
  rcfg.style_noise = noise.double()
  for v in Pref.values():
    v.requires_grad_(True)
  tr.store.zero_grad('g')
  tr._set_requires_grad(g=True, d=False)
  gl, gterms = T.generator_loss(tr.P, dev['s'], dev['t'], cfg, style_noise=noise.to('cuda:0'))
  rgl, rterms = R.generator_loss(Pref, ref['s'], ref['t'], rcfg)
  assert set(gterms) == set(rterms) and 'l_style_s' in rterms
  for k in rterms:
    assert abs(gterms[k].item() - rterms[k].item()) < 1e-4 * max(1.0, abs(rterms[k].item())), k
  gl.backward()
  rgl.backward()
  _grads_close(tr, Pref, tr.store.names('g'), FP32_GRAD_TOL, 'generator(style)', var_tol=FP32_VAR_GRAD_TOL)
  # the discriminator step only needs the forward of the styled generators
  tr.store.zero_grad('d')
  tr._set_requires_grad(g=False, d=True)
  dl, dterms = T.discriminator_loss(tr.P, dev['s'], dev['t'], cfg, dev['a_s'], dev['a_t'])
  assert np.isfinite(float(dl))


def test_style_embedding_bf16_graph_step_runs():
  from twingan_amd import Config
  from twingan_amd.twingan import Trainer
  cfg = Config(hw=32, max_ch=32, use_style_embedding=True)
  tr = Trainer(cfg, device='cuda:0', seed=1, use_graph=True)
  g = torch.Generator().manual_seed(3)
  s = torch.rand(4, 32, 32, 3, generator=g).to('cuda:0').bfloat16()
  t = torch.rand(4, 32, 32, 3, generator=g).to('cuda:0').bfloat16()
  # detach(): an autograd op on a parameter here would create its AccumulateGrad node on the default stream, and the
  # captured backward would then synchronise with a stream that is not capturing (hipStreamEndCapture crashes)
  before = tr.store.P['encoder_style/prediction/fully_connected/weights'].detach().clone()
  for _ in range(6):
    loss, terms = tr.run(s, t)
    assert np.isfinite(float(loss)) and all(np.isfinite(float(v)) for v in terms.values())
  after = tr.store.P['encoder_style/prediction/fully_connected/weights'].detach()
  assert float((after - before).abs().max()) > 0


@pytest.mark.parametrize('hw,growing', [(64, False), (128, True), (512, False)])
def test_full_width_stages_run(hw, growing):
  """Every progressive stage of the reference schedule at its real channel widths (pggan_max_num_channels=256): the
  kernel dispatch of each resolution (4x4 ... 512x512, 8 ... 256 channels) on the bf16 path.  At 64x64 the bf16 MFMA
  losses are also checked against the fp32 direct kernels on the same weights and inputs."""
  from twingan_amd import Config
  from twingan_amd import twingan as T
  from twingan_amd.twingan import Trainer
  b = 2 if hw < 512 else 1
  g = torch.Generator().manual_seed(hw)
  s = torch.rand(b, hw, hw, 3, generator=g).to('cuda:0')
  t = torch.rand(b, hw, hw, 3, generator=g).to('cuda:0')
  a = torch.rand(b, generator=g).to('cuda:0')
  kw = dict(hw=hw, max_ch=256, is_growing=growing, alpha_grow=0.3 if growing else 0.0)
  tr = Trainer(Config(precision='bf16', **kw), device='cuda:0', seed=4)
  for _ in range(2):
    loss, terms = tr.run(s.bfloat16(), t.bfloat16())
    assert np.isfinite(float(loss)) and all(np.isfinite(float(v)) for v in terms.values())
  if hw == 64:
    sd = tr.store.state_dict()
    ref = Trainer(Config(precision='fp32', **kw), device='cuda:0', seed=4)
    ref.store.load_state_dict(sd)
    with torch.no_grad():
      lb, tb = T.generator_loss(tr.P, s.bfloat16(), t.bfloat16(), tr.cfg)
      lf, tf_ = T.generator_loss(ref.P, s, t, ref.cfg)
    for k in tf_:
      assert abs(float(tb[k]) - float(tf_[k])) < 5e-2 * max(1.0, abs(float(tf_[k]))), (k, float(tb[k]), float(tf_[k]))
    tr._set_requires_grad(g=False, d=True)
    ref._set_requires_grad(g=False, d=True)
    lb, tb = T.discriminator_loss(tr.P, s.bfloat16(), t.bfloat16(), tr.cfg, a, a)
    lf, tf_ = T.discriminator_loss(ref.P, s, t, ref.cfg, a, a)
    for k in tf_:
      assert abs(float(tb[k]) - float(tf_[k])) < 5e-2 * max(1.0, abs(float(tf_[k]))) + 2e-2, (k, float(tb[k]), float(tf_[k]))


def test_full_size_bf16_layers_teacher_forced():
  """The bf16 / MFMA path at FULL size (256x256, 256 channels, batch 4 -- enough tiles for the weight-resident thin-layer
  kernels), layer by layer: every conv layer of E, G and D (conv + instance norm + LeakyReLU + pixel norm [+ avg-pool],
  or conv + bias + LeakyReLU [+ avg-pool], incl. the two-source concat convs and the minibatch-stddev tail) receives
  the ORACLE's input for that layer (float64, rounded to bf16 at the product's storage points) and its output is held
  to the oracle's output for the same input: rel-L2 <= 1e-2 (SURVEY.md 8c's forward bound).  Teacher forcing removes
  the chaos of the whole random-weight graph under storage rounding (tools/bf16_sensitivity.py), which is what forces
  the loose whole-model bf16 bounds above: here a wrong layer cannot hide behind it."""
  from twingan_amd import Config, pggan
  from twingan_amd.twingan import Trainer
  hw, mc, batch = 256, 256, 4
  cfg = Config(hw=hw, max_ch=mc, precision='bf16')
  rcfg = R.Config(hw=hw, max_ch=mc)
  P = R.init_params(rcfg, seed=31, dtype=torch.float64, std='he')

  def rnd(v):
    return v.to(torch.bfloat16).to(v.dtype)
  P = {k: (rnd(v.float()).double() if k.endswith('/weights') and v.dim() == 4 else v.float().double()) for k, v in P.items()}
  tr = Trainer(cfg, device='cuda:0', seed=0)
  tr.store.load_state_dict({k: v.float() for k, v in P.items()})
  g = torch.Generator().manual_seed(32)
  s = rnd(torch.rand(batch, hw, hw, 3, generator=g)).double()

  # ---- oracle pass, recording every layer's (unrounded) output; the next layer sees it rounded to bf16
  rec = {}
  o_ge, o_d, o_pool = R.ge_conv, R.d_conv, R.avg_pool2

  def r_ge(P_, scope, x, *a, **kw):
    y = o_ge(P_, scope, x, *a, **kw)
    rec[scope] = y.float()
    return rnd(y.float()).double()

  def r_d(P_, scope, x, *a, **kw):
    y = o_d(P_, scope, x, *a, **kw)
    rec[scope] = y.float()
    return rnd(y.float()).double()
  R.ge_conv, R.d_conv, R.avg_pool2 = r_ge, r_d, lambda x: rnd(o_pool(x).float()).double()
  try:
    with torch.no_grad():
      rnet, rep = R.encoder(P, s, 's', rcfg)
      rout, _ = R.generator(P, rnet, 't', rcfg, rep)
      rpred, _ = R.discriminator(P, rnd(rout.float()).double(), rcfg, 'discriminator_t')
  finally:
    R.ge_conv, R.d_conv, R.avg_pool2 = o_ge, o_d, o_pool

  # ---- product pass: each layer computes from what it is given (= oracle outputs of the layers before it), is
  # compared, and hands the oracle's output on
  errs = {}
  p_ge, p_d = pggan._ge_conv, pggan._d_conv

  def forced(orig):
    def layer(P_, scope, x, *a, **kw):
      out = orig(P_, scope, x, *a, **kw)
      pooled = isinstance(out, tuple)
      z = out[0] if pooled else out
      want = rec[scope].to(z.device)
      errs[scope] = float((z.float() - want).norm() / (want.norm() + 1e-30))
      zt = want.to(torch.bfloat16).contiguous()
      if pooled:
        pw = o_pool(rec[scope].double()).float().to(z.device)
        errs[scope + ' (pooled)'] = float((out[1].float() - pw).norm() / (pw.norm() + 1e-30))
        # what the oracle chain feeds the next layer: the pool of the ROUNDED output, rounded
        chain = o_pool(rnd(rec[scope]).double()).float().to(torch.bfloat16)
        return zt, chain.to(z.device).contiguous()
      return zt
    return layer
  pggan._ge_conv, pggan._d_conv = forced(p_ge), forced(p_d)
  try:
    with torch.no_grad():
      sd = s.float().to('cuda:0').to(torch.bfloat16).contiguous()
      net, ep = pggan.encoder_before_classification(tr.P, sd, 's', cfg)
      out, _ = pggan.generator(tr.P, net, 't', cfg, ep)
      pred, _ = pggan.discriminator(tr.P, out, cfg, 'discriminator_t')
  finally:
    pggan._ge_conv, pggan._d_conv = p_ge, p_d
  assert set(k for k in errs if not k.endswith('(pooled)')) == set(rec), set(rec) ^ set(errs)
  assert len(rec) == 13 + 15 + 15      # encoder, generator, discriminator conv layers at 256x256
  worst = max(errs.items(), key=lambda kv: kv[1])
  assert worst[1] < 1e-2, worst
  e = float((pred.double().cpu() - rpred).norm() / (rpred.norm() + 1e-30))
  assert e < 1e-2, ('prediction', e)


def test_progressive_stages_with_warm_start_match_oracle():
  """runner.run_progressive 4 -> 4to8 -> 8 -> 8to16 -> 16 (pggan_runner.py:82-160): every stage is a fresh trainer
  warm-started from the previous one (variables that exist in both, model_inheritor.py:576-644), growing stages fade in
  with alpha_grow = step / steps.  The oracle walks the same schedule -- its own stage list, its own warm-start rule (a
  dictionary merge), its own fade-in, torch_ref.train_step -- from the same fresh initialisations, inputs and
  gradient-penalty draws; after each stage the parameters are compared as UPDATES relative to the stage's start."""
  from twingan_amd import Config
  from twingan_amd.runner import run_progressive, stage_schedule
  from twingan_amd.twingan import Trainer
  base = Config(hw=4, max_ch=16, precision='fp32')
  table = {4: 2, 8: 2, 16: 2}
  steps = 2                                   # generator applies per stage; n_critic = 2 runs each
  sched = stage_schedule(4, 16, table, num_images_per_resolution=steps * 2)
  assert [s[0] for s in sched] == ['4', '4to8', '8', '8to16', '16']

  def batch(hw, bsz, call):                   # deterministic inputs and GP draws, the same on both sides
    g = torch.Generator().manual_seed(1000 * hw + call)
    return (torch.rand(bsz, hw, hw, 3, generator=g), torch.rand(bsz, hw, hw, 3, generator=g), torch.rand(bsz, generator=g),
            torch.rand(bsz, generator=g))
  calls = {}

  def batch_fn(hw, bsz):
    c = calls[hw] = calls.get(hw, -1) + 1
    return tuple(t.to('cuda:0') for t in batch(hw, bsz, c))
  ends = {}
  # the last stable stage trains "indefinitely" (pggan_runner.py:103-104): cap it like the others
  state, hist = run_progressive(base, batch_fn, 4, 16, table, num_images_per_resolution=steps * 2, device='cuda:0', seed=5,
                                max_steps_per_stage=steps,
                                on_stage_end=lambda name, tr: ends.__setitem__(name, tr.store.state_dict()))
  assert [h['stage'] for h in hist] == [s[0] for s in sched] and all(h['steps'] == steps for h in hist)

  # ---- the oracle's walk
  prev = None
  ocalls = {}
  for name, hw, growing, bsz, nsteps in sched:
    nsteps = min(nsteps, steps)
    cfg = dataclasses.replace(base, hw=hw, is_growing=growing, alpha_grow=0.0)
    fresh_tr = Trainer(cfg, device='cuda:0', seed=5)          # the product's fresh initialisation of this stage
    fresh = {k: v.double().cpu() for k, v in fresh_tr.store.state_dict().items()}
    fresh_tr.close()
    P = dict(fresh)
    if prev is not None:                                      # ignore_missing_vars warm start
      for k, v in prev.items():
        if k in P and tuple(P[k].shape) == tuple(v.shape):
          P[k] = v.clone()
    start = {k: v.clone() for k, v in P.items()}
    rcfg = R.Config(hw=hw, max_ch=base.max_ch, is_growing=growing, alpha_grow=0.0)
    opt = R.AdamState(P, rcfg)
    counter = 0
    for step in range(nsteps):
      if growing:
        rcfg.alpha_grow = step / nsteps
      for _ in range(2):
        c = ocalls[hw] = ocalls.get(hw, -1) + 1
        s, t, a_s, a_t = batch(hw, bsz, c)
        R.train_step(P, opt, s.double(), t.double(), rcfg, a_s.double().reshape(-1, 1, 1, 1), a_t.double().reshape(-1, 1, 1, 1),
                     counter=counter)
        counter += 1
    prev = {k: v.detach().clone() for k, v in P.items()}
    got = ends[name]
    assert set(got) == set(P), (name, set(got) ^ set(P))
    num = den = 0.0
    flipped = total = 0
    for k in P:
      d_dev = got[k].double().cpu() - start[k]
      d_ref = P[k].detach() - start[k]
      num += float(((d_dev - d_ref) ** 2).sum())
      den += float((d_ref ** 2).sum())
      flipped += int(((d_dev - d_ref).abs() > base.learning_rate).sum())      # a step of one weight went the other way
      total += d_ref.numel()
    e = np.sqrt(num / den)
    print('[progressive] stage %-5s update rel-L2 %.3e, weights off by more than one step: %d of %d' % (name, e, flipped, total))
    # Adam's first steps are sign-like: a gradient component near zero flips one weight's step by 2 lr on a last-bit
    # difference of a sum, and the flipped weights carry over the stages -- so the aggregate figure of the late stages moves
    # with the ORDER of the fp32 sums (measured 1.5e-5 / 2.4e-5 / 2.6e-5 / 2.2e-4 / 3.3e-2 in round 4; 1.5e-5 / 5.1e-4 /
    # 4.4e-4 / 6.6e-3 / 1.6e-1 after round 5 re-ordered the generator batch and fused the loss tails).  What must hold at every
    # stage: the early stages tight, and only a small FRACTION of the weights off by a whole step (a wrong gradient moves
    # all of them)
    assert e < (1e-3 if hw <= 8 else 0.3), (name, e)
    assert flipped < 0.02 * total, (name, flipped, total)


def test_progressive_run_through_checkpoint_files_equals_in_memory_and_resumes(tmp_path):
  """The same stage walk with the reference's directory protocol (TF-format checkpoints, checkpoint.py): equal to the
  in-memory warm start bit for bit, a finished run is skipped, and a stage interrupted after its first step resumes
  from its own checkpoint (optimiser slots, beta powers, global_step) and lands on the same parameters."""
  from twingan_amd import Config, checkpoint as C
  from twingan_amd.runner import run_progressive
  from twingan_amd.twingan import Trainer
  base = Config(hw=4, max_ch=8, precision='fp32')
  table = {4: 2, 8: 2}

  def batches():
    calls = {}

    def fn(hw, bsz):
      c = calls[hw] = calls.get(hw, -1) + 1
      g = torch.Generator().manual_seed(77 * hw + c)
      return tuple(t.to('cuda:0') for t in (torch.rand(bsz, hw, hw, 3, generator=g), torch.rand(bsz, hw, hw, 3, generator=g),
                                             torch.rand(bsz, generator=g), torch.rand(bsz, generator=g)))
    return fn
  kw = dict(start_hw=4, max_hw=8, hw_to_batch_size=table, num_images_per_resolution=4, device='cuda:0', seed=9,
            max_steps_per_stage=2)
  mem, _ = run_progressive(base, batches(), **kw)
  root = str(tmp_path / 'run')
  disk, hist = run_progressive(base, batches(), train_dir=root, **kw)
  assert [h['steps'] for h in hist] == [2, 2, 2]
  def same(a, b):      # the fp32 path is bit-reproducible (test_fp32_path_is_bit_reproducible)
    return set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)
  assert same(mem, disk)
  _, again = run_progressive(base, batches(), train_dir=root, **kw)
  assert all(h.get('skipped') for h in again)

  # ---- resume inside a stage: stage '4' stopped after its first generator apply
  tr = Trainer(dataclasses.replace(base, hw=4), device='cuda:0', seed=9)
  fn = batches()
  for _ in range(base.n_critic):
    tr.run(*fn(4, 2))
  part = str(tmp_path / 'resume')
  C.save(tr, os.path.join(part, '4'))
  assert C.latest_checkpoint(os.path.join(part, '4')).endswith('model.ckpt-1')
  tr.close()
  calls_done = base.n_critic

  def resumed_batches():
    inner = batches()
    skipped = {'n': 0}

    def fn(hw, bsz):
      while hw == 4 and skipped['n'] < calls_done:      # the data the interrupted run already consumed
        inner(4, bsz)
        skipped['n'] += 1
      return inner(hw, bsz)
    return fn
  res, hist = run_progressive(base, resumed_batches(), train_dir=part, **kw)
  assert hist[0]['steps'] == 2 and not hist[0].get('skipped')
  assert same(mem, res)


@pytest.mark.parametrize('norm', ['instance_norm', 'batch_norm'])
def test_fp32_path_is_bit_reproducible(norm):
  """The exact-parity (fp32) path has no order-dependent float atomic left: statistics, filter / bias / gamma / beta
  gradients, loss sums and per-sample sums are each taken by one workgroup or summed from per-workgroup partials in a
  fixed order (tg_common.h exact_path).  Two trainers fed the same data follow the same trajectory bit for bit."""
  from twingan_amd import Config
  from twingan_amd.twingan import Trainer
  cfg = Config(hw=32, max_ch=16, precision='fp32', generator_norm_type=norm)
  g = torch.Generator().manual_seed(3)
  data = [(torch.rand(3, 32, 32, 3, generator=g), torch.rand(3, 32, 32, 3, generator=g), torch.rand(3, generator=g),
           torch.rand(3, generator=g)) for _ in range(4)]
  ends = []
  for rep in range(2):
    tr = Trainer(cfg, device='cuda:0', seed=4)
    for s, t, a_s, a_t in data:
      tr.run(s.cuda(), t.cuda(), a_s.cuda(), a_t.cuda())
    torch.cuda.synchronize()
    ends.append((tr.store.state_dict(include_state=True), {k: (m, v) for k, (m, v) in tr.store.adam_dict().items()}))
    tr.close()
  (pa, sa), (pb, sb) = ends
  bad = [k for k in pa if not torch.equal(pa[k], pb[k])]
  assert not bad, (len(bad), bad[:5])
  bad = [k for k in sa if not (torch.equal(sa[k][0], sb[k][0]) and torch.equal(sa[k][1], sb[k][1]))]
  assert not bad, (len(bad), bad[:5])


class _Deterministic:
  """tg_set_deterministic(1) inside the block (process-wide switch of the C ABI), the previous setting afterwards."""

  def __enter__(self):
    from twingan_amd import _lib
    self.lib = _lib.load()
    self.was = self.lib.tg_set_deterministic(1)

  def __exit__(self, *exc):
    self.lib.tg_set_deterministic(self.was)


def _trajectory(prec, scale, graph, steps=4, **kw):
  from twingan_amd import Config
  from twingan_amd.twingan import Trainer
  g = torch.Generator().manual_seed(8)
  s, t = torch.rand(4, 32, 32, 3, generator=g), torch.rand(4, 32, 32, 3, generator=g)
  dt = dict(fp16=torch.float16, bf16=torch.bfloat16, fp32=torch.float32)[prec]
  tr = Trainer(Config(hw=32, max_ch=32, precision=prec, loss_scale=scale, **kw), device='cuda:0', seed=3, use_graph=graph)
  p0 = {k: v.clone() for k, v in tr.store.state_dict().items()}
  torch.manual_seed(11)                                   # the device RNG draws the GP alphas
  for _ in range(steps):
    loss, terms = tr.run(s.cuda().to(dt), t.cuda().to(dt))
    assert torch.isfinite(loss).all() and all(torch.isfinite(v).all() for v in terms.values())
  assert not graph or tr.graph_fallback_reason is None, tr.graph_fallback_reason
  sd = {k: v.clone() for k, v in tr.store.state_dict(include_state=True).items()}
  assert all(torch.isfinite(v).all() for v in sd.values())
  tr.close()
  return p0, sd


def _update_err(a, b, p0):
  num = sum(float((((a[k] - p0[k]) - (b[k] - p0[k])).double() ** 2).sum()) for k in p0)
  den = sum(float(((b[k] - p0[k]).double() ** 2).sum()) for k in p0)
  return (num / den) ** 0.5


def test_fp16_training_steps_with_loss_scale_and_graph():
  """precision='fp16' (TG_F16 storage, the reference's --dataset_dtype float16) with the static loss scale 128 of
  model_deploy.py:308-313 / model_inheritor.py:568-570: the scaled backward stays finite in half precision, Adam sees the
  unscaled gradients (tg_adam_step divides by the scale), eager and hipGraph steps agree with each other, and the
  parameters move like the fp32 trainer's from the same start.

  Graph vs eager.  Round 2 asked the default 16-bit path for 1e-6 here and the driver measured 3.26e-3.  Root cause
  (tools/fp16_repro.py, gpurun_out r3a): TWO EAGER fp16 runs already differ by exactly that 3.26e-3 (32-70 of the
  parameter tensors not bit-identical) -- the default 16-bit path ends its bias / gamma / beta / loss sums in fp32 atomics
  in arrival order, a last-ulp difference that Adam's sign-like first steps turn into 2 lr on a few weights; graph
  replays differ from eager launches by the same amount, no more.  In deterministic mode (tg_set_deterministic, every such
  sum in a fixed order) eager and graph trajectories are bit-identical: that is the assertion that catches a capture bug
  (a stale pack, a baked alpha, an RNG offset).  The default mode keeps a bound at its own run-to-run noise."""
  with _Deterministic():
    p0, eager = _trajectory('fp16', 128.0, False)
    _, graph = _trajectory('fp16', 128.0, True)
    _, unscaled = _trajectory('fp16', 1.0, False)
  bad = [k for k in eager if not torch.equal(eager[k], graph[k])]
  assert not bad, ('deterministic fp16: hipGraph replay left %d tensors different from eager launches' % len(bad), bad[:5])
  _, f32 = _trajectory('fp32', 1.0, False)
  _, eager_nd = _trajectory('fp16', 128.0, False)
  _, graph_nd = _trajectory('fp16', 128.0, True)
  e_graph = _update_err(graph_nd, eager_nd, p0)
  e_scale = _update_err(eager, unscaled, p0)
  e_f32 = _update_err(eager, f32, p0)
  print('[fp16] update rel-L2: graph vs eager (atomics mode) %.3e, scale 128 vs 1 %.3e, fp16 vs fp32 %.3e' % (e_graph, e_scale, e_f32))
  # Adam's first steps are sign-like, so the storage rounding (and the rounding pattern a different loss scale gives)
  # moves a fraction of the weights by 2 lr.  The atomics-order noise of the default mode is amplified the same way and
  # comes in DISCRETE levels -- a last-ulp difference either flips the first update of a weight whose gradient is ~0 or it
  # does not: round 3 measured 3.3e-3 run to run; round 4 (profiles/r04_c_fp16_noise_by_switch.txt: 3 eager + 3 graph
  # runs against eager #0, per kernel switch) 7e-9, 7.0e-4 or 3.26e-2, EAGER and GRAPH runs alike, and 1.4e-8 throughout
  # when the generator's concat backward rounds twice (TG_UPCAT_BWD_FUSED=0) -- which weights sit on the knife edge
  # depends on the last bit of every kernel's rounding.  The bound is therefore the largest level seen x 3; what pins
  # the capture is the bit-equality in deterministic mode above.
  assert e_graph < 1e-1 and e_scale < 0.35 and e_f32 < 0.5, (e_graph, e_scale, e_f32)


@pytest.mark.parametrize('prec,kw', [
    ('bf16', {}), ('fp16', {}), ('bf16', dict(generator_norm_type='batch_norm', loss_architecture='hinge')),
    ('fp16', dict(spectral_norm=True, do_self_attention=True, self_attention_hw=16))])
def test_deterministic_mode_makes_16_bit_training_bit_reproducible(prec, kw):
  """tg_set_deterministic(1) (TG_DETERMINISTIC=1): the 16-bit storage types take the fixed-order sums the fp32 parity
  path always takes (tg_common.h exact_grid: one workgroup per channel / sample / tensor sum, per-image partials added in
  image order, bias gradients outside the filter-gradient kernel) -- two eager trajectories and a hipGraph trajectory
  of four G/D runs end on the same bits, parameters and non-trainable state alike.  Measured cost of the mode on the
  bench step: DESIGN.md section 8c."""
  with _Deterministic():
    scale = 128.0 if prec == 'fp16' else 1.0
    _, a = _trajectory(prec, scale, False, **kw)
    _, b = _trajectory(prec, scale, False, **kw)
    _, c = _trajectory(prec, scale, True, **kw)
  for other, what in ((b, 'a second eager run'), (c, 'the hipGraph replay')):
    bad = [k for k in a if not torch.equal(a[k], other[k])]
    assert not bad, ('%s: %d tensors differ from the first eager run in %s' % (prec, len(bad), what), bad[:5])


def test_config4_half_precision_flash_path_matches_composed_path_and_oracle():
  """BASELINE configs[4] at model level in ITS dtype (fp16 storage, loss scale 128): spectral norm on the discriminator
  convs + self-attention in E / G / D + WGAN-GP.  The discriminator loss differentiates the attention twice (gradient
  penalty, image_generation.py:414-439), the generator loss once -- with the flash kernels (tg_flash_attention_fwd / _bwd /
  _bwd_bwd, the path bench.py --config 4 runs), with the composed batched-GEMM / softmax path (attention.hip, which
  materialises the [n, hw^2, hw^2] map), and in the float64 oracle on the same fp16-rounded inputs.  Bounds: loss terms
  5e-2 relative; aggregate gradients within the fp16 storage sensitivity of this graph (tools/bf16_sensitivity.py: 0.10 for
  fp16 at 32 x 32; measured here: see the printed figures) of the oracle, and the two HIP paths closer to each other than
  either is to the oracle allows."""
  from twingan_amd import ops, pggan
  from twingan_amd import twingan as T
  kw = dict(hw=32, max_ch=64, spectral_norm=True, do_self_attention=True, self_attention_hw=16, loss_architecture='wgan_gp',
            loss_scale=128.0)
  cfg, rcfg, tr, Pref, dev, ref = make(kw, 'fp16', seed=6, batch=2)
  rcfg.sn_state = R.init_sn_state(Pref, seed=3)
  sn0 = {k: v.float() for k, v in rcfg.sn_state.items()}
  for k, v in sn0.items():
    rcfg.sn_state[k] = v.double()
  for v in Pref.values():
    v.requires_grad_(True)
  # ---- oracle: generator loss, then discriminator loss, from the same pre-run u (no end_run in between)
  rgl, rgterms = R.generator_loss(Pref, ref['s'], ref['t'], rcfg)
  rgl.backward()
  ref_g = {k: Pref[k].grad.numpy().copy() for k in tr.store.names('g') if Pref[k].grad is not None}
  for v in Pref.values():
    v.grad = None
  if rcfg.sn_cache:                      # drop the run's normalised kernels (their graph is spent) but keep the pre-run u
    rcfg.sn_cache.clear()
  for k, v in sn0.items():
    rcfg.sn_state[k] = v.double()
  rdl, rdterms = R.discriminator_loss(Pref, ref['s'], ref['t'], rcfg, ref['a_s'], ref['a_t'])
  rdl.backward()
  ref_d = {k: Pref[k].grad.numpy().copy() for k in tr.store.names('d') if Pref[k].grad is not None}

  def hip_pass(flash):
    saved = ops.USE_FLASH_ATTENTION
    ops.USE_FLASH_ATTENTION = flash
    launched = []
    try:
      out = {}
      for grp in ('g', 'd'):
        for k, v in sn0.items():
          tr.store.state[k].copy_(v)
        tr.store.zero_grad(grp)
        tr._set_requires_grad(g=grp == 'g', d=grp == 'd')
        if grp == 'g':
          loss, terms = T.generator_loss(tr.P, dev['s'], dev['t'], cfg)
        else:
          loss, terms = T.discriminator_loss(tr.P, dev['s'], dev['t'], cfg, dev['a_s'], dev['a_t'])
        (loss * cfg.loss_scale).backward()
        pggan.end_run(tr.P)                # drops the run's normalised kernels; u is reset from sn0 above
        torch.cuda.synchronize()
        grads = {k: (v.double().cpu().numpy() / cfg.loss_scale) for k, v in tr.store.grad_dict().items()
                 if k in tr.store.names(grp)}
        assert all(np.isfinite(v).all() for v in grads.values()), (grp, flash)
        out[grp] = ({k: float(v) for k, v in terms.items()}, grads)
      return out
    finally:
      ops.USE_FLASH_ATTENTION = saved

  def agg(a, b):      # aggregate rel-L2 and cosine of gradient dictionaries a vs b (over b's keys)
    num = sum(np.sum((a[k] - b[k]) ** 2) for k in b)
    den = sum(np.sum(b[k] ** 2) for k in b)
    dot = sum(np.sum(a[k] * b[k]) for k in b)
    na = sum(np.sum(a[k] ** 2) for k in b)
    return float(np.sqrt(num / den)), float(dot / np.sqrt(na * den))

  # the flash node must actually be on the tape of the first pass and absent from the second
  seen = []
  orig_call = ops.call

  def spy(name, *a, **k):
    seen.append(name)
    return orig_call(name, *a, **k)
  ops.call = spy
  try:
    flash = hip_pass(True)
    n_flash = {n: seen.count(n) for n in ('tg_flash_attention_fwd', 'tg_flash_attention_bwd', 'tg_flash_attention_bwd_bwd')}
    del seen[:]
    composed = hip_pass(False)
    assert not any(n.startswith('tg_flash') for n in seen)
  finally:
    ops.call = orig_call
  assert all(v > 0 for v in n_flash.values()), n_flash      # forward, first-order and second-order kernels all ran

  # what fp16 STORAGE ROUNDING ALONE does to the float64 gradients of this graph on these inputs (oracle/rounding.py: the
  # oracle with fp16 rounding at the kernels' storage points, incl. the normalised kernels' packs and the attention's stored
  # tensors): the bound on the kernels is 1.5 x that figure, per loss group
  from oracle import rounding

  def sn_reset():
    if rcfg.sn_cache:
      rcfg.sn_cache.clear()
    for k, v in sn0.items():
      rcfg.sn_state[k] = v.double()
  P0 = {k: v.detach() for k, v in Pref.items()}
  gnames = [k for k in tr.store.names('g') if k in ref_g]
  dnames = [k for k in tr.store.names('d') if k in ref_d]
  model = {
      'g': rounding.gradient_sensitivity(P0, gnames, lambda Q: R.generator_loss(Q, ref['s'], ref['t'], rcfg)[0], torch.float16,
                                         reset=sn_reset),
      'd': rounding.gradient_sensitivity(P0, dnames, lambda Q: R.discriminator_loss(Q, ref['s'], ref['t'], rcfg, ref['a_s'], ref['a_t'])[0],
                                         torch.float16, reset=sn_reset)}
  # The discriminator loss of a fresh network is a DIFFERENCE of nearly equal parts (WGAN: mean D(fake) - mean D(real), the
  # two within 1 % of each other here), and a 16-bit forward flips the LeakyReLU mask of the few units whose pre-activation
  # is ~0 -- one flipped unit under the FC layer moves the NET gradient of the tail by tens of percent, in the kernels and in
  # the rounded oracle alike but not for the same unit (tools/diag_c4_dgroup.py, seeds 6-9: kernels 0.009 ... 0.23 against
  # the net gradient where the rounded oracle shows 0.010 ... 0.024, either one ahead).  The statement that is stable is the
  # deviation relative to the PARTS that are added up: sqrt(sum over loss parts of |gradient of the part|^2), float64 oracle.
  def part_norm(grp):
    sn_reset()
    Q = {k: v.detach().clone().requires_grad_(True) for k, v in P0.items()}
    if grp == 'g':
      _, tt = R.generator_loss(Q, ref['s'], ref['t'], rcfg)
      parts, names = list(tt.values()), gnames
    else:
      _, tt = R.discriminator_loss(Q, ref['s'], ref['t'], rcfg, ref['a_s'], ref['a_t'])
      parts, names = [v for k, v in tt.items() if 'gradient_penalty' in k], dnames
      with torch.no_grad():
        o = R.forward_generators(Q, ref['s'], ref['t'], rcfg)
      for d, real, prime in (('s', ref['s'], o['s_prime']), ('t', ref['t'], o['t_prime'])):      # hw 32: no cycle-GAN terms
        parts.append(R.discriminator(Q, real, rcfg, 'discriminator_' + d)[0].mean())
        parts.append(R.discriminator(Q, prime, rcfg, 'discriminator_' + d)[0].mean())
    tot = 0.0
    for pt in parts:
      gs = torch.autograd.grad(pt, [Q[k] for k in names], retain_graph=True, allow_unused=True)
      tot += sum(float((x ** 2).sum()) for x in gs if x is not None)
    return tot ** 0.5
  sn_reset()
  for grp, rterms, rgrads in (('g', rgterms, ref_g), ('d', rdterms, ref_d)):
    pn = part_norm(grp)
    _, rnd_g, ex_g = model[grp]
    m_abs = sum(float(((rnd_g[k] - ex_g[k]) ** 2).sum()) for k in ex_g) ** 0.5
    for name, res in (('flash', flash), ('composed', composed)):
      terms, grads = res[grp]
      for k in rterms:
        want = float(rterms[k])
        assert abs(terms[k] - want) < 5e-2 * abs(want) + 2e-2, (grp, name, k, terms[k], want)
      e, cos = agg(grads, rgrads)
      e_abs = float(np.sqrt(sum(np.sum((grads[k] - rgrads[k]) ** 2) for k in rgrads)))
      print('[config4 fp16] %s %s: gradients vs oracle rel-L2 %.3e (net) cosine %.5f; relative to the loss parts %.3e, fp16 storage '
            'rounding alone %.3e (net %.3e)' % (grp, name, e, cos, e_abs / pn, m_abs / pn, model[grp][0]))
      # round 3 held the net figure to a fixed 0.25 (measured g 0.108 / 0.099, d 0.1435 / 0.1433); it stays as the coarse
      # bound, the statement proper is: no further from the float64 gradients than 1.5 x what fp16 storage rounding alone
      # does to the oracle, relative to the parts
      assert e < 0.25 and cos > 0.97, (grp, name, e, cos)
      # + 1e-2: the flipped-unit floor -- one unit under the FC layer is ~1/128 of a part; measured here: g 0.113 / 0.104 for a
      # model figure of 0.110, d 8.0e-3 for 1.2e-3 (the kernels' forward is deterministic, so these do not move run to run)
      assert e_abs / pn < 1.5 * m_abs / pn + 1e-2, (grp, name, e_abs / pn, m_abs / pn)
    e, cos = agg(flash[grp][1], composed[grp][1])
    print('[config4 fp16] %s: flash vs composed rel-L2 %.3e cosine %.5f' % (grp, e, cos))
    # the two HIP paths are closer to each other than either is to the oracle: measured g 0.068, d 0.013
    assert e < 0.15 and cos > 0.99, (grp, e, cos)
  tr.close()


@pytest.mark.parametrize('hw,growing,k', [(16, False, 7), (8, True, 4)])
def test_larger_filter_at_rgb_layer_matches_oracle(hw, growing, k):
  """--use_larger_filter_at_rgb_layer (nets/pggan.py:47-50,172-175,194-197): the to-RGB kernels are min(7, hw / 2) instead
  of 1 x 1 -- 7 x 7 at 16 x 16; at the growing 8 x 8 stage BOTH to-RGB layers get an even 4 x 4 SAME kernel (TF pads 1 low,
  2 high).  The oracle is pinned against the reference's own code for both cases
  (tests/test_reference_live.py::larger_rgb_16 / larger_rgb_growing_8); here the HIP path (direct kernels: the MFMA
  kernels take 1 x 1 / 3 x 3 / dense 4 x 4 only) against the oracle: variables, generated images, generator loss and
  gradients."""
  from twingan_amd import twingan as T
  kw = dict(hw=hw, max_ch=8, use_larger_filter_at_rgb_layer=True, is_growing=growing, alpha_grow=0.4 if growing else 0.0)
  cfg, rcfg, tr, Pref, dev, ref = make(kw, 'fp32', seed=9, batch=2)
  rcfg.alpha_grow = cfg.alpha_grow
  assert set(tr.store.state_dict()) == set(Pref)
  rgb = [n for n in Pref if 'generator_to_rgb' in n and n.endswith('/weights')]
  assert len(rgb) == (2 if growing else 1) and all(tuple(Pref[n].shape[:2]) == (k, k) for n in rgb), [(n, Pref[n].shape) for n in rgb]
  for v in Pref.values():
    v.requires_grad_(True)
  tr.store.zero_grad('g')
  tr._set_requires_grad(g=True, d=False)
  gl, gterms = T.generator_loss(tr.P, dev['s'], dev['t'], cfg)
  rgl, rterms = R.generator_loss(Pref, ref['s'], ref['t'], rcfg)
  assert set(gterms) == set(rterms)
  for name in rterms:
    assert abs(gterms[name].item() - rterms[name].item()) < 1e-4 * max(1.0, abs(rterms[name].item())), name
  gl.backward()
  rgl.backward()
  _grads_close(tr, Pref, tr.store.names('g'), FP32_GRAD_TOL, 'generator, larger RGB filter', var_tol=FP32_VAR_GRAD_TOL)
  for n in rgb:      # the gradient of the large kernel itself, tap by tap
    assert rel_l2(tr.store.grad_dict()[n], Pref[n].grad) < 1e-4, n
  tr.close()


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
def test_discriminator_sign_bit_path_equals_the_tensor_path(precision):
  """The discriminators' block-end convs hand the pool and the LeakyReLU sign bits over instead of their full-resolution
  output (ops.Conv2dPoolSignsFn, first-order passes; the gradient-penalty pass keeps z).  Same loss terms bit for bit and
  the same gradients as with TG_POOL_SIGNS=0 (bias sums: fp32 atomics order), generator step and discriminator step."""
  from twingan_amd import ops
  from twingan_amd import twingan as T
  cfg, rcfg, tr, Pref, dev, ref = make(dict(hw=64, max_ch=32), precision, seed=4, batch=2)
  out = {}
  seen = []
  orig_call = ops.call

  def spy(name, *a, **k):
    seen.append(name)
    return orig_call(name, *a, **k)
  for signs in (True, False):
    saved, ops.USE_POOL_SIGNS = ops.USE_POOL_SIGNS, signs
    ops.call = spy
    del seen[:]
    try:
      res = {}
      for grp in ('g', 'd'):
        tr.store.zero_grad(grp)
        tr._set_requires_grad(g=grp == 'g', d=grp == 'd')
        if grp == 'g':
          loss, terms = T.generator_loss(tr.P, dev['s'], dev['t'], cfg)
        else:
          loss, terms = T.discriminator_loss(tr.P, dev['s'], dev['t'], cfg, dev['a_s'], dev['a_t'])
        loss.backward()
        res[grp] = ({k: float(v) for k, v in terms.items()}, {k: v.clone() for k, v in tr.store.grad_dict().items()
                                                               if k in tr.store.names(grp)})
      out[signs] = res
      n_sig = seen.count('tg_conv2d_fwd_pool_signs')
      # the backward reads the sign bytes in the block end's own backward-data kernel (round 4) or in tg_lrelu_pool_bwd_signs
      n_bwd = seen.count('tg_lrelu_pool_bwd_signs') + seen.count('tg_conv2d_bwd_data_unpool')
      assert (n_sig > 0) == signs and (n_bwd > 0) == signs, (signs, n_sig, n_bwd)
      assert seen.count('tg_conv2d_fwd_pool') > 0      # the gradient-penalty pass keeps the tensor either way
    finally:
      ops.call = orig_call
      ops.USE_POOL_SIGNS = saved
  for grp in ('g', 'd'):
    ta, tb = out[True][grp][0], out[False][grp][0]
    # the forward values are the same tensors; the gradient-penalty term (which does not use the sign-bit path at all)
    # ends in fp32 atomics on the 16-bit paths: last-ulp run-to-run noise
    assert set(ta) == set(tb) and all(abs(ta[k] - tb[k]) <= 2e-6 * max(1.0, abs(tb[k])) for k in tb), (grp, ta, tb)
    num = sum(float(((out[True][grp][1][k] - out[False][grp][1][k]).double() ** 2).sum()) for k in out[True][grp][1])
    den = sum(float((out[False][grp][1][k].double() ** 2).sum()) for k in out[True][grp][1])
    assert (num / den) ** 0.5 < 1e-5, (grp, (num / den) ** 0.5)
  tr.close()


@pytest.mark.parametrize('precision', ['fp32', 'bf16', 'fp16'])
def test_discriminator_pair_equals_the_two_separate_discriminators(precision):
  """From 32 x 32 down the two discriminators run as ONE batch through grouped convs (pggan.discriminator_pair: the stacked
  twin variables of ParamStore.pairs, TgConvDesc.groups = 2 -- half the launches of the two towers the reference builds,
  twingan.py:105-110, image_generation.py:348-439).  Every loss term and every gradient -- generator step (through the
  frozen discriminators) and discriminator step (batched pass + the gradient penalty's double backward) -- equals the
  per-domain path's: the forward values bit for bit (per-image arithmetic), gradients to fp32 summation order."""
  from twingan_amd import ops, pggan
  from twingan_amd import twingan as T
  cfg, rcfg, tr, Pref, dev, ref = make(dict(hw=64, max_ch=32), precision, seed=5, batch=2)
  saved, pggan.USE_DISCRIMINATOR_PAIR = pggan.USE_DISCRIMINATOR_PAIR, True      # off by default (measured: see pggan.py)
  try:
    assert pggan.discriminator_pair_supported(tr.P, cfg, cfg.hw)
  finally:
    pggan.USE_DISCRIMINATOR_PAIR = saved
  assert tr.P.pairs['discriminator_*/encoder_block_32x32x32/Conv/weights'].shape == (2, 3, 3, 32, 32)
  out, seen = {}, []
  orig_call = ops.call

  def spy(name, *a, **k):
    seen.append(name)
    return orig_call(name, *a, **k)
  for pair in (True, False):
    saved, pggan.USE_DISCRIMINATOR_PAIR = pggan.USE_DISCRIMINATOR_PAIR, pair
    ops.call = spy
    del seen[:]
    try:
      res = {}
      for grp in ('g', 'd'):
        tr.store.zero_grad(grp)
        tr._set_requires_grad(g=grp == 'g', d=grp == 'd')
        if grp == 'g':
          loss, terms = T.generator_loss(tr.P, dev['s'], dev['t'], cfg)
        else:
          loss, terms = T.discriminator_loss(tr.P, dev['s'], dev['t'], cfg, dev['a_s'], dev['a_t'])
        loss.backward()
        res[grp] = ({k: float(v) for k, v in terms.items()}, {k: v.clone() for k, v in tr.store.grad_dict().items()
                                                               if k in tr.store.names(grp)})
      out[pair] = (res, seen.count('tg_conv2d_fwd') + seen.count('tg_conv2d_fwd_pool') + seen.count('tg_conv2d_fwd_pool_signs'))
    finally:
      ops.call = orig_call
      pggan.USE_DISCRIMINATOR_PAIR = saved
  assert out[True][1] < out[False][1], (out[True][1], out[False][1])      # fewer conv calls with the pair
  for grp in ('g', 'd'):
    ta, tb = out[True][0][grp][0], out[False][0][grp][0]
    # losses: same forward tensors; the sums end in fp32 atomics on the 16-bit paths (last-ulp run-to-run noise)
    assert set(ta) == set(tb) and all(abs(ta[k] - tb[k]) <= 2e-6 * max(1.0, abs(tb[k])) for k in tb), (grp, ta, tb)
    ga, gb = out[True][0][grp][1], out[False][0][grp][1]
    num = sum(float(((ga[k] - gb[k]).double() ** 2).sum()) for k in ga)
    den = sum(float((gb[k].double() ** 2).sum()) for k in ga)
    # fp32: fixed summation orders; 16-bit: the filter / bias gradients end in fp32 atomics (measured 1.9e-5 in fp16)
    assert (num / den) ** 0.5 < (1e-5 if precision == 'fp32' else 5e-5), (grp, (num / den) ** 0.5)
    worst = max(rel_l2(ga[k], gb[k]) for k in ga if float(gb[k].abs().max()) > 0)
    assert worst < 1e-3, (grp, worst)
  tr.close()


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_rccl_segmented_capture_one_rank(precision):
  """The path the first real 8-GPU run takes (deployment/model_deploy.py:265-268,473-503 replaced by it): a torch.distributed
  "nccl" (= RCCL) process group alive in the process, the backward cut into segments (ops.Cuts) and captured as one hipGraph
  per segment in thread_local mode, an asynchronous all-reduce of each segment's range of the flat gradient buffer between
  the replays, the wait before the apply graph.  With ONE rank (all a 1-GPU box allows; a one-rank sum is the identity) the
  parameters after three G+D steps must equal the unsegmented single-graph trainer's (fp32: to summation order) -- with
  collectives actually issued and no fallback to an unsegmented capture.  tools/rccl_smoke.py in a child process: the process
  group must not leak into the other tests."""
  import json
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
  env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
  r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'rccl_smoke.py'), '--precision', precision, '--steps', '3'],
                     capture_output=True, text=True, timeout=900, env=env)
  assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
  d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
  assert d['backend'] == 'nccl' and d['world'] == 1 and d['segments'] == {'g': 3, 'd': 2}
  assert d['use_graph'] and d['capture_note'] is None, d
  # six runs: 3 generator applies x 3 ranges + 3 discriminator applies x 2 ranges (every byte of both gradient buffers once
  # per apply), plus the eager warm-up run of each step kind that precedes its capture (on a snapshot: 3 + 2 more)
  assert d['collectives'] == 3 * 3 + 3 * 2 + 3 + 2 and d['finishes'] >= 6 and d['allreduce_bytes'] > 0, d
  if precision == 'fp32':
    # same kernels on the same tensors; the order in which contributions reach a gradient sink differs with the cut
    # (test_segmented_backward_leaves_the_same_gradients: <= 1e-4 per gradient), then Adam's sign-like first steps
    assert d['params_rel_l2'] < 1e-5, d      # measured 6.5e-8 (40 of 148 tensors not bit-equal)
  else:
    assert d['params_rel_l2'] < 1e-3, d      # measured 2.6e-6: 16-bit run-to-run noise (fp32 atomics order)
