"""Pins the oracle: float64 NumPy restatement <-> torch-CPU restatement, plus the analytic
known-answer tests of SURVEY.md Appendix B.  (These cover the TensorFlow-op level, which no
TensorFlow run pins here; the composition is pinned against the reference's own code by tests/test_golden.py --
see oracle/__init__.py.)"""
import math

import numpy as np
import pytest
import torch

from oracle import np_ops as N
from oracle import torch_ref as R

RNG = np.random.RandomState(0)


def t64(a):
  return torch.from_numpy(np.asarray(a, np.float64))


# ---------------------------------------------------------------- channel schedule / shapes
def test_channel_schedule():
  # stage -> C: 0:256 1:256 2:256 3:128 4:64 5:32 6:16 (7:8)
  assert [N.get_num_channels(s) for s in range(8)] == [256, 256, 256, 128, 64, 32, 16, 8]
  assert [R.get_num_channels(s, 512) for s in range(3)] == [512, 512, 256]


def test_param_counts_match_survey():
  P = R.init_params(R.Config(hw=256))
  count = lambda top: sum(v.numel() for k, v in P.items() if k.startswith(top + '/') and 'InstanceNorm' not in k)
  assert count('encoder_content') == 2946864     # 2.95 M (3*16 + sum 9(c^2 + c*c_prev))
  assert count('generator') == 5697840           # 5.70 M
  assert count('discriminator_s') == 4590097     # 4.59 M
  assert P['discriminator_t/before_fc_1x1x256/Conv/weights'].shape == (3, 3, 257, 256)
  assert P['generator/block_64x64x64/Conv/weights'].shape == (3, 3, 256, 64)
  assert P['encoder_content/from_rgb_256x256/Conv/weights'].shape == (1, 1, 3, 16)


# ---------------------------------------------------------------- conv: numpy <-> torch, KATs
@pytest.mark.parametrize('k,pad,cin,cout,hw', [(3, 'SAME', 5, 7, 6), (1, 'SAME', 3, 4, 5), (4, 'VALID', 3, 5, 4),
                                               (4, 'VALID', 2, 3, 7), (4, 'SAME', 2, 3, 5)])
def test_conv_np_vs_torch(k, pad, cin, cout, hw):
  x = RNG.randn(2, hw, hw, cin)
  w = RNG.randn(k, k, cin, cout)
  y = N.conv2d(x, w, pad)
  xt, wt = t64(x).requires_grad_(True), t64(w).requires_grad_(True)
  yt = R.conv2d(xt, wt, pad)
  np.testing.assert_allclose(y, yt.detach().numpy(), rtol=1e-12, atol=1e-12)
  gy = RNG.randn(*y.shape)
  gx, gw = torch.autograd.grad(yt, [xt, wt], t64(gy))
  np.testing.assert_allclose(N.conv2d_bwd_data(gy, w, (hw, hw), pad), gx.numpy(), rtol=1e-11, atol=1e-11)
  np.testing.assert_allclose(N.conv2d_bwd_weight(x, gy, (k, k), pad), gw.numpy(), rtol=1e-11, atol=1e-11)


def test_conv_delta_kernel_is_shift():
  x = RNG.randn(1, 5, 5, 2)
  w = np.zeros((3, 3, 2, 2))
  w[1, 1] = np.eye(2)
  np.testing.assert_allclose(N.conv2d(x, w), x)
  w = np.zeros((3, 3, 2, 2))
  w[0, 1] = np.eye(2)                               # tap (dy=-1): out[y] = in[y-1]
  y = N.conv2d(x, w)
  np.testing.assert_allclose(y[:, 1:], x[:, :-1])
  np.testing.assert_allclose(y[:, 0], 0)


def test_conv1x1_is_matmul():
  x = RNG.randn(2, 3, 3, 4)
  w = RNG.randn(1, 1, 4, 6)
  np.testing.assert_allclose(N.conv2d(x, w), x @ w[0, 0], rtol=1e-13)


def test_conv4x4_valid_on_4x4_is_fc():
  x = RNG.randn(3, 4, 4, 5)
  w = RNG.randn(4, 4, 5, 6)
  np.testing.assert_allclose(N.conv2d(x, w, 'VALID')[:, 0, 0], x.reshape(3, -1) @ w.reshape(-1, 6), rtol=1e-12)


# ---------------------------------------------------------------- pointwise / norm KATs
def test_lrelu():
  np.testing.assert_allclose(N.leaky_relu(np.array([-2.0, 0.0, 3.0])), [-0.4, 0.0, 3.0])


def test_pixel_norm_constant_vector():
  c = 3.0
  x = np.full((1, 2, 2, 8), c)
  np.testing.assert_allclose(N.pixel_norm(x), 1.0 / math.sqrt(1.0 + 1e-6 / c ** 2), rtol=1e-12)
  np.testing.assert_allclose(N.pixel_norm(-x), -1.0 / math.sqrt(1.0 + 1e-6 / c ** 2), rtol=1e-12)


def test_instance_norm_checkerboard():
  x = np.indices((4, 4)).sum(0) % 2                # values {0,1}: mean .5, biased var .25
  x = x[None, :, :, None].astype(np.float64)
  y = N.instance_norm(x, gamma=2.0, beta=0.5)
  r = 1.0 / math.sqrt(0.25 + 1e-6)
  np.testing.assert_allclose(y, (x - 0.5) * r * 2.0 + 0.5, rtol=1e-12)


def test_norms_np_vs_torch():
  x = RNG.randn(3, 6, 6, 5)
  g, b = RNG.randn(5), RNG.randn(5)
  np.testing.assert_allclose(N.instance_norm(x, g, b), R.instance_norm(t64(x), t64(g), t64(b)).numpy(), rtol=1e-11,
                             atol=1e-12)
  np.testing.assert_allclose(N.pixel_norm(x), R.pixel_norm(t64(x)).numpy(), rtol=1e-12)
  np.testing.assert_allclose(N.leaky_relu(x), R.leaky_relu(t64(x)).numpy(), rtol=1e-15)
  np.testing.assert_allclose(N.upsample2x(x), R.upsample2x(t64(x)).numpy())
  np.testing.assert_allclose(N.avg_pool2(x), R.avg_pool2(t64(x)).numpy(), rtol=1e-13)
  x4 = RNG.randn(5, 4, 4, 6)
  np.testing.assert_allclose(N.minibatch_state_concat(x4), R.minibatch_state_concat(t64(x4)).numpy(), rtol=1e-12)


def test_mbstd_identical_samples():
  x = np.tile(RNG.randn(1, 4, 4, 3), (5, 1, 1, 1))
  y = N.minibatch_state_concat(x)
  assert y.shape == (5, 4, 4, 4)
  np.testing.assert_allclose(y[..., 3], math.sqrt(1e-8), rtol=1e-9)


def test_upsample_avgpool_identity():
  x = RNG.randn(2, 3, 3, 4)
  np.testing.assert_allclose(N.avg_pool2(N.upsample2x(x)), x, rtol=1e-14)
  up = N.upsample2x(x)
  assert up[0, 3, 5, 1] == x[0, 1, 2, 1]


def test_lerp_endpoints():
  a, b = RNG.randn(4), RNG.randn(4)
  np.testing.assert_allclose(N.lerp(a, b, 1.0), a)
  np.testing.assert_allclose(N.lerp(a, b, 0.0), b)


# ---------------------------------------------------------------- losses / optimiser KATs
def test_gp_of_unit_linear_critic_is_zero():
  w = RNG.randn(4, 4, 3)
  w /= np.linalg.norm(w)
  g = np.tile(w[None], (5, 1, 1, 1))               # grad of <w, x> wrt x is w for every sample
  assert abs(N.gradient_penalty(g, 10.0)) < 1e-24
  assert abs(N.gradient_penalty(2 * g, 10.0) - 10.0) < 1e-12


def test_adam_first_step_is_sign():
  th, g = RNG.randn(10), RNG.randn(10)
  th1, m, v = N.adam_step(th, g, np.zeros(10), np.zeros(10), 1, lr=1e-4)
  np.testing.assert_allclose(th1 - th, -1e-4 * np.sign(g), rtol=1e-5)


def test_adam_np_vs_torch_shared_counter():
  cfg = R.Config()
  P = {'a': t64(RNG.randn(6)), 'b': t64(RNG.randn(4))}
  ref = {k: v.numpy().copy() for k, v in P.items()}
  m = {k: np.zeros_like(v) for k, v in ref.items()}
  v_ = {k: np.zeros_like(v) for k, v in ref.items()}
  opt = R.AdamState(P, cfg)
  for t, key in enumerate(['a', 'b', 'a', 'b'], start=1):     # alternating applies share t
    g = RNG.randn(*ref[key].shape)
    opt.apply(P, {key: t64(g)})
    ref[key], m[key], v_[key] = N.adam_step(ref[key], g, m[key], v_[key], t)
  for k in P:
    np.testing.assert_allclose(P[k].numpy(), ref[k], rtol=1e-12)


# ---------------------------------------------------------------- network-level checks
def _np_ge_conv(P, scope, x, d, k_pad='SAME', act=True, pn=True):
  y = N.conv2d(x, P[scope + '/weights'].numpy(), k_pad)
  y = N.instance_norm(y, P[scope + '/InstanceNorm/gamma_' + d].numpy(), P[scope + '/InstanceNorm/beta_' + d].numpy())
  if act:
    y = N.leaky_relu(y)
  if pn:
    y = N.pixel_norm(y)
  return y


def test_encoder_generator_np_vs_torch_small():
  cfg = R.Config(hw=8, max_ch=8)
  P = R.init_params(cfg, seed=3, dtype=torch.float64, std='he')
  x = RNG.rand(2, 8, 8, 3)
  net, ep = R.encoder(P, t64(x), 's', cfg)
  # numpy: from_rgb 1x1 -> block(2 convs) -> pool
  h = _np_ge_conv(P, 'encoder_content/from_rgb_8x8/Conv', x, 's')
  h = _np_ge_conv(P, 'encoder_content/encoder_block_8x8x8/Conv', h, 's')
  skip = _np_ge_conv(P, 'encoder_content/encoder_block_8x8x8/Conv_1', h, 's')
  np.testing.assert_allclose(ep['encoder_block_8x8x8'].numpy(), skip, rtol=1e-9, atol=1e-10)
  code = N.avg_pool2(skip)
  np.testing.assert_allclose(net.numpy(), code, rtol=1e-9, atol=1e-10)
  out, _ = R.generator(P, net, 't', cfg, ep)
  g = _np_ge_conv(P, 'generator/block_4x4x8/Conv', code, 't')
  g = _np_ge_conv(P, 'generator/block_4x4x8/Conv_1', g, 't')
  g = np.concatenate([N.upsample2x(g), skip], axis=3)       # generator features first (pggan_utils.py:298)
  g = _np_ge_conv(P, 'generator/block_8x8x8/Conv', g, 't')
  g = _np_ge_conv(P, 'generator/block_8x8x8/Conv_1', g, 't')
  rgb = _np_ge_conv(P, 'generator/generator_to_rgb_8x8/Conv', g, 't', act=False, pn=False)   # norm still applies
  np.testing.assert_allclose(out.numpy(), rgb, rtol=1e-8, atol=1e-9)


def test_discriminator_np_vs_torch_small():
  cfg = R.Config(hw=8, max_ch=8)
  P = R.init_params(cfg, seed=4, dtype=torch.float64, std='he')
  x = RNG.rand(3, 8, 8, 3)
  pred, _ = R.discriminator(P, t64(x), cfg, 'discriminator_t')
  p = lambda s: P['discriminator_t/' + s].numpy()
  dc = lambda s, h, pad='SAME': N.leaky_relu(N.conv2d(h, p(s + '/weights'), pad) + p(s + '/biases'))
  h = dc('from_rgb_8x8/Conv', x)
  h = dc('encoder_block_8x8x8/Conv', h)
  h = dc('encoder_block_8x8x8/Conv_1', h)
  h = N.minibatch_state_concat(N.avg_pool2(h))
  h = dc('before_fc_1x1x8/Conv', h)
  h = dc('before_fc_1x1x8/Conv_1', h, 'VALID')
  ref = N.fully_connected(h.reshape(3, -1), p('prediction/fully_connected/weights'),
                          p('prediction/fully_connected/biases'))
  np.testing.assert_allclose(pred.numpy(), ref, rtol=1e-9, atol=1e-10)


def test_growing_alpha_endpoints():
  cfg1 = R.Config(hw=16, max_ch=8, is_growing=True, alpha_grow=1.0, use_unet=False)
  P = R.init_params(cfg1, seed=5, dtype=torch.float64, std='he')
  x = t64(RNG.rand(2, 16, 16, 3))
  cfg_ng = R.Config(hw=16, max_ch=8, use_unet=False)
  # alpha = 1: growing network == non-growing network on the same weights
  net1, ep1 = R.encoder(P, x, 's', cfg1)
  net0, ep0 = R.encoder(P, x, 's', cfg_ng)
  np.testing.assert_allclose(net1.numpy(), net0.numpy(), rtol=1e-12)
  # (with UNet on, the growing generator legitimately differs: at hw 8 it picks the encoder's
  #  *interpolated* end-point, nets/pggan_utils.py:291-297 -- so compare without skips)
  o1, _ = R.generator(P, net1, 't', cfg1, None)
  o0, _ = R.generator(P, net0, 't', cfg_ng, None)
  np.testing.assert_allclose(o1.numpy(), o0.numpy(), rtol=1e-12)
  p1, _ = R.discriminator(P, x, cfg1, 'discriminator_s')
  p0, _ = R.discriminator(P, x, cfg_ng, 'discriminator_s')
  np.testing.assert_allclose(p1.numpy(), p0.numpy(), rtol=1e-12)
  # real-image fade-in (image_generation.py:1001-1006)
  np.testing.assert_allclose(R.growing_image(x, 1.0).numpy(), x.numpy())
  np.testing.assert_allclose(R.growing_image(x, 0.0).numpy(), N.upsample2x(N.avg_pool2(x.numpy())), rtol=1e-13)


def test_unet_interpolated_endpoint_rule():
  # nets/pggan_utils.py:291-297: interpolated end-point used only when channel counts coincide (hw 8/16 stages)
  cfg = R.Config(hw=16, max_ch=8, is_growing=True, alpha_grow=0.3)
  P = R.init_params(cfg, seed=6, dtype=torch.float64, std='he')
  x = t64(RNG.rand(1, 16, 16, 3))
  net, ep = R.encoder(P, x, 's', cfg)
  assert 'encoder_block_interpolated_8x8x8' in ep
  layer = torch.zeros(1, 8, 8, 8, dtype=torch.float64)
  cat = R._concat_unet(layer, ep, cfg.max_ch)
  np.testing.assert_allclose(cat[..., 8:].numpy(), ep['encoder_block_interpolated_8x8x8'].numpy())


def test_losses_structure_and_gp_double_backward():
  cfg = R.Config(hw=8, max_ch=8)
  P = R.init_params(cfg, seed=7, dtype=torch.float64, std='he')
  s, t = t64(RNG.rand(3, 8, 8, 3)), t64(RNG.rand(3, 8, 8, 3))
  a = t64(RNG.rand(3, 1, 1, 1))
  for v in P.values():
    v.requires_grad_(True)
  gl, gt = R.generator_loss(P, s, t, cfg)
  assert set(gt) == {'l_cyc_s', 'l_cyc_t', 'generator_fool_loss_prime_s', 'generator_fool_loss_prime_t',
                     'l_content_s', 'l_content_t'}       # no cycle-GAN term below 64x64 (twingan.py:466)
  dl, dt = R.discriminator_loss(P, s, t, cfg, a, a)
  assert set(dt) == {'discriminator_loss_prime_s', 'discriminator_loss_prime_t',
                     'discriminator_gradient_penalty_prime_s', 'discriminator_gradient_penalty_prime_t'}
  d_names = R.discriminator_var_names(P)
  gd = R.grads_of(dl, P, d_names)
  gg = R.grads_of(gl, P, R.generator_var_names(P))
  assert any(float(v.abs().sum()) > 0 for v in gg.values())
  # finite-difference check of the GP double backward on one weight entry
  k = 'discriminator_s/encoder_block_8x8x8/Conv/weights'
  idx = (1, 1, 2, 3)
  eps = 1e-6
  with torch.no_grad():
    P[k][idx] += eps
  lp, _ = R.discriminator_loss(P, s, t, cfg, a, a)
  with torch.no_grad():
    P[k][idx] -= 2 * eps
  lm, _ = R.discriminator_loss(P, s, t, cfg, a, a)
  with torch.no_grad():
    P[k][idx] += eps
  fd = (lp.item() - lm.item()) / (2 * eps)
  assert abs(fd - gd[k][idx].item()) < 1e-5 * max(1.0, abs(fd))


def test_cycle_gan_term_from_64():
  cfg = R.Config(hw=64, max_ch=4)
  P = R.init_params(cfg, seed=8, dtype=torch.float32, std='he')
  s, t = torch.rand(1, 64, 64, 3), torch.rand(1, 64, 64, 3)
  a = torch.rand(1, 1, 1, 1)
  with torch.no_grad():
    _, gt = R.generator_loss(P, s, t, cfg)
  assert 'generator_fool_loss_cycle_s' in gt and 'generator_fool_loss_cycle_t' in gt
  for v in P.values():
    v.requires_grad_(True)
  _, dt = R.discriminator_loss(P, s, t, cfg, a, a)
  assert 'discriminator_loss_cycle_s' in dt and 'discriminator_gradient_penalty_prime_t' in dt


def test_train_step_alternates():
  cfg = R.Config(hw=8, max_ch=4)
  P = R.init_params(cfg, seed=9, dtype=torch.float64, std='he')
  P0 = {k: v.clone() for k, v in P.items()}
  opt = R.AdamState(P, cfg)
  s, t = t64(RNG.rand(2, 8, 8, 3)), t64(RNG.rand(2, 8, 8, 3))
  a = t64(RNG.rand(2, 1, 1, 1))
  out = R.train_step(P, opt, s, t, cfg, a, a, counter=0)      # counter % n_critic == 0 -> G apply
  assert 'g_loss' in out and 'd_loss' not in out
  gk, dk = R.generator_var_names(P), R.discriminator_var_names(P)
  assert all(torch.equal(P[k], P0[k]) for k in dk)
  assert any(not torch.equal(P[k], P0[k]) for k in gk)
  P1 = {k: v.clone() for k, v in P.items()}
  out = R.train_step(P, opt, s, t, cfg, a, a, counter=1)      # D apply
  assert 'd_loss' in out
  assert all(torch.equal(P[k], P1[k]) for k in gk)
  assert any(not torch.equal(P[k], P1[k]) for k in dk)
  assert opt.t == 2                                           # shared beta-power counter


def test_gan_loss_variants_np_vs_torch():
  """image_generation.py:331-400,441-449: sigmoid cross entropy (gan / dragan), hinge, drift, perturbed batch."""
  import torch.nn.functional as F
  rng = np.random.RandomState(40)
  pf, pr = rng.randn(6, 1) * 2, rng.randn(6, 1) * 2
  t = lambda a: torch.from_numpy(a)
  assert abs(N.sigmoid_cross_entropy(np.ones_like(pf), pf, 0.7) -
             float(F.binary_cross_entropy_with_logits(t(pf), torch.ones(6, 1, dtype=torch.float64)) * 0.7)) < 1e-12
  assert abs(N.sigmoid_cross_entropy(np.zeros_like(pf), pf) -
             float(F.binary_cross_entropy_with_logits(t(pf), torch.zeros(6, 1, dtype=torch.float64)))) < 1e-12
  assert abs(N.hinge_d_loss(pf, pr) - float(F.relu(1 + t(pf)).mean() + F.relu(1 - t(pr)).mean())) < 1e-12
  assert abs(N.drift_loss(pr, 0.001) - 0.001 * float((t(pr) ** 2).mean())) < 1e-15
  # known answers: xent at logit 0 is log 2 either way; hinge of a perfect critic is 0
  assert abs(N.sigmoid_cross_entropy(np.ones(3), np.zeros(3)) - np.log(2)) < 1e-15
  assert N.hinge_d_loss(np.full(4, -2.0), np.full(4, 2.0)) == 0.0
  x, noise = rng.rand(2, 4, 4, 3), rng.rand(2, 4, 4, 3) * 2 - 1
  assert np.allclose(N.dragan_perturbed_batch(x, noise), x + 0.5 * x.var() * noise)


@pytest.mark.parametrize('loss', ['hinge', 'gan', 'dragan'])
def test_loss_architectures_run_and_differentiate(loss):
  cfg = R.Config(hw=16, max_ch=8, loss=loss, drift=0.0)
  P = R.init_params(cfg, seed=1, dtype=torch.float64, std='he')
  g = torch.Generator().manual_seed(3)
  s, t_ = torch.rand(2, 16, 16, 3, generator=g).double(), torch.rand(2, 16, 16, 3, generator=g).double()
  a = torch.rand(2, 1, 1, 1, generator=g).double()
  noise = torch.rand(2, 16, 16, 3, generator=g).double() * 2 - 1
  for v in P.values():
    v.requires_grad_(True)
  gl, gt = R.generator_loss(P, s, t_, cfg)
  dl, dt = R.discriminator_loss(P, s, t_, cfg, a, a, noise, noise)
  if loss == 'hinge':
    assert set(dt) == {'discriminator_loss_prime_s', 'discriminator_loss_prime_t'}
  else:
    assert 'discriminator_fake_loss_prime_s' in dt and 'discriminator_real_loss_prime_t' in dt
    assert ('discriminator_gradient_penalty_prime_s' in dt) == (loss == 'dragan')
  grads = R.grads_of(dl, P, R.discriminator_var_names(P))
  assert all(torch.isfinite(v).all() for v in grads.values())


def test_batch_norm_np_vs_torch_and_moving_average():
  """libs/batch_norm.py:430,464-470 (training-mode moments over N,H,W; eps 1e-3) and :283-300 (moving average)."""
  rng = np.random.RandomState(41)
  x = rng.randn(3, 4, 4, 8) * 2 + 1
  ga, be = 1 + 0.1 * rng.randn(8), 0.1 * rng.randn(8)
  y, m, v = R.batch_norm_train(torch.from_numpy(x), torch.from_numpy(ga), torch.from_numpy(be))
  assert np.allclose(y.numpy(), N.batch_norm_train(x, ga, be), atol=1e-12)
  assert np.allclose(m.numpy(), x.mean(axis=(0, 1, 2))) and np.allclose(v.numpy(), x.var(axis=(0, 1, 2)))
  mm = R.moving_average_update(torch.zeros(8, dtype=torch.float64), m)
  assert np.allclose(mm.numpy(), 0.001 * x.mean(axis=(0, 1, 2)))
  # a per-channel constant input normalises to beta
  c = np.tile(np.arange(8.0), (2, 3, 3, 1))
  yc, _, _ = R.batch_norm_train(torch.from_numpy(c), torch.from_numpy(ga), torch.from_numpy(be))
  assert np.allclose(yc.numpy(), np.broadcast_to(be, c.shape), atol=1e-9)


def test_tf_stand_in_primitives_agree_with_numpy_restatement():
  """The TensorFlow stand-in that executes the reference's source (oracle/tf_shim) and the float64 NumPy oracle
  (oracle/np_ops.py) restate the same TensorFlow ops independently (torch vs hand-written NumPy loops): they must
  agree to round-off -- conv2d (SAME / VALID, k = 1, 3, 4), avg_pool, nearest upsampling, moments + batch_normalization,
  l2_normalize, tf.losses reductions, Adam."""
  from oracle import np_ops as N
  from oracle.tf_shim import core, tfapi
  core.STATE.reset(0)
  r = np.random.RandomState(11)
  T = lambda a: core.Tensor(torch.tensor(np.asarray(a, np.float64)))
  close = lambda a, b, tol=1e-12: np.abs(np.asarray(a) - np.asarray(b)).max() < tol
  x = r.randn(2, 8, 8, 5)
  for k, pad in ((1, 'SAME'), (3, 'SAME'), (4, 'VALID'), (4, 'SAME'), (3, 'VALID')):
    w = r.randn(k, k, 5, 7)
    if pad == 'SAME' and k == 4:
      continue      # np_ops restates only the paddings the hot path uses
    assert close(tfapi.nn_conv2d(T(x), T(w), [1, 1, 1, 1], pad).t.numpy(), N.conv2d(x, w, pad)), (k, pad)
  assert close(tfapi.nn_avg_pool(T(x), (1, 2, 2, 1), (1, 2, 2, 1), 'VALID').t.numpy(), N.avg_pool2(x))
  assert close(tfapi.image_resize_nearest(T(x), (16, 16)).t.numpy(), N.upsample2x(x))
  gamma, beta = 1 + 0.1 * r.randn(5), 0.1 * r.randn(5)
  mean, var = tfapi.nn_moments(T(x), [1, 2], keep_dims=True)
  inorm = tfapi.nn_batch_normalization(T(x), mean, var, T(beta), T(gamma), 1e-6)
  assert close(inorm.t.numpy(), N.instance_norm(x, gamma, beta))
  mean, var = tfapi.nn_moments(T(x), [0, 1, 2])
  bnorm = tfapi.nn_batch_normalization(T(x), mean, var, T(beta), T(gamma), 1e-3)
  ref_bn = N.batch_norm_train(x, gamma, beta)
  assert close(bnorm.t.numpy(), ref_bn[0] if isinstance(ref_bn, tuple) else ref_bn)
  a, b = r.rand(2, 4, 4, 3), r.rand(2, 4, 4, 3)
  assert abs(float(tfapi.absolute_difference(T(a), T(b), weights=0.7, loss_collection=None).t) -
             N.absolute_difference(a, b, 0.7)) < 1e-14
  logits, labels = r.randn(6, 1), np.ones((6, 1))
  assert abs(float(tfapi.sigmoid_cross_entropy(T(labels), T(logits), weights=0.3, loss_collection=None).t) -
             N.sigmoid_cross_entropy(labels, logits, 0.3)) < 1e-14
  # Adam: three applies of the stand-in optimizer == three np_ops.adam_step with t = 1, 2, 3
  theta0 = r.randn(9)
  v = core.get_variable('adam_probe', [9], core.float32, core.constant_initializer(theta0))
  opt = tfapi.AdamOptimizer(1e-3, beta1=0.5, beta2=0.99, epsilon=1e-8)
  th, m, vv = theta0.copy(), np.zeros(9), np.zeros(9)
  for t in range(1, 4):
    g = r.randn(9)
    opt.apply_gradients([(T(g), v)])
    th, m, vv = N.adam_step(th, g, m, vv, t, lr=1e-3, beta1=0.5, beta2=0.99, eps=1e-8)
    assert close(v.t.detach().numpy(), th, 1e-14), t
  core.STATE.reset(0)


def test_gemm_forms_of_the_conv_oracle_equal_the_einsum_forms():
  """np_ops.conv2d_gemm / conv2d_bwd_data_gemm / conv2d_bwd_weight_gemm (one BLAS product per tap; what the full-size
  layer checks of tests/test_gpu_bench_shapes.py use) are the same sums as the einsum restatements above them."""
  rng = np.random.RandomState(0)
  for (n, h, w, ci, co, k, pad) in [(2, 6, 5, 3, 4, 3, 'SAME'), (3, 4, 4, 5, 6, 4, 'VALID'), (2, 7, 7, 3, 5, 4, 'SAME'),
                                    (2, 5, 5, 4, 6, 1, 'SAME'), (2, 7, 6, 3, 5, 4, 'VALID')]:
    x, wt = rng.randn(n, h, w, ci), rng.randn(k, k, ci, co)
    y = N.conv2d(x, wt, pad)
    assert np.allclose(y, N.conv2d_gemm(x, wt, pad), atol=1e-12)
    gy = rng.randn(*y.shape)
    assert np.allclose(N.conv2d_bwd_data(gy, wt, (h, w), pad), N.conv2d_bwd_data_gemm(gy, wt, (h, w), pad), atol=1e-12)
    gw = N.conv2d_bwd_weight(x, gy, (k, k), pad)
    assert np.allclose(gw, N.conv2d_bwd_weight_gemm(x, gy, (k, k), pad), atol=1e-12)
    assert np.allclose(gw, N.conv2d_bwd_weight_gemm(x, gy, (k, k), pad, per_image=True).sum(0), atol=1e-12)


def test_attention_backward_closed_forms_match_autograd():
  """oracle/N.attention_backward / attention_backward_backward -- the formulas csrc/flash.hip evaluates tile by
  tile -- against float64 autograd of matmul -> softmax -> matmul (libs/self_attention.py:56-63), the way the reference
  obtains them (tf.gradients, twice under the gradient penalty: image_generation.py:414-439)."""
  rng = np.random.RandomState(2)
  n, ln, dk, dv = 2, 24, 4, 6
  q, k, v, go = rng.randn(n, ln, dk), rng.randn(n, ln, dk), rng.randn(n, ln, dv), rng.randn(n, ln, dv)
  aq, ak, av = rng.randn(n, ln, dk), rng.randn(n, ln, dk), rng.randn(n, ln, dv)
  tq, tk, tv, tg = (torch.tensor(x, dtype=torch.float64, requires_grad=True) for x in (q, k, v, go))
  o = torch.softmax(tq @ tk.transpose(1, 2), -1) @ tv
  assert np.abs(N.attention_forward(q, k, v)[0] - o.detach().numpy()).max() < 1e-12
  g1 = torch.autograd.grad(o, (tq, tk, tv), tg, create_graph=True)
  for got, want in zip(N.attention_backward(q, k, v, go), g1):
    assert np.abs(got - want.detach().numpy()).max() < 1e-12
  loss = sum((g * torch.tensor(a)).sum() for g, a in zip(g1, (aq, ak, av)))
  g2 = torch.autograd.grad(loss, (tq, tk, tv, tg))
  for got, want in zip(N.attention_backward_backward(q, k, v, go, aq, ak, av), g2):
    assert np.abs(got - want.numpy()).max() < 1e-11


def test_storage_rounding_sensitivity_of_the_headline_graph():
  """oracle/rounding.py: the float64 oracle with bf16 / fp16 rounding inserted at the 16-bit path's storage points.  The
  numbers DESIGN.md section 2 quotes for the bf16 tolerance (32 x 32, 32 channels, the inputs of
  tests/test_gpu_model.py::test_losses_and_gradients): the generator gradients move by ~0.3 under bf16 storage (most of it
  from rounding the weights and forward tensors), ~0.1 under fp16, ~1e-2 when only the back-propagated tensors are
  rounded -- the graph is chaotic in its forward activations, not in its backward.  The GPU test holds the kernels to
  these figures."""
  from oracle import rounding
  from oracle import torch_ref as R
  cfg = R.Config(hw=32, max_ch=32)
  P = {k: v.float().double() for k, v in R.init_params(cfg, seed=2, dtype=torch.float64, std='he').items()}
  g = torch.Generator().manual_seed(1234 + 2)
  s = torch.rand(2, 32, 32, 3, generator=g).to(torch.bfloat16).double()
  t = torch.rand(2, 32, 32, 3, generator=g).to(torch.bfloat16).double()
  e_bf16, _, _ = rounding.generator_gradient_sensitivity(P, s, t, cfg, torch.bfloat16)
  e_fp16, _, _ = rounding.generator_gradient_sensitivity(P, s, t, cfg, torch.float16)
  e_bwd, _, _ = rounding.generator_gradient_sensitivity(P, s, t, cfg, torch.bfloat16, forward=False, weights=False)
  print('[sensitivity] bf16 %.3f fp16 %.3f bf16 backward-only %.4f' % (e_bf16, e_fp16, e_bwd))
  assert 0.1 < e_bf16 < 0.6 and 0.02 < e_fp16 < 0.25 and e_fp16 < e_bf16 and e_bwd < 0.05
  # the patch is undone: the plain oracle is exact again
  la, _ = R.generator_loss(P, s, t, cfg)
  lb, _ = R.generator_loss(P, s, t, cfg)
  assert float(la) == float(lb) and R.conv2d.__module__ == 'oracle.torch_ref'


def test_storage_rounding_covers_spectral_norm_and_attention():
  """oracle/rounding.py on BASELINE configs[4]'s ingredients (spectral norm on the discriminator kernels, self-attention in
  E / G / D, WGAN-GP): the probe runs both loss groups with the pre-run u put back before each evaluation, fp16 rounding
  moves the gradients less than bf16, and with no rounding at all (every switch off) the patched graph IS the oracle."""
  from oracle import rounding
  from oracle import torch_ref as R
  cfg = R.Config(hw=16, max_ch=16, spectral_norm=True, do_self_attention=True, self_attention_hw=8)
  P = {k: v.float().double() for k, v in R.init_params(cfg, seed=5, dtype=torch.float64, std='he').items()}
  for k in P:
    if k.endswith('/sa_gamma'):
      P[k] = torch.full_like(P[k], 0.3)      # the gate starts at 0 (libs/self_attention.py:68): open it so attention matters
  sn0 = R.init_sn_state(P, seed=3)
  g = torch.Generator().manual_seed(77)
  s = torch.rand(2, 16, 16, 3, generator=g).to(torch.float16).double()
  t = torch.rand(2, 16, 16, 3, generator=g).to(torch.float16).double()
  a = torch.rand(2, 1, 1, 1, generator=g).double()

  def reset():
    cfg.sn_cache = None
    cfg.sn_state = {k: v.clone() for k, v in sn0.items()}
  losses = {'g': lambda Q: R.generator_loss(Q, s, t, cfg)[0], 'd': lambda Q: R.discriminator_loss(Q, s, t, cfg, a, a)[0]}
  names = {'g': R.generator_var_names(P), 'd': R.discriminator_var_names(P)}
  for grp in ('g', 'd'):
    e16, _, exact = rounding.gradient_sensitivity(P, names[grp], losses[grp], torch.float16, reset=reset)
    eb16, _, _ = rounding.gradient_sensitivity(P, names[grp], losses[grp], torch.bfloat16, reset=reset)
    e0, _, _ = rounding.gradient_sensitivity(P, names[grp], losses[grp], torch.float16, reset=reset, forward=False, backward=False,
                                             weights=False)
    print('[sensitivity sn+attention] %s: fp16 %.4f bf16 %.4f off %.1e' % (grp, e16, eb16, e0))
    assert 0.0 < e16 < eb16 < 1.0 and e0 < 1e-12
    assert any(float(exact[k].abs().max()) > 0 for k in names[grp] if '/self_attention_' in k)
  assert R.self_attention.__module__ == 'oracle.torch_ref'
