"""Every conv kernel variant one bench step dispatches, AT the bench's shapes (256x256 final stage, batch 16 per GPU
as the trainer batches it: n = 16 / 32 / 48 / 64, two-segment filter gradients 48+16 / 32+32), through the C ABI,
against the float64 oracle on the same bf16-rounded inputs -- plus the normalisation kernels at their bench shapes.

tests/golden/bench_dispatch_shapes.json (tools/make_dispatch_shapes.py, from the per-shape table of bench.py's
roofline pass) lists the (entry point, layer shape, n) triples; tests/golden/bench_dispatch_kernels.json holds the
kernel symbol each one selected (tg_last_kernel) when the table was recorded, so a silent change of dispatch is a
failure too (TG_RECORD_KERNELS=<path> re-records).

Tolerances: bf16 outputs (forward, backward-data) rel-L2 <= 4e-3 vs the oracle (one bf16 rounding of the output is
1.1e-3) on the first and last image of the batch, and <= 2e-3 vs the direct (one-thread-per-output) kernel over the
WHOLE tensor; fp32 outputs (filter and bias gradients, fp32 accumulation of exact bf16 products) rel-L2 <= 1e-4 vs the
oracle over the whole batch.  SURVEY.md 8c asks 1e-2 / 3e-2.
"""
import json
import os
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import np_ops as N          # noqa: E402  (checker only)

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, 'golden', 'bench_dispatch_shapes.json')) as _fh:
  LAYERS = json.load(_fh)['layers']
KERNELS_PATH = os.path.join(HERE, 'golden', 'bench_dispatch_kernels.json')
BF16_OUT_TOL, VS_DIRECT_TOL, F32_OUT_TOL = 4e-3, 2e-3, 1e-4
NMAX = 64
RECORDED = {}


def rel_l2(a, b):
  a = np.asarray(a, np.float64)
  b = np.asarray(b, np.float64)
  assert a.shape == b.shape, (a.shape, b.shape)
  return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def host(t):
  return t.detach().float().cpu().numpy().astype(np.float64)


def last_kernel():
  from twingan_amd import _lib
  return _lib.load().tg_last_kernel().decode()


def _note(key, ep, n, sym=None):
  RECORDED.setdefault(key, {}).setdefault(ep, {})[n] = sym if sym is not None else last_kernel()


def _rand_bf16(shape, seed, dtype=torch.bfloat16):
  g = torch.Generator(device='cuda').manual_seed(seed)
  return torch.randn(shape, generator=g, device='cuda', dtype=torch.float32).to(dtype).contiguous()


def _check_partials(st, y, key):
  """sum over chunks of the conv's statistics partials == (sum y, sum y^2) per image and channel of the tensor it wrote."""
  n, h, w, c = y.shape
  part = st.part.view(n, st.chunks, 2, c).double().sum(dim=1)
  yd = y.double().view(n, h * w, c)
  s1, s2 = yd.sum(dim=1), (yd * yd).sum(dim=1)
  e1 = float(((part[:, 0] - s1).abs() / (s2 * h * w).sqrt().clamp_min(1e-30)).max())
  e2 = float((part[:, 1] / s2.clamp_min(1e-30) - 1).abs().max())
  assert e1 < 2e-6 and e2 < 2e-6, ('statistics partials', key, e1, e2)


class _Direct:
  """Forces the direct (one thread per output) algorithm inside the block."""

  def __enter__(self):
    import twingan_amd.ops as O
    self.O, self.saved = O, O._mfma_ok
    O._mfma_ok = lambda *a: False

  def __exit__(self, *a):
    self.O._mfma_ok = self.saved


@pytest.mark.parametrize('key', sorted(k for k in LAYERS if '+' not in k.split('>')[0]))
def test_conv_variants_at_bench_shapes(key):
  _conv_variants(key, torch.bfloat16)


@pytest.mark.parametrize('key', sorted(k for k in LAYERS if '+' not in k.split('>')[0]))
def test_fp16_conv_variants_at_config4_shapes(key):
  """BASELINE configs[4] runs the same layer table in half precision (--dataset_dtype float16,
  deployment/model_deploy.py:146-183): every (entry point, layer shape, n) of the table again with TG_F16 operands
  through the f16 MFMA instantiations.  Forward launches that carry a statistics / pool epilogue in bf16 are plain
  forwards here when the epilogue has no f16 instantiation (the host then takes the unfused route)."""
  _conv_variants(key, torch.float16)


def _conv_variants(key, dtype):
  import twingan_amd.ops as O
  from twingan_amd._lib import TG_EPI_BIAS, TG_EPI_LRELU
  f16 = dtype == torch.float16
  # one rounding of the output to the storage type: 2^-9 relative for bf16, 2^-12 for fp16
  BF16_OUT_TOL, VS_DIRECT_TOL = (6e-4, 4e-4) if f16 else (4e-3, 2e-3)
  note = (lambda *a, **k: None) if f16 else _note      # the recorded symbol table is config 3's (bf16)
  k, cin, cout, hw = (int(v) for v in re.match(r'k(\d):c(\d+)>(\d+):hw(\d+)', key).groups())
  valid = k == 4
  hin = 4 if valid else hw
  spec = O.ConvSpec(k, 'VALID' if valid else 'SAME')
  eps = dict(LAYERS[key])
  if f16:
    probe = _rand_bf16((1, hin, hin, cin), 1, dtype)
    wprobe = torch.zeros(k, k, cin, cout, device='cuda')
    plain = set(eps.get('tg_conv2d_fwd', []))
    if 'tg_conv2d_fwd_stats' in eps and O.conv_fwd_stats_raw(probe, wprobe, spec)[1] is None:
      plain |= set(eps.pop('tg_conv2d_fwd_stats'))
    if 'tg_conv2d_fwd_pool' in eps:
      import ctypes
      from twingan_amd import _lib
      d = O._desc(probe.shape, cout, spec, dtype, TG_EPI_BIAS | TG_EPI_LRELU)
      if not (d.algo == _lib.TG_ALGO_MFMA and _lib.load().tg_conv2d_fwd_pool_supported(ctypes.byref(d))):
        plain |= set(eps.pop('tg_conv2d_fwd_pool'))
    if plain:
      eps['tg_conv2d_fwd'] = sorted(plain, key=int)
  x = _rand_bf16((NMAX, hin, hin, cin), 1, dtype)
  gy = _rand_bf16((NMAX, hw, hw, cout), 2, dtype)
  g = torch.Generator().manual_seed(3)
  w = (torch.randn(k, k, cin, cout, generator=g) / (k * k * cin) ** 0.5).to(dtype).float().cuda().contiguous()
  bias = (torch.randn(cout, generator=g) * 0.1).cuda()
  wn, bn = host(w), host(bias)
  pad = 'VALID' if valid else 'SAME'

  # ---- forward: both epilogues (generator / encoder: none; discriminator: bias + LeakyReLU)
  for n in (int(v) for v in eps.get('tg_conv2d_fwd', [])):
    sel = sorted({0, n - 1})
    xs = host(x[sel])
    lin = N.conv2d_gemm(xs, wn, pad)
    for epi, ref in ((0, lin), (TG_EPI_BIAS | TG_EPI_LRELU, N.leaky_relu(lin + bn))):
      y = O.conv_fwd_raw(x[:n], w, bias if epi else None, spec, epi)
      note(key, 'tg_conv2d_fwd', str(n))
      e = rel_l2(host(y[sel]), ref)
      assert e < BF16_OUT_TOL, ('fwd', key, n, epi, e)
      with _Direct():
        yd = O.conv_fwd_raw(x[:n], w, bias if epi else None, spec, epi)
      if epi == 0:      # with LeakyReLU a 1-ulp difference across zero is a 5x jump of the element: linear part only
        e = rel_l2(host(y), host(yd))
        assert e < VS_DIRECT_TOL, ('fwd vs direct', key, n, e)
      del y, yd

  # ---- forward with the statistics epilogue (every normalised encoder / generator conv): the same tensor as the plain
  # forward, bit for bit, and partial sums that add up to the sums of that tensor
  for n in (int(v) for v in eps.get('tg_conv2d_fwd_stats', [])):
    y, st = O.conv_fwd_stats_raw(x[:n], w, spec)
    note(key, 'tg_conv2d_fwd_stats', str(n))
    assert st is not None, ('no statistics epilogue', key, n)
    sel = sorted({0, n - 1})
    e = rel_l2(host(y[sel]), N.conv2d_gemm(host(x[sel]), wn, pad))
    assert e < BF16_OUT_TOL, ('fwd_stats', key, n, e)
    with _Direct():
      yd = O.conv_fwd_raw(x[:n], w, None, spec, 0)
    e = rel_l2(host(y), host(yd))
    assert e < VS_DIRECT_TOL, ('fwd_stats vs direct', key, n, e)
    _check_partials(st, y, key)
    del y, yd

  # ---- forward that also writes its 2x2 average pool (the last conv of a discriminator block): z bit-identical to the
  # plain forward's, the pooled tensor = the pool of that z
  for n in (int(v) for v in eps.get('tg_conv2d_fwd_pool', [])):
    epi = TG_EPI_BIAS | TG_EPI_LRELU
    z, zp = O.conv_fwd_pool_raw(x[:n], w, bias, spec, epi)
    note(key, 'tg_conv2d_fwd_pool', str(n))
    z_plain = O.conv_fwd_raw(x[:n], w, bias, spec, epi)
    assert torch.equal(z, z_plain), ('fwd_pool z', key, n)
    sel = sorted({0, n - 1})
    e = rel_l2(host(z[sel]), N.leaky_relu(N.conv2d_gemm(host(x[sel]), wn, pad) + bn))
    assert e < BF16_OUT_TOL, ('fwd_pool', key, n, e)
    want = z.float().view(n, hw // 2, 2, hw // 2, 2, cout).mean(dim=(2, 4))
    e = rel_l2(host(zp), host(want))
    assert e < 2e-3, ('fwd_pool pooled', key, n, e)      # one bf16 rounding of the pooled value
    # the sign-bit form of the same launch (tg_conv2d_fwd_pool_signs: what the discriminators' first-order passes run):
    # the very same pooled tensor, and bit j of byte q = (z[.., 8q+j] > 0) of the z the plain launch stored
    assert O.conv_fwd_pool_signs_supported(x[:n], w, spec, epi), ('no sign-bit variant', key, n)
    signs, zp2 = O.conv_fwd_pool_signs_raw(x[:n], w, bias, spec, epi)
    note(key, 'tg_conv2d_fwd_pool_signs', str(n))
    assert torch.equal(zp2, zp), ('fwd_pool_signs pooled', key, n)
    bits = (z > 0).view(n, hw, hw, cout // 8, 8).to(torch.int32)
    want_bytes = (bits << torch.arange(8, device='cuda', dtype=torch.int32)).sum(dim=-1).to(torch.uint8)
    assert torch.equal(signs, want_bytes), ('fwd_pool_signs bits', key, n, int((signs != want_bytes).sum()))
    # ... and the backward that consumes them: the same tensor as the one rebuilt from z
    gzp = _rand_bf16(tuple(zp.shape), 21, dtype)
    g_ref, gb_ref = O.lrelu_pool_bwd(None, gzp, z, 0.2, bias, True)
    g_sig, gb_sig = O.lrelu_pool_bwd_signs(gzp, signs, 0.2, bias, True)
    assert torch.equal(g_sig, g_ref), ('lrelu_pool_bwd_signs', key, n)
    assert rel_l2(host(gb_sig), host(gb_ref)) < 1e-5, ('lrelu_pool_bwd_signs bias', key, n)
    del z, zp, z_plain, signs, zp2, g_ref, g_sig

  # ---- backward-data, plain and with the producer's LeakyReLU mask in the epilogue
  for ep in ('tg_conv2d_bwd_data', 'tg_conv2d_bwd_data_masked'):
    for n in (int(v) for v in eps.get(ep, [])):
      sel = sorted({0, n - 1})
      ref = N.conv2d_bwd_data_gemm(host(gy[sel]), wn, (hin, hin), pad)
      if ep.endswith('masked'):
        gx = O.conv_bwd_data_masked_raw(gy[:n], w, x[:n], spec)
        ref = ref * np.where(host(x[sel]) > 0, 1.0, 0.2)
      else:
        gx = O.conv_bwd_data_raw(gy[:n], w, (n, hin, hin, cin), spec)
      note(key, ep, str(n))
      e = rel_l2(host(gx[sel]), ref)
      assert e < BF16_OUT_TOL, (ep, key, n, e)
      with _Direct():
        gd = O.conv_bwd_data_raw(gy[:n], w, (n, hin, hin, cin), spec)
        if ep.endswith('masked'):
          gd = O.lrelu_bwd_raw(gd, x[:n], 0.2)
      e = rel_l2(host(gx), host(gd))
      # the direct kernel rounds once more before the mask: 2 roundings apart
      assert e < (2 * VS_DIRECT_TOL if ep.endswith('masked') else VS_DIRECT_TOL), (ep + ' vs direct', key, n, e)
      del gx, gd

  # ---- filter gradients (fp32): single batch, with the bias gradient, two segments
  wg_eps = [e for e in eps if 'bwd_weight' in e]
  if wg_eps:
    per = N.conv2d_bwd_weight_gemm(host(x), host(gy), (k, k), pad, per_image=True)      # [64, k, k, cin, cout]
    bsum = host(gy).sum(axis=(1, 2))                                                       # [64, cout]
    for ep in wg_eps:
      for nn in eps[ep]:
        parts = [int(v) for v in nn.split('+')]
        n = sum(parts)
        want_b = ep.endswith('_bias')
        gb = torch.zeros(cout, dtype=torch.float32, device='cuda') if want_b else None
        if len(parts) == 1:
          gw = O.conv_bwd_weight_raw(x[:n], gy[:n], spec, gbias=gb)
          note(key, ep, nn)
          bias_ref = bsum[:n].sum(0)
        else:
          a = parts[0]
          gw = torch.zeros((k, k, cin, cout), dtype=torch.float32, device='cuda')
          # the trainer's bias segments: only the batched pass (segment a) feeds the bias in a D step
          ok = O.conv_bwd_weight2_raw(x[:a], gy[:a], x[a:n], gy[a:n], spec, gw, gb, 1 if want_b else 3)
          assert ok, ('two-segment filter gradient refused', key, nn)
          note(key, ep, nn)
          bias_ref = bsum[:a].sum(0)
        e = rel_l2(host(gw), per[:n].sum(0))
        assert e < F32_OUT_TOL, (ep, key, nn, e)
        if want_b:
          e = rel_l2(host(gb), bias_ref)
          assert e < F32_OUT_TOL, (ep + ' bias', key, nn, e)
        del gw


@pytest.mark.parametrize('key', sorted(k for k in LAYERS if '+' in k.split('>')[0]))
def test_upcat_conv_at_bench_shapes(key):
  """conv3x3(concat(up2(x0), skip)) read from its two sources (generator_three_layer_block's first conv), forward and
  filter gradient, with the trainer's skip-group permutation (four generator passes read two encoder passes)."""
  import twingan_amd.ops as O
  c0, c1, cout, hw = (int(v) for v in re.match(r'k3:c(\d+)\+(\d+)>(\d+):hw(\d+)', key).groups())
  n, gsz, perm = NMAX, 16, (1, 0, 0, 1)
  x0 = _rand_bf16((n, hw // 2, hw // 2, c0), 5)
  x1 = _rand_bf16((2 * gsz, hw, hw, c1), 6)
  gy = _rand_bf16((n, hw, hw, cout), 7)
  g = torch.Generator().manual_seed(8)
  w = (torch.randn(3, 3, c0 + c1, cout, generator=g) / (9 * (c0 + c1)) ** 0.5).to(torch.bfloat16).float().cuda()
  w = w.contiguous().requires_grad_(True)
  assert O.upcat_conv_supported(x0, x1, w)
  x0.requires_grad_(True)
  x1.requires_grad_(True)
  if 'tg_conv2d_upcat_fwd_stats' in LAYERS[key]:
    y, st = O.upcat_conv_stats(x0, x1, w, gsz, perm)
    _note(key, 'tg_conv2d_upcat_fwd_stats', str(n))
    assert st is not None, ('no statistics epilogue', key)
    _check_partials(st, y.detach(), key)
  else:
    y = O.upcat_conv(x0, x1, w, gsz, perm)
    _note(key, 'tg_conv2d_upcat_fwd', str(n))
  wn = host(w)

  def cat_of(i):      # the materialised input of image i
    up = np.repeat(np.repeat(host(x0[i:i + 1]), 2, axis=1), 2, axis=2)
    return np.concatenate([up, host(x1[perm[i // gsz] * gsz + i % gsz][None])], axis=3)
  for i in (0, gsz + 3, n - 1):
    e = rel_l2(host(y[i:i + 1]), N.conv2d_gemm(cat_of(i), wn))
    assert e < BF16_OUT_TOL, ('upcat fwd', key, i, e)
  y.backward(gy)
  _note(key, 'tg_conv2d_upcat_bwd_weight', str(n))      # the filter gradient is the last conv launch of the backward
  gyn = host(gy)
  ref = np.zeros_like(wn)
  for i in range(n):
    ref += N.conv2d_bwd_weight_gemm(cat_of(i), gyn[i:i + 1], (3, 3))
  e = rel_l2(host(w.grad), ref)
  assert e < F32_OUT_TOL, ('upcat wgrad', key, e)
  # input gradients: backward-data over the concat layout, split into the two sources (skip groups summed: every
  # encoder image is read by two generator passes)
  gcat0 = N.conv2d_bwd_data_gemm(gyn[:1], wn, (hw, hw))
  g0_ref = gcat0[..., :c0].reshape(1, hw // 2, 2, hw // 2, 2, c0).sum(axis=(2, 4))
  e = rel_l2(host(x0.grad[:1]), g0_ref)
  assert e < 2 * BF16_OUT_TOL, ('upcat gx0', key, e)
  j = 5                                            # skip image 5 (group 0) is read by output groups 1 and 2
  tot = sum(N.conv2d_bwd_data_gemm(gyn[i:i + 1], wn, (hw, hw))[..., c0:] for i in (gsz + j, 2 * gsz + j))
  e = rel_l2(host(x1.grad[j:j + 1]), tot)
  assert e < 2 * BF16_OUT_TOL, ('upcat gx1', key, e)
  # the backward-data kernel that wrote them (concat adjoint in its epilogue), called directly: same bits, and its symbol
  if O.USE_UPCAT_BWD_FUSED:
    d = O._desc((n, hw, hw, c0 + c1), cout, O.ConvSpec(3, 'SAME'), gy.dtype, 0)
    g0, g1 = torch.empty_like(x0), torch.empty_like(x1)
    wpack = O.PackCache.get(w.detach(), d, 1)      # held in a local: an uncached pack must outlive the launch
    O.call('tg_conv2d_upcat_bwd_data', gy.data_ptr(), wpack.data_ptr(), g0.data_ptr(), g1.data_ptr(),
           n, hw, hw, c0, c1, cout, gsz, O._pack_perm(perm), O._dt(gy), O._stream())
    _note(key, 'tg_conv2d_upcat_bwd_data', str(n))
    assert torch.equal(g0, x0.grad) and torch.equal(g1, x1.grad), ('upcat bwd_data direct call', key)
    # and against the unfused composition (backward-data into the concat layout, then the split): two roundings there
    gcat = O.conv_bwd_data_raw(gy, w.detach(), (n, hw, hw, c0 + c1), O.ConvSpec(3, 'SAME'))
    r0, r1 = torch.empty_like(x0), torch.empty_like(x1)
    O.call('tg_upsample2x_concat_bwd', gcat.data_ptr(), r0.data_ptr(), r1.data_ptr(), n, hw // 2, hw // 2, c0, c1, gsz,
           O._pack_perm(perm), O._dt(gcat), O._stream())
    e0, e1 = rel_l2(host(g0), host(r0)), rel_l2(host(g1), host(r1))
    assert e0 < BF16_OUT_TOL and e1 < BF16_OUT_TOL, ('fused vs composed concat backward', key, e0, e1)


@pytest.mark.parametrize('n', [32, 48, 64])
def test_flash_attention_at_config4_shapes(n):
  """The attention core BASELINE configs[4] dispatches (`bench.py --config 4`: self-attention at 64 x 64 = 4096 positions,
  64 channels -> d_qk 8, d_v 64, fp16; n = 32 encoder passes, 48 discriminator passes, 64 generator passes):
  tg_flash_attention_fwd / _bwd / _bwd_bwd over the whole batch, first and last image against the float64 closed forms of
  oracle/np_ops.py (libs/self_attention.py:56-63 and its tf.gradients / gradient-penalty derivatives) on the same
  fp16-rounded operands.  Tolerances: the fp16 bounds of tests/test_gpu_ops.py (forward 8e-4, backward 2e-3, second order
  4e-3) -- P / dS are rounded to fp16 before the second product, sums over 4096 keys."""
  import twingan_amd.ops as O
  ln, dk, dv, dt = 4096, 8, 64, torch.float16
  rng = np.random.RandomState(100 + n)

  def op(shape, scale=1.0, squash=False):
    a = rng.randn(*shape) * scale
    return torch.tensor(np.tanh(a) if squash else a, dtype=torch.float32).to(dt)
  q, k = op((n, ln, dk), 1.0, True), op((n, ln, dk), 2.0, True)
  v, go = op((n, ln, dv)), op((n, ln, dv))
  aq, ak, av = op((n, ln, dk)), op((n, ln, dk)), op((n, ln, dv))
  sel = [0, n - 1]
  h64 = lambda t: t[sel].double().numpy()
  qd, kd, vd, gd = (t.cuda().requires_grad_(True) for t in (q, k, v, go))
  assert O.flash_attention_supported(qd, vd) and O.flash_attention_trainable(qd, vd)
  # forward
  o, lse = O.flash_attention_fwd_raw(qd.detach(), kd.detach(), vd.detach())
  ref_o, _ = N.attention_forward(h64(q), h64(k), h64(v))
  e = rel_l2(host(o[sel]), ref_o)
  print('[flash c4 n%d] fwd %.2e' % (n, e))
  assert e < 8e-4, ('flash fwd', n, e)
  assert bool(torch.isfinite(o.float()).all()) and bool(torch.isfinite(lse).all())
  # first-order backward (no create_graph: tg_flash_attention_bwd)
  of = O.flash_attention(qd, kd, vd)
  gq, gk, gv = torch.autograd.grad(of, (qd, kd, vd), go.cuda())
  for got, ref, nm in zip((gq, gk, gv), N.attention_backward(h64(q), h64(k), h64(v), h64(go)), ('dq', 'dk', 'dv')):
    e = rel_l2(host(got[sel]), ref)
    print('[flash c4 n%d] bwd %s %.2e' % (n, nm, e))
    assert e < 2e-3, ('flash bwd ' + nm, n, e)
    assert bool(torch.isfinite(got.float()).all())
  # second order: the gradient-penalty pattern (create_graph backward, then the backward of that)
  with O.second_order():
    assert O.flash_attention_trainable(qd, vd) == O.USE_FLASH_BWD_BWD
  of = O.flash_attention(qd, kd, vd)
  gq, gk, gv = torch.autograd.grad(of, (qd, kd, vd), gd, create_graph=True)
  loss = (gq.float() * aq.cuda().float()).sum() + (gk.float() * ak.cuda().float()).sum() + (gv.float() * av.cuda().float()).sum()
  adj = torch.autograd.grad(loss, (qd, kd, vd, gd))
  ref = N.attention_backward_backward(h64(q), h64(k), h64(v), h64(go), h64(aq), h64(ak), h64(av))
  for got, want, nm in zip(adj, ref, ('adj q', 'adj k', 'adj v', 'adj dO')):
    e = rel_l2(host(got[sel]), want)
    print('[flash c4 n%d] bwd_bwd %s %.2e' % (n, nm, e))
    assert e < 4e-3, ('flash bwd_bwd ' + nm, n, e)
    assert bool(torch.isfinite(got.float()).all())


NORM_SHAPES = [
    # c, hw, n, pool    (encoder: [s; t] and [s'; t'] batches of 32 with the block-end avg-pool; generator: 4 passes = 64)
    (16, 256, 32, False), (32, 256, 32, True), (64, 128, 32, True), (128, 64, 32, True), (256, 32, 32, True),
    (256, 8, 32, True), (16, 256, 64, False), (32, 128, 64, False), (64, 64, 64, False), (128, 32, 64, False),
    (256, 16, 64, False), (256, 4, 64, False),
]


@pytest.mark.parametrize('c,hw,n,pool', NORM_SHAPES)
def test_norm_act_at_bench_shapes(c, hw, n, pool):
  """instance norm (two domains along N) + LeakyReLU + pixel norm (+ the encoder's avg-pool), forward and backward, at
  the bench's tensor shapes; the oracle (float64 autograd of the literal formulas) on the same bf16 inputs."""
  import twingan_amd.ops as O
  y = _rand_bf16((n, hw, hw, c), 11)
  y = (y.float() * 0.7 + 0.3).to(torch.bfloat16).contiguous().requires_grad_(True)
  g = torch.Generator().manual_seed(12)
  par = [(torch.randn(c, generator=g) * s + o).cuda().requires_grad_(True) for s, o in ((0.2, 1.0), (0.2, 0.0)) * 2]
  ga, be, ga2, be2 = par
  split = n // 2
  out = O.norm_act(y, ga, be, lrelu=True, pixel_norm=True, gamma2=ga2, beta2=be2, split=split, pool=pool)
  z, zp = out if pool else (out, None)
  gz = _rand_bf16(tuple(z.shape), 13)
  gzp = _rand_bf16(tuple(zp.shape), 14) if pool else None
  torch.autograd.backward([z, zp] if pool else [z], [gz, gzp] if pool else [gz])
  want = [np.zeros(c) for _ in range(4)]
  checked = (0, split - 1, split, n - 1)
  for i in range(n):
    yi = y[i:i + 1].detach().double().cpu().requires_grad_(True)
    p = [t.detach().double().cpu().requires_grad_(True) for t in ((ga, be) if i < split else (ga2, be2))]
    mean = yi.mean(dim=(1, 2), keepdim=True)
    var = ((yi - mean) ** 2).mean(dim=(1, 2), keepdim=True)
    u = (yi - mean) * torch.rsqrt(var + 1e-6) * p[0] + p[1]
    a = torch.maximum(0.2 * u, u)
    zr = a * torch.rsqrt((a * a).mean(dim=3, keepdim=True) + 1e-6)
    outs, gouts = [zr], [gz[i:i + 1].double().cpu()]
    if pool:
      outs.append(zr.reshape(1, hw // 2, 2, hw // 2, 2, c).mean(dim=(2, 4)))
      gouts.append(gzp[i:i + 1].double().cpu())
    torch.autograd.backward(outs, gouts)
    k = 0 if i < split else 2
    want[k] += p[0].grad.numpy()
    want[k + 1] += p[1].grad.numpy()
    if i in checked:
      assert rel_l2(host(z[i:i + 1]), zr.detach().numpy()) < 6e-3, ('z', i)
      if pool:
        assert rel_l2(host(zp[i:i + 1]), outs[1].detach().numpy()) < 6e-3, ('zp', i)
      e = rel_l2(host(y.grad[i:i + 1]), yi.grad.numpy())
      assert e < 1.5e-2, ('gy', i, e)
  for t, wv, nm in zip(par, want, ('gamma', 'beta', 'gamma2', 'beta2')):
    e = rel_l2(host(t.grad), wv)
    assert e < 2e-3, (nm, e)      # fp32 sums of bf16-exact inputs; the bf16 rounding of z does not enter


def test_dispatch_table_matches_recorded_kernels():
  """Runs last in this file: every dispatch above must have selected the kernel symbol recorded in
  tests/golden/bench_dispatch_kernels.json."""
  if not RECORDED:
    pytest.skip('the shape tests of this file did not run in this session')
  dst = os.environ.get('TG_RECORD_KERNELS')
  if dst:
    with open(dst, 'w') as fh:
      json.dump(RECORDED, fh, indent=1, sort_keys=True)
    return
  assert os.path.exists(KERNELS_PATH), 'record with TG_RECORD_KERNELS=tests/golden/bench_dispatch_kernels.json'
  with open(KERNELS_PATH) as fh:
    want = json.load(fh)
  for key, eps in RECORDED.items():
    for ep, by_n in eps.items():
      for n, sym in by_n.items():
        assert want.get(key, {}).get(ep, {}).get(n) == sym, (key, ep, n, sym, want.get(key, {}).get(ep, {}).get(n))
