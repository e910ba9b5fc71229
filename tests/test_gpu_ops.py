"""GPU parity: every HIP operator (through the C ABI) against the oracle on the same seeded inputs.

Tolerances: fp32 kernels vs the float64 oracle: rel-L2 <= 2e-5 (single primitive);
bf16 kernels vs the oracle evaluated on the same bf16-rounded inputs: rel-L2 <= 1e-2 forward,
3e-2 gradients; MFMA bf16 kernels vs the direct bf16 kernels (same rounding, different summation
order): rel-L2 <= 2e-3.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import np_ops as N          # noqa: E402  (checker only)
from oracle import torch_ref as R       # noqa: E402

F32_TOL = 2e-5
BF16_FWD_TOL = 1e-2
BF16_GRAD_TOL = 3e-2
MFMA_VS_DIRECT_TOL = 2e-3


def dev():
  return torch.device('cuda:0')


def rel_l2(a, b):
  a = np.asarray(a, np.float64)
  b = np.asarray(b, np.float64)
  assert a.shape == b.shape, (a.shape, b.shape)
  return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def to_dev(a, dtype=torch.float32):
  return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev()).to(dtype).contiguous()


def host(t):
  return t.detach().float().cpu().numpy().astype(np.float64)


def bf16_round(a):
  return torch.from_numpy(np.asarray(a, np.float32)).to(torch.bfloat16).float().numpy().astype(np.float64)


def tol_for(dtype, grad=False):
  if dtype == torch.float32:
    return F32_TOL
  return BF16_GRAD_TOL if grad else BF16_FWD_TOL


@pytest.fixture(scope='module')
def ops():
  from twingan_amd import ops as _ops
  return _ops


@pytest.fixture
def record_calls():
  """-> the list of (entry point, args) of every library call made while the test runs."""
  import twingan_amd.ops as O
  real, seen = O.call, []

  def spy(name, *a, **kw):
    seen.append((name, a))
    return real(name, *a, **kw)
  O.call = spy
  yield seen
  O.call = real


# ---------------------------------------------------------------------------------------------- conv
CONV_CASES = [
    # n, h, w, cin, cout, k, padding
    (2, 6, 6, 5, 7, 3, 'SAME'),
    (1, 9, 5, 3, 4, 3, 'SAME'),
    (2, 5, 5, 4, 6, 1, 'SAME'),
    (3, 4, 4, 5, 6, 4, 'VALID'),
    (2, 7, 7, 3, 5, 4, 'VALID'),
    (2, 8, 8, 16, 16, 3, 'SAME'),
]


@pytest.mark.parametrize('n,h,w,cin,cout,k,padding', CONV_CASES)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_conv_direct_vs_oracle(ops, n, h, w, cin, cout, k, padding, dtype):
  rng = np.random.RandomState(1)
  x = rng.randn(n, h, w, cin)
  wt = rng.randn(k, k, cin, cout) / np.sqrt(k * k * cin)
  b = rng.randn(cout) * 0.1
  if dtype == torch.bfloat16:
    x, wr = bf16_round(x), bf16_round(wt)
  else:
    wr = wt
  import twingan_amd.ops as O
  saved = O._mfma_ok
  O._mfma_ok = lambda *a: False          # force the direct algorithm
  try:
    xd = to_dev(x, dtype).requires_grad_(True)
    wd = to_dev(wt).requires_grad_(True)
    bd = to_dev(b).requires_grad_(True)
    y = ops.conv2d(xd, wd, bd, k, padding, lrelu=True)
    ref = N.leaky_relu(N.conv2d(x, wr, padding) + b)
    assert rel_l2(host(y), ref) < tol_for(dtype)
    gy = rng.randn(*ref.shape)
    if dtype == torch.bfloat16:
      gy = bf16_round(gy)
    y.backward(to_dev(gy, dtype))
    gpre = gy * np.where(ref > 0, 1.0, 0.2)
    if dtype == torch.bfloat16:
      gpre = bf16_round(gpre)
    assert rel_l2(host(xd.grad), N.conv2d_bwd_data(gpre, wr, (h, w), padding)) < tol_for(dtype, True)
    assert rel_l2(host(wd.grad), N.conv2d_bwd_weight(x, gpre, (k, k), padding)) < tol_for(dtype, True)
    assert rel_l2(host(bd.grad), gpre.sum(axis=(0, 1, 2))) < tol_for(dtype, True)
  finally:
    O._mfma_ok = saved


MFMA_CASES = [
    # n, h, w, cin, cout, k, padding    (model layer shapes at reduced extent + ragged tiles)
    (2, 16, 16, 16, 16, 3, 'SAME'),
    (2, 16, 16, 16, 32, 3, 'SAME'),
    (1, 32, 32, 32, 64, 3, 'SAME'),
    (2, 8, 8, 64, 128, 3, 'SAME'),
    (2, 8, 8, 128, 256, 3, 'SAME'),
    (3, 4, 4, 256, 256, 3, 'SAME'),
    (2, 8, 8, 512, 256, 3, 'SAME'),
    (1, 16, 16, 256, 64, 3, 'SAME'),
    (1, 32, 32, 64, 16, 3, 'SAME'),
    (2, 16, 16, 128, 128, 3, 'SAME'),
    (2, 16, 32, 256, 256, 3, 'SAME'),
    (5, 4, 4, 264, 256, 3, 'SAME'),      # D tail conv after minibatch-stddev padding
    (5, 4, 4, 64, 64, 4, 'VALID'),       # dense rewrite of the 4x4 VALID conv
    (16, 4, 4, 256, 256, 4, 'VALID'),
    (1, 20, 12, 24, 40, 3, 'SAME'),      # ragged spatial tiles, channels not multiples of 16/32
    (3, 10, 18, 8, 8, 3, 'SAME'),
    (2, 12, 12, 32, 48, 1, 'SAME'),      # generic 1x1
    (1, 40, 24, 16, 16, 3, 'SAME'),
]


@pytest.mark.parametrize('n,h,w,cin,cout,k,padding', MFMA_CASES)
def test_conv_mfma_vs_direct_and_oracle(ops, n, h, w, cin, cout, k, padding):
  import twingan_amd.ops as O
  rng = np.random.RandomState(2)
  x = bf16_round(rng.randn(n, h, w, cin))
  wt = rng.randn(k, k, cin, cout) / np.sqrt(k * k * cin)
  b = rng.randn(cout) * 0.1
  spec = O.ConvSpec(k, padding)
  assert O._mfma_ok(torch.bfloat16, cin, cout, spec, h, w)
  res = {}
  for algo in ('mfma', 'direct'):
    saved = O._mfma_ok
    if algo == 'direct':
      O._mfma_ok = lambda *a: False
    try:
      xd = to_dev(x, torch.bfloat16).requires_grad_(True)
      wd = to_dev(wt).requires_grad_(True)
      bd = to_dev(b).requires_grad_(True)
      with torch.no_grad():
        y = ops.conv2d(xd, wd, bd, k, padding, lrelu=True)
      # gradients through the linear part only: with LeakyReLU the two algorithms' 1-ulp differences in z flip
      # masks near zero, which at 16 pixels (the dense 4x4 VALID case) moves gw by several percent
      y_lin = ops.conv2d(xd, wd, bd, k, padding, lrelu=False)
      gy = bf16_round(np.random.RandomState(3).randn(*y.shape))
      y_lin.backward(to_dev(gy, torch.bfloat16))
      res[algo] = (host(y), host(xd.grad), host(wd.grad), host(bd.grad))
    finally:
      O._mfma_ok = saved
  names = ('y', 'gx', 'gw', 'gb')
  for i, nm in enumerate(names):
    e = rel_l2(res['mfma'][i], res['direct'][i])
    assert e < MFMA_VS_DIRECT_TOL, '%s mfma vs direct rel-L2 %.3e' % (nm, e)
  ref = N.leaky_relu(N.conv2d(x, bf16_round(wt), padding) + b)
  assert rel_l2(res['mfma'][0], ref) < BF16_FWD_TOL


def test_conv_mfma_transpose_detecting(ops):
  """A = identity-like weights with an ASYMMETRIC pattern catch swapped rows/cols or flipped taps."""
  cin = cout = 32
  x = np.zeros((1, 8, 8, cin))
  x[0, 2, 5, 3] = 1.0
  x[0, 6, 1, 17] = 2.0
  w = np.zeros((3, 3, cin, cout))
  w[0, 2, 3, 9] = 1.0        # tap (dy=-1, dx=+1): out[y, x] += in[y-1, x+1]
  w[2, 1, 17, 30] = 0.5      # tap (dy=+1, dx=0)
  y = ops.conv2d(to_dev(x, torch.bfloat16), to_dev(w), None, 3, 'SAME')
  ref = N.conv2d(x, w)
  assert np.array_equal(host(y), ref)
  assert host(y)[0, 3, 4, 9] == 1.0 and host(y)[0, 5, 1, 30] == 1.0


def test_conv_double_backward_matches_oracle(ops):
  """Second-order path used by WGAN-GP: d/dw of ||d conv / d x||^2."""
  rng = np.random.RandomState(4)
  x = rng.randn(2, 6, 6, 4)
  w = rng.randn(3, 3, 4, 5) * 0.3
  xd = to_dev(x).requires_grad_(True)
  wd = to_dev(w).requires_grad_(True)
  y = ops.conv2d(xd, wd, None, 3, 'SAME', lrelu=True)
  gx, = torch.autograd.grad(y, xd, grad_outputs=torch.ones_like(y), create_graph=True)
  pen = ops.gradient_penalty(gx.contiguous(), 10.0)
  pen.backward()
  xt = torch.from_numpy(x).requires_grad_(True)
  wt = torch.from_numpy(w).requires_grad_(True)
  yt = R.leaky_relu(R.conv2d(xt, wt, 'SAME'))
  gxt, = torch.autograd.grad(yt.sum(), xt, create_graph=True)
  pt = ((torch.sqrt((gxt ** 2).sum(dim=(1, 2, 3))) - 1) ** 2).mean() * 10.0
  pt.backward()
  assert abs(pen.item() - pt.item()) < 1e-4 * abs(pt.item())
  assert rel_l2(host(wd.grad), wt.grad.numpy()) < 1e-4


# ---------------------------------------------------------------------------------------------- rgb 1x1
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('cin,cout', [(3, 16), (3, 256), (16, 3), (256, 3), (3, 12), (20, 3)])
def test_pointwise_conv(ops, dtype, cin, cout):
  rng = np.random.RandomState(5)
  x = rng.randn(3, 7, 5, cin)
  w = rng.randn(1, 1, cin, cout) / np.sqrt(cin)
  b = rng.randn(cout) * 0.1
  if dtype == torch.bfloat16:
    x, wr = bf16_round(x), bf16_round(w)
  else:
    wr = w
  xd, wd, bd = to_dev(x, dtype).requires_grad_(True), to_dev(w).requires_grad_(True), to_dev(b).requires_grad_(True)
  y = ops.pointwise_conv(xd, wd, bd, lrelu=True)
  ref = N.leaky_relu(x @ wr[0, 0] + b)
  assert rel_l2(host(y), ref) < tol_for(dtype)
  gy = rng.randn(*ref.shape)
  if dtype == torch.bfloat16:
    gy = bf16_round(gy)
  y.backward(to_dev(gy, dtype))
  gpre = gy * np.where(ref > 0, 1.0, 0.2)
  if dtype == torch.bfloat16:
    gpre = bf16_round(gpre)
  assert rel_l2(host(xd.grad), gpre @ wr[0, 0].T) < tol_for(dtype, True)
  assert rel_l2(host(wd.grad)[0, 0], np.einsum('nhwc,nhwo->co', x, gpre)) < tol_for(dtype, True)
  assert rel_l2(host(bd.grad), gpre.sum(axis=(0, 1, 2))) < tol_for(dtype, True)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize('n,hw,cs,cb,small_in', [(2, 64, 3, 16, True), (16, 256, 3, 16, True), (3, 32, 3, 256, True),
                                                   (2, 64, 3, 16, False), (1, 6, 3, 16, True), (2, 8, 4, 16, True)])
def test_pointwise_rgb_filter_and_bias_gradient(ops, dtype, n, hw, cs, cb, small_in):
  """tg_pointwise_conv_bwd_weight(_bias) at the fromRGB / toRGB shapes (nets/pggan.py:233-240,176-200): the four-pixel RGB
  kernel (16-bit, 3 small channels, npix % 4 == 0; incl. the bench shape 16 x 256 x 256 x 3 -> 16) and the generic one,
  filter gradient with and without the bias gradient riding along, written and accumulated, against float64 sums of the
  same (rounded) inputs."""
  from twingan_amd import ops as O
  g = torch.Generator(device='cpu').manual_seed(7)
  small = torch.randn(n, hw, hw, cs, generator=g).to(dtype)
  big = torch.randn(n, hw, hw, cb, generator=g).to(dtype)
  x, gy = (small, big) if small_in else (big, small)
  cin, cout = x.shape[-1], gy.shape[-1]
  ref_w = np.einsum('pc,po->co', host(x).reshape(-1, cin), host(gy).reshape(-1, cout))
  ref_b = host(gy).reshape(-1, cout).sum(axis=0)
  xd, gyd = x.to(dev()), gy.to(dev())
  npix = n * hw * hw
  tol = 2e-5 if dtype == torch.float32 else 2e-4      # fp32 accumulation of exactly representable products
  for accumulate in (0, 1):
    gw = torch.full((cin, cout), 3.0 if accumulate else float('nan'), device=dev())
    O.call('tg_pointwise_conv_bwd_weight', xd.data_ptr(), gyd.data_ptr(), gw.data_ptr(), npix, cin, cout, accumulate,
           O._dt(xd), O._stream())
    assert rel_l2(host(gw) - (3.0 if accumulate else 0.0), ref_w) < tol
    if small_in:
      gw = torch.full((cin, cout), 3.0 if accumulate else float('nan'), device=dev())
      gb = torch.full((cout,), -2.0 if accumulate else float('nan'), device=dev())
      O.call('tg_pointwise_conv_bwd_weight_bias', xd.data_ptr(), gyd.data_ptr(), gw.data_ptr(), gb.data_ptr(), npix, cin, cout,
             accumulate, O._dt(xd), O._stream())
      assert rel_l2(host(gw) - (3.0 if accumulate else 0.0), ref_w) < tol
      assert rel_l2(host(gb) + (2.0 if accumulate else 0.0), ref_b) < tol


def test_pointwise_conv_bias_gradient_rides_in_the_filter_gradient(ops):
  """The discriminator's fromRGB under a trainer-style gradient sink: one tg_pointwise_conv_bwd_weight_bias launch instead of
  the filter gradient + a channel-sum pass -- same sums as the unfused autograd path."""
  from twingan_amd import ops as O
  g = torch.Generator(device='cpu').manual_seed(8)
  x = torch.randn(2, 32, 32, 3, generator=g).to(dev()).to(torch.bfloat16)
  w = (torch.randn(1, 1, 3, 16, generator=g) * 0.5).to(dev()).requires_grad_(True)
  b = (torch.randn(16, generator=g) * 0.1).to(dev()).requires_grad_(True)
  gy = torch.randn(2, 32, 32, 16, generator=g).to(dev()).to(torch.bfloat16)
  y = ops.pointwise_conv(x, w, b, lrelu=True)
  y.backward(gy)
  gw_ref, gb_ref = w.grad.clone(), b.grad.clone()
  sw, sb = torch.zeros_like(w), torch.zeros_like(b)
  O.GradSink.register(w, sw)
  O.GradSink.register(b, sb)
  try:
    y = ops.pointwise_conv(x, w, b, lrelu=True)
    y.grad_fn.tg_premasked = True      # as when the consumer conv's backward-data applied the LeakyReLU mask
    z = y.detach()
    gpre = (gy.float() * torch.where(z.float() > 0, 1.0, 0.2)).to(torch.bfloat16)
    torch.autograd.backward(y, gpre)
  finally:
    O.GradSink.unregister(w)
    O.GradSink.unregister(b)
  assert rel_l2(host(sw), host(gw_ref)) < 2e-3      # gpre is re-rounded to bf16 here, not in the unfused path
  assert rel_l2(host(sb), host(gb_ref)) < 2e-3


# ---------------------------------------------------------------------------------------------- norm_act
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('n,h,w,c,lrelu,pn', [(2, 8, 8, 16, True, True), (3, 4, 4, 256, True, True),
                                              (2, 16, 16, 32, True, False), (2, 8, 8, 3, False, False),
                                              (1, 32, 32, 64, True, True), (2, 5, 7, 8, True, True)])
def test_norm_act(ops, dtype, n, h, w, c, lrelu, pn):
  rng = np.random.RandomState(6)
  y = rng.randn(n, h, w, c) * 1.5 + 0.7
  gamma = 1.0 + 0.2 * rng.randn(c)
  beta = 0.1 * rng.randn(c)
  gz = rng.randn(n, h, w, c)
  if dtype == torch.bfloat16:
    y, gz = bf16_round(y), bf16_round(gz)
  yd = to_dev(y, dtype).requires_grad_(True)
  gd, bd = to_dev(gamma).requires_grad_(True), to_dev(beta).requires_grad_(True)
  z = ops.norm_act(yd, gd, bd, lrelu=lrelu, pixel_norm=pn)
  z.backward(to_dev(gz, dtype))
  yt = torch.from_numpy(y).requires_grad_(True)
  gt = torch.from_numpy(gamma).requires_grad_(True)
  bt = torch.from_numpy(beta).requires_grad_(True)
  zt = R.instance_norm(yt, gt, bt)
  if lrelu:
    zt = R.leaky_relu(zt)
  if pn:
    zt = R.pixel_norm(zt)
  zt.backward(torch.from_numpy(gz))
  assert rel_l2(host(z), zt.detach().numpy()) < tol_for(dtype)
  assert rel_l2(host(yd.grad), yt.grad.numpy()) < tol_for(dtype, True)
  assert rel_l2(host(gd.grad), gt.grad.numpy()) < tol_for(dtype, True)
  assert rel_l2(host(bd.grad), bt.grad.numpy()) < tol_for(dtype, True)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('n,h,w,c,lrelu,pn,pool,split', [(2, 8, 8, 16, True, True, False, None), (3, 4, 4, 256, True, True, False, 1),
                                                        (2, 16, 16, 32, True, False, True, None), (2, 8, 8, 3, False, False, False, None),
                                                        (4, 32, 32, 64, True, True, True, 2), (2, 5, 7, 8, True, True, False, None)])
def test_layer_norm_act(ops, dtype, n, h, w, c, lrelu, pn, pool, split):
  """ops.layer_norm_act = tf.contrib.layers.layer_norm (statistics of one image over (H, W, C), gamma / beta per channel,
  epsilon 1e-12; nets/pggan_utils.py:189-197) + LeakyReLU + pixel norm (+ the 2x2 pool), with per-domain parameters
  (images [split, n) use the second pair).  Outputs and all gradients against the same composition in float64."""
  rng = np.random.RandomState(61)
  y = rng.randn(n, h, w, c) * 1.5 + 0.7 + 0.5 * rng.randn(1, 1, 1, c)
  gam = [1.0 + 0.2 * rng.randn(c) for _ in range(2)]
  bet = [0.1 * rng.randn(c) for _ in range(2)]
  gz = rng.randn(n, h, w, c)
  gzp = rng.randn(n, h // 2, w // 2, c)
  if dtype == torch.bfloat16:
    y, gz, gzp = bf16_round(y), bf16_round(gz), bf16_round(gzp)
  yd = to_dev(y, dtype).requires_grad_(True)
  gd = [to_dev(g).requires_grad_(True) for g in gam]
  bd = [to_dev(b).requires_grad_(True) for b in bet]
  two = split is not None
  out = ops.layer_norm_act(yd, gd[0], bd[0], lrelu=lrelu, pixel_norm=pn, pool=pool, gamma2=gd[1] if two else None,
                           beta2=bd[1] if two else None, split=split)
  yt = torch.from_numpy(y).requires_grad_(True)
  gt = [torch.from_numpy(g).requires_grad_(True) for g in gam]
  bt = [torch.from_numpy(b).requires_grad_(True) for b in bet]
  mean = yt.mean(dim=(1, 2, 3), keepdim=True)
  var = ((yt - mean) ** 2).mean(dim=(1, 2, 3), keepdim=True)
  sel = (torch.arange(n) >= (split if two else n)).view(n, 1, 1, 1)
  zt = (yt - mean) * torch.rsqrt(var + 1e-12) * torch.where(sel, gt[1], gt[0]) + torch.where(sel, bt[1], bt[0])
  if lrelu:
    zt = R.leaky_relu(zt)
  if pn:
    zt = R.pixel_norm(zt)
  if pool:
    z, zp = out
    ztp = R.avg_pool2(zt)
    torch.autograd.backward([z, zp], [to_dev(gz, dtype), to_dev(gzp, dtype)])
    torch.autograd.backward([zt, ztp], [torch.from_numpy(gz), torch.from_numpy(gzp)])
    assert rel_l2(host(zp), ztp.detach().numpy()) < tol_for(dtype)
  else:
    z = out
    z.backward(to_dev(gz, dtype))
    zt.backward(torch.from_numpy(gz))
  assert rel_l2(host(z), zt.detach().numpy()) < tol_for(dtype)
  assert rel_l2(host(yd.grad), yt.grad.numpy()) < tol_for(dtype, True)
  for i in range(2 if two else 1):
    assert rel_l2(host(gd[i].grad), gt[i].grad.numpy()) < tol_for(dtype, True), i
    assert rel_l2(host(bd[i].grad), bt[i].grad.numpy()) < tol_for(dtype, True), i


def test_instance_norm_large_mean_is_stable(ops):
  """Shifted-sum statistics: a large common offset must not destroy the variance in fp32."""
  rng = np.random.RandomState(7)
  y = rng.randn(1, 64, 64, 8) * 0.01 + 100.0
  z = ops.norm_act(to_dev(y), to_dev(np.ones(8)), to_dev(np.zeros(8)), lrelu=False, pixel_norm=False)
  ref = N.instance_norm(np.asarray(to_dev(y).cpu().numpy(), np.float64), 1.0, 0.0)
  assert rel_l2(host(z), ref) < 1e-3


# ---------------------------------------------------------------------------------------------- resampling
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('c0,c1', [(8, 8), (16, 0), (3, 0), (256, 256), (5, 3)])
def test_upsample_concat(ops, dtype, c0, c1):
  rng = np.random.RandomState(8)
  x0 = rng.randn(2, 3, 5, c0)
  x1 = rng.randn(2, 6, 10, c1) if c1 else None
  if dtype == torch.bfloat16:
    x0 = bf16_round(x0)
    x1 = bf16_round(x1) if c1 else None
  else:                                   # exact-copy check: start from fp32-representable values
    x0 = x0.astype(np.float32).astype(np.float64)
    x1 = x1.astype(np.float32).astype(np.float64) if c1 else None
  a = to_dev(x0, dtype).requires_grad_(True)
  b = to_dev(x1, dtype).requires_grad_(True) if c1 else None
  out = ops.upsample2x_concat(a, b)
  ref = N.upsample2x(x0)
  if c1:
    ref = np.concatenate([ref, x1], axis=3)
  assert np.array_equal(host(out), ref)
  go = rng.randn(*ref.shape)
  go = bf16_round(go) if dtype == torch.bfloat16 else go.astype(np.float32).astype(np.float64)
  out.backward(to_dev(go, dtype))
  g0 = go[..., :c0].reshape(2, 3, 2, 5, 2, c0).sum(axis=(2, 4))
  assert rel_l2(host(a.grad), g0) < tol_for(dtype)
  if c1:
    assert np.array_equal(host(b.grad), go[..., c0:])


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('c', [16, 3, 256])
def test_avg_pool_and_double_backward(ops, dtype, c):
  rng = np.random.RandomState(9)
  x = rng.randn(2, 6, 8, c)
  if dtype == torch.bfloat16:
    x = bf16_round(x)
  xd = to_dev(x, dtype).requires_grad_(True)
  y = ops.avg_pool2(xd)
  assert rel_l2(host(y), N.avg_pool2(x)) < tol_for(dtype)
  gy = rng.randn(*y.shape)
  if dtype == torch.bfloat16:
    gy = bf16_round(gy)
  gyd = to_dev(gy, dtype).requires_grad_(True)
  gx, = torch.autograd.grad(y, xd, grad_outputs=gyd, create_graph=True)
  assert rel_l2(host(gx), N.upsample2x(gy) * 0.25) < tol_for(dtype)
  v = rng.randn(*x.shape)
  if dtype == torch.bfloat16:
    v = bf16_round(v)
  ggy, = torch.autograd.grad(gx, gyd, grad_outputs=to_dev(v, dtype))
  assert rel_l2(host(ggy), N.avg_pool2(v)) < tol_for(dtype)       # adjoint of the adjoint


def test_lerp_and_cast(ops):
  rng = np.random.RandomState(10)
  a, b = rng.randn(2, 4, 4, 6), rng.randn(2, 4, 4, 6)
  ad, bd = to_dev(a).requires_grad_(True), to_dev(b).requires_grad_(True)
  out = ops.lerp(ad, bd, 0.3)
  assert rel_l2(host(out), N.lerp(a, b, 0.3)) < F32_TOL
  out.backward(torch.ones_like(out))
  assert rel_l2(host(ad.grad), np.full(a.shape, 0.3)) < F32_TOL
  assert rel_l2(host(bd.grad), np.full(a.shape, 0.7)) < F32_TOL
  c = ops.cast(to_dev(a), torch.bfloat16)
  assert c.dtype == torch.bfloat16 and np.array_equal(host(c), bf16_round(a))


# ---------------------------------------------------------------------------------------------- mbstd
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('n,c', [(5, 8), (16, 256), (3, 32)])
def test_mbstd_fwd_bwd_bwdbwd(ops, dtype, n, c):
  from twingan_amd.params import mbstd_cpad
  rng = np.random.RandomState(11)
  x = rng.randn(n, 4, 4, c)
  if dtype == torch.bfloat16:
    x = bf16_round(x)
  cpad = mbstd_cpad(c)
  xd = to_dev(x, dtype).requires_grad_(True)
  out = ops.minibatch_state_concat(xd, cpad)
  assert out.shape == (n, 4, 4, cpad)
  eps = 1e-8 if dtype == torch.float32 else 1e-6
  ref = N.minibatch_state_concat(x, eps)
  assert rel_l2(host(out)[..., :c + 1], ref) < tol_for(dtype)
  assert np.all(host(out)[..., c + 1:] == 0)
  # first + second order vs torch autograd on the oracle
  go = rng.randn(n, 4, 4, cpad)
  v = rng.randn(n, 4, 4, c)
  if dtype == torch.bfloat16:
    go, v = bf16_round(go), bf16_round(v)
  god = to_dev(go, dtype).requires_grad_(True)
  gx, = torch.autograd.grad(out, xd, grad_outputs=god, create_graph=True)
  ggo, gx2 = torch.autograd.grad(gx, [god, xd], grad_outputs=to_dev(v, dtype))

  xt = torch.from_numpy(x).requires_grad_(True)
  got = torch.from_numpy(go[..., :c + 1].copy()).requires_grad_(True)
  mean = xt.mean(dim=0, keepdim=True)
  std = torch.sqrt(((xt - mean) ** 2).mean(dim=0, keepdim=True) + eps)
  outt = torch.cat([xt, std.mean().reshape(1, 1, 1, 1).expand(n, 4, 4, 1)], dim=3)
  gxt, = torch.autograd.grad(outt, xt, grad_outputs=got, create_graph=True)
  ggot, gx2t = torch.autograd.grad(gxt, [got, xt], grad_outputs=torch.from_numpy(v))
  gtol = tol_for(dtype, True)
  assert rel_l2(host(gx), gxt.detach().numpy()) < gtol
  assert rel_l2(host(ggo)[..., :c + 1], ggot.numpy()) < gtol
  assert rel_l2(host(gx2), gx2t.numpy()) < (gtol if dtype == torch.float32 else 0.1)


# ---------------------------------------------------------------------------------------------- dense, losses
def test_fully_connected_and_grads(ops):
  """layers.fully_connected: (a) inside ops.second_order() -- the gradient-penalty pass -- the twice-differentiable
  composition (cast, GEMM, bias add); (b) in first-order passes ONE launch each way (tg_fc_fwd / tg_fc_bwd: FcFn), with
  the parameter gradients written or, under gradient sinks, added in place; fp32 and 16-bit features."""
  rng = np.random.RandomState(12)
  x, w, b = rng.randn(6, 40), rng.randn(40, 3), rng.randn(3)
  xd, wd, bd = to_dev(x).requires_grad_(True), to_dev(w).requires_grad_(True), to_dev(b).requires_grad_(True)
  with ops.second_order():
    y = ops.fully_connected(xd, wd, bd)
  assert rel_l2(host(y), N.fully_connected(x, w, b)) < F32_TOL
  g = rng.randn(6, 3)
  gx, = torch.autograd.grad(y, xd, grad_outputs=to_dev(g), create_graph=True)
  assert rel_l2(host(gx), g @ w.T) < F32_TOL
  v = rng.randn(6, 40)
  (gx * to_dev(v)).sum().backward()            # d/dw of <v, g w^T> = v^T g
  assert rel_l2(host(wd.grad), v.T @ g) < F32_TOL
  # (b) first order
  for dtype in (torch.float32, torch.bfloat16):
    xr = bf16_round(x) if dtype == torch.bfloat16 else x
    xd = to_dev(xr, dtype).requires_grad_(True)
    wd, bd = to_dev(w).requires_grad_(True), to_dev(b).requires_grad_(True)
    y = ops.fully_connected(xd, wd, bd)
    assert type(y.grad_fn).__name__ == 'FcFnBackward' and y.dtype == torch.float32
    assert rel_l2(host(y), N.fully_connected(xr, w, b)) < F32_TOL
    y.backward(to_dev(g))
    assert rel_l2(host(xd.grad), g @ w.T) < tol_for(dtype, True)
    assert rel_l2(host(wd.grad), xr.T @ g) < F32_TOL and rel_l2(host(bd.grad), g.sum(0)) < F32_TOL
    # gradient sinks: added into the caller's buffers, twice
    ops.GradSink.clear()
    ws, bs = torch.ones_like(wd), torch.ones_like(bd)
    ops.GradSink.register(wd, ws)
    ops.GradSink.register(bd, bs)
    wd.grad = bd.grad = None
    for _ in range(2):
      ops.fully_connected(xd, wd, bd).backward(to_dev(g))
    ops.GradSink.clear()
    assert wd.grad is None and bd.grad is None
    assert rel_l2(host(ws), 1.0 + 2.0 * (xr.T @ g)) < F32_TOL and rel_l2(host(bs), 1.0 + 2.0 * g.sum(0)) < F32_TOL


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_losses(ops, dtype):
  rng = np.random.RandomState(13)
  a, b = rng.rand(2, 8, 8, 3), rng.rand(2, 8, 8, 3)
  if dtype == torch.bfloat16:
    a, b = bf16_round(a), bf16_round(b)
  ad, bd = to_dev(a, dtype).requires_grad_(True), to_dev(b, dtype).requires_grad_(True)
  l1 = ops.abs_diff_mean(ad, bd, 0.1)
  assert abs(l1.item() - N.absolute_difference(a, b, 0.1)) < 1e-6
  (l1 * 3.0).backward()
  assert rel_l2(host(ad.grad), 0.3 * np.sign(a - b) / a.size) < tol_for(dtype)
  assert rel_l2(host(bd.grad), -0.3 * np.sign(a - b) / a.size) < tol_for(dtype)
  p = rng.randn(7, 1)
  pd = to_dev(p).requires_grad_(True)
  m = ops.mean(pd, -1.0)
  assert abs(m.item() - N.wgan_g_loss(p)) < 1e-6
  m.backward()
  assert rel_l2(host(pd.grad), np.full(p.shape, -1.0 / 7)) < F32_TOL
  g = rng.randn(4, 8, 8, 3) * 0.05
  if dtype == torch.bfloat16:
    g = bf16_round(g)
  gd = to_dev(g, dtype).requires_grad_(True)
  gp = ops.gradient_penalty(gd, 10.0)
  assert abs(gp.item() - N.gradient_penalty(g, 10.0)) < 1e-4 * N.gradient_penalty(g, 10.0)
  gp.backward()
  gt = torch.from_numpy(g).requires_grad_(True)
  (((torch.sqrt((gt ** 2).sum(dim=(1, 2, 3))) - 1) ** 2).mean() * 10.0).backward()
  assert rel_l2(host(gd.grad), gt.grad.numpy()) < tol_for(dtype, True)


def test_gp_unit_linear_critic_is_zero(ops):
  w = np.random.RandomState(14).randn(4, 4, 3)
  w /= np.linalg.norm(w)
  g = np.tile(w[None], (5, 1, 1, 1))
  assert ops.gradient_penalty(to_dev(g), 10.0).item() < 1e-10


def test_sample_lerp(ops):
  rng = np.random.RandomState(15)
  x, y, a = rng.rand(4, 8, 8, 3), rng.rand(4, 8, 8, 3), rng.rand(4)
  out = ops.sample_lerp(to_dev(x), to_dev(y), to_dev(a))
  assert rel_l2(host(out), x + a[:, None, None, None] * (y - x)) < F32_TOL


def test_adam_kernel_tf_semantics():
  from twingan_amd._lib import call
  rng = np.random.RandomState(16)
  th, g = rng.randn(1000), rng.randn(1000)
  m, v = np.zeros(1000), np.zeros(1000)
  thd, md, vd = to_dev(th), to_dev(m), to_dev(v)
  for t in (1, 2, 3):
    g = rng.randn(1000)
    lr_t = 1e-4 * np.sqrt(1 - 0.99 ** t) / (1 - 0.5 ** t)
    call('tg_adam_step', thd.data_ptr(), to_dev(g).data_ptr(), md.data_ptr(), vd.data_ptr(), None, 1000, float(lr_t),
         None, 0.5, 0.99, 1e-8, 1.0, torch.cuda.current_stream().cuda_stream)
    th, m, v = N.adam_step(th, g, m, v, t)
  assert rel_l2(host(thd), th) < 1e-6
  assert rel_l2(host(md), m) < 1e-5 and rel_l2(host(vd), v) < 1e-5


def test_adam_device_tick_matches_host_schedule():
  """tg_adam_tick: the shared step counter and bias-corrected rate kept on the device (graph replay)."""
  from twingan_amd._lib import call
  rng = np.random.RandomState(17)
  th = rng.randn(512)
  m, v = np.zeros(512), np.zeros(512)
  thd, md, vd = to_dev(th), to_dev(m), to_dev(v)
  step = torch.zeros(1, dtype=torch.int64, device='cuda:0')
  lr_t = torch.zeros(1, dtype=torch.float32, device='cuda:0')
  st = torch.cuda.current_stream().cuda_stream
  for t in (1, 2, 3, 4):
    g = rng.randn(512)
    call('tg_adam_tick', step.data_ptr(), lr_t.data_ptr(), 1e-4, 0.5, 0.99, st)
    call('tg_adam_step', thd.data_ptr(), to_dev(g).data_ptr(), md.data_ptr(), vd.data_ptr(), None, 512, 0.0,
         lr_t.data_ptr(), 0.5, 0.99, 1e-8, 1.0, st)
    th, m, v = N.adam_step(th, g, m, v, t)
    assert int(step.item()) == t
    assert abs(lr_t.item() - 1e-4 * np.sqrt(1 - 0.99 ** t) / (1 - 0.5 ** t)) < 1e-10
  assert rel_l2(host(thd), th) < 1e-6


def test_errors_are_loud(ops):
  from twingan_amd._lib import TgError
  with pytest.raises(TgError):
    ops.conv2d(torch.zeros(1, 4, 4, 4), torch.zeros(3, 3, 4, 4))          # CPU tensors: no fallback
  with pytest.raises(TgError):
    ops.pointwise_conv(to_dev(np.zeros((1, 2, 2, 8))), to_dev(np.zeros((1, 1, 8, 8))))


# ---------------------------------------------------------------------------------------------- batched passes
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_norm_act_domain_split_equals_separate_calls(ops, dtype):
  """Two reference passes (domains s and t: different gamma/beta, nets/pggan_utils.py:102-113) batched along N."""
  rng = np.random.RandomState(21)
  x = to_dev(bf16_round(rng.randn(5, 8, 8, 16)), dtype)
  ga, be, ga2, be2 = (to_dev(a) for a in (1 + 0.1 * rng.randn(16), 0.1 * rng.randn(16), 1 + 0.2 * rng.randn(16),
                                          0.3 * rng.randn(16)))
  gz = to_dev(bf16_round(rng.randn(5, 8, 8, 16)), dtype)
  split = 2
  leaves = [t.clone().requires_grad_(True) for t in (x, ga, be, ga2, be2)]
  z = ops.norm_act(leaves[0], leaves[1], leaves[2], gamma2=leaves[3], beta2=leaves[4], split=split)
  z.backward(gz)
  parts, grads = [], []
  for lo, hi, g_, b_ in ((0, split, ga, be), (split, 5, ga2, be2)):
    xi = x[lo:hi].clone().requires_grad_(True)
    gi, bi = g_.clone().requires_grad_(True), b_.clone().requires_grad_(True)
    zi = ops.norm_act(xi, gi, bi)
    zi.backward(gz[lo:hi].contiguous())
    parts.append(zi)
    grads.append((xi.grad, gi.grad, bi.grad))
  tol = 1e-6 if dtype == torch.float32 else 1e-2
  assert rel_l2(host(z), host(torch.cat(parts))) < tol
  assert rel_l2(host(leaves[0].grad), host(torch.cat([grads[0][0], grads[1][0]]))) < tol
  for i, (a, b) in enumerate(((leaves[1].grad, grads[0][1]), (leaves[2].grad, grads[0][2]), (leaves[3].grad, grads[1][1]),
                              (leaves[4].grad, grads[1][2]))):
    assert rel_l2(host(a), host(b)) < 1e-4, i


def test_mbstd_groups_equal_separate_calls(ops):
  """Three discriminator calls batched along N keep their own minibatch-stddev statistic (pggan_utils.py:353-366)."""
  rng = np.random.RandomState(22)
  x = to_dev(rng.randn(6, 4, 4, 16)).requires_grad_(True)
  go = to_dev(rng.randn(6, 4, 4, 24))
  out = ops.minibatch_state_concat(x, 24, groups=3)
  out.backward(go)
  for g in range(3):
    xi = x.detach()[2 * g:2 * g + 2].clone().requires_grad_(True)
    oi = ops.minibatch_state_concat(xi, 24)
    oi.backward(go[2 * g:2 * g + 2].contiguous())
    assert rel_l2(host(out[2 * g:2 * g + 2]), host(oi)) < 1e-6
    assert rel_l2(host(x.grad[2 * g:2 * g + 2]), host(xi.grad)) < 1e-5


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_upsample_concat_group_permutation(ops, dtype):
  """Four generator passes batched along N read the skips of the [s; t] encoder batch as groups (t, s, s, t)."""
  rng = np.random.RandomState(23)
  gsz, perm = 2, (1, 0, 0, 1)
  x0 = to_dev(bf16_round(rng.randn(8, 3, 4, 8)), dtype).requires_grad_(True)
  x1 = to_dev(bf16_round(rng.randn(4, 6, 8, 16)), dtype).requires_grad_(True)
  out = ops.upsample2x_concat(x0, x1, gsz, perm)
  go = to_dev(bf16_round(rng.randn(8, 6, 8, 24)), dtype)
  out.backward(go)
  a = x0.detach().clone().requires_grad_(True)
  b = x1.detach().clone().requires_grad_(True)
  bs, bt = b[:2], b[2:]
  ref = ops.upsample2x_concat(a, torch.cat([bt, bs, bs, bt], dim=0).contiguous())
  ref.backward(go)
  assert np.array_equal(host(out), host(ref))
  assert rel_l2(host(x0.grad), host(a.grad)) < 1e-6
  assert rel_l2(host(x1.grad), host(b.grad)) < (1e-6 if dtype == torch.float32 else 1e-2)


# ---------------------------------------------------------------------------------------------- fused pooling
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_conv_pool_fused_backward_equals_composition(ops, dtype):
  """conv + bias + LeakyReLU followed by tf.nn.avg_pool (nets/pggan.py:304-306): the fused op's backward (pool folded
  into the LeakyReLU / bias-gradient kernel) against conv2d -> avg_pool2, with and without the pre-pool output used."""
  rng = np.random.RandomState(31)
  x0 = bf16_round(rng.randn(2, 8, 16, 16))
  w0 = rng.randn(3, 3, 16, 16) / 12.0
  b0 = rng.randn(16) * 0.1
  gp = bf16_round(rng.randn(2, 4, 8, 16))
  gf = bf16_round(rng.randn(2, 8, 16, 16))
  for use_full in (False, True):
    res = []
    for fused in (True, False):
      x = to_dev(x0, dtype).requires_grad_(True)
      w = to_dev(w0).requires_grad_(True)
      b = to_dev(b0).requires_grad_(True)
      if fused:
        z, zp = ops.conv2d(x, w, b, 3, 'SAME', lrelu=True, pool=True)
      else:
        z = ops.conv2d(x, w, b, 3, 'SAME', lrelu=True)
        zp = ops.avg_pool2(z)
      outs, grads = [zp], [to_dev(gp, dtype)]
      if use_full:
        outs.append(z)
        grads.append(to_dev(gf, dtype))
      torch.autograd.backward(outs, grads)
      res.append((host(zp), host(x.grad), host(w.grad), host(b.grad)))
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for i, nm in enumerate(('zp', 'gx', 'gw', 'gb')):
      assert rel_l2(res[0][i], res[1][i]) < tol, (use_full, nm, rel_l2(res[0][i], res[1][i]))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_norm_act_pool_fused_backward_equals_composition(ops, dtype):
  """Last layer of an encoder block + avg_pool (nets/pggan.py:466-468) with the UNet skip also consuming z."""
  rng = np.random.RandomState(32)
  y0 = bf16_round(rng.randn(3, 8, 8, 16) * 1.5 + 0.3)
  ga0, be0 = 1 + 0.1 * rng.randn(16), 0.1 * rng.randn(16)
  gp = bf16_round(rng.randn(3, 4, 4, 16))
  gf = bf16_round(rng.randn(3, 8, 8, 16))
  for use_full in (False, True):
    res = []
    for fused in (True, False):
      y = to_dev(y0, dtype).requires_grad_(True)
      ga, be = to_dev(ga0).requires_grad_(True), to_dev(be0).requires_grad_(True)
      if fused:
        z, zp = ops.norm_act(y, ga, be, pool=True)
      else:
        z = ops.norm_act(y, ga, be)
        zp = ops.avg_pool2(z)
      outs, grads = [zp], [to_dev(gp, dtype)]
      if use_full:
        outs.append(z)
        grads.append(to_dev(gf, dtype))
      torch.autograd.backward(outs, grads)
      res.append((host(zp), host(y.grad), host(ga.grad), host(be.grad)))
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for i, nm in enumerate(('zp', 'gy', 'ggamma', 'gbeta')):
      assert rel_l2(res[0][i], res[1][i]) < tol, (use_full, nm, rel_l2(res[0][i], res[1][i]))


def test_multi_tensor_pack_matches_single_packs(ops):
  """PackCache.refresh (one tg_conv2d_pack_weights_multi launch) == one tg_conv2d_pack_weights per pack."""
  from twingan_amd.ops import PackCache
  PackCache.clear()
  g = torch.Generator().manual_seed(5)
  ws = [torch.randn(3, 3, ci, co, generator=g).to(dev()) for ci, co in ((16, 32), (40, 16), (64, 64))]
  ws.append(torch.randn(4, 4, 32, 32, generator=g).to(dev()))
  # 8x8 maps (conv_img): fragment-ordered packs, tg_conv2d_pack_layout = 1 -- a 32-channel kernel (two K chunks) and wider
  on8 = [torch.randn(3, 3, ci, co, generator=g).to(dev()) for ci, co in ((32, 64), (128, 32))]
  ws += on8
  for w in ws:
    PackCache.register(w)
  packs = []
  for w in ws:
    k = w.shape[0]
    hw = 8 if any(w is v for v in on8) else (16 if k == 3 else 4)
    x = torch.randn(2, hw, hw, w.shape[2], generator=g).to(dev()).bfloat16()
    if hw == 8:
      from twingan_amd import _lib
      d8 = ops._desc(x.shape, w.shape[3], ops.ConvSpec(3, 'SAME'), x.dtype, 0)
      import ctypes
      assert _lib.load().tg_conv2d_pack_layout(ctypes.byref(d8), 0) == 1 and _lib.load().tg_conv2d_pack_layout(ctypes.byref(d8), 1) == 1
    spec = ops.ConvSpec(k, 'SAME' if k == 3 else 'VALID')
    d = ops._desc(x.shape, w.shape[3], spec, x.dtype, 0)
    for mode in (0, 1):
      packs.append((w, mode, PackCache.get(w, d, mode)))
  want = [p.clone() for _, _, p in packs]
  with torch.no_grad():
    for w in ws:
      w.mul_(2.0)
  PackCache.version += 1
  assert PackCache.refresh(ws) == len(packs)
  torch.cuda.synchronize()
  for (w, mode, p), ref in zip(packs, want):
    assert torch.equal(p.float(), (ref.float() * 2.0)), (tuple(w.shape), mode)
  PackCache.clear()


@pytest.mark.parametrize('n,h,c0,c1,cout,gsz,perm', [(2, 8, 32, 32, 16, 0, ()), (4, 16, 64, 64, 64, 1, (1, 0, 0, 1)),
                                                        (2, 8, 64, 32, 128, 0, ())])
def test_upcat_conv_matches_materialised_path(ops, n, h, c0, c1, cout, gsz, perm):
  """conv3x3(concat(up2(x0), skip)) read from the two sources (tg_conv2d_upcat_fwd / _bwd_weight) == the same conv
  over the materialised tg_upsample2x_concat_fwd tensor: output, both input gradients and the filter gradient."""
  g = torch.Generator().manual_seed(11)
  n1 = (max(perm) + 1) * gsz if gsz else n
  x0 = torch.randn(n, h, h, c0, generator=g).to(dev()).bfloat16().requires_grad_(True)
  x1 = torch.randn(n1, 2 * h, 2 * h, c1, generator=g).to(dev()).bfloat16().requires_grad_(True)
  w = (torch.randn(3, 3, c0 + c1, cout, generator=g) * (2.0 / (9 * (c0 + c1))) ** 0.5).to(dev()).requires_grad_(True)
  gy = torch.randn(n, 2 * h, 2 * h, cout, generator=g).to(dev()).bfloat16()
  assert ops.upcat_conv_supported(x0, x1, w)
  y_ref = ops.conv2d(ops.upsample2x_concat(x0, x1, gsz, perm), w, None, 3, 'SAME')
  y_ref.backward(gy)
  ref = [t.grad.clone() for t in (x0, x1, w)]
  for t in (x0, x1, w):
    t.grad = None
  y = ops.upcat_conv(x0, x1, w, gsz, perm)
  y.backward(gy)
  assert rel_l2(host(y), host(y_ref)) < 1e-6           # same kernel arithmetic, same accumulation order per tile
  # input gradients: with the concat adjoint in the backward-data epilogue (tg_conv2d_upcat_bwd_data, 16 x 16 maps and up)
  # the fp32 sums are rounded ONCE; the materialised path rounds the concat-layout gradient and then its 2x2 / group sums
  gtol = 4e-3 if (ops.USE_UPCAT_BWD_FUSED and 2 * h >= 16) else 1e-6
  assert rel_l2(host(x0.grad), host(ref[0])) < gtol
  assert rel_l2(host(x1.grad), host(ref[1])) < gtol
  assert rel_l2(host(w.grad), host(ref[2])) < 1e-5


UPBWD_CASES = [
    # n, hw, c0, c1, cout, gsz, perm                      kernel the shape dispatches
    (4, 256, 32, 32, 16, 1, (1, 0, 0, 1)),              # conv_tile_wres_kernel<3,16,64,1,upboth>: a 32 + 32 concat in one block
    (8, 128, 64, 64, 32, 2, (1, 0, 0, 1)),              # conv_tile_wres_kernel<3,32,64,1,upbwd>
    (8, 128, 32, 32, 32, 2, (0, 1, 1, 0)),              # conv_tile_wres_kernel<3,32,32,1,upbwd>
    (8, 128, 64, 64, 16, 0, ()),                        # conv_tile_wres_kernel<3,16,64,1,upbwd>, no groups
    (4, 128, 64, 64, 16, 1, (0, 0, 0, 1)),              # conv_tile_kernel<3,16,64,1,upbwd>, three sources / one source
    (8, 64, 128, 128, 64, 2, (1, 0, 0, 1)),             # conv_tile_kernel<3,32,64,2,upbwd>
    (16, 64, 64, 64, 32, 4, (1, 0, 0, 1)),              # conv_tile_kernel<3,32,64,1,upbwd>
    (8, 32, 256, 256, 128, 2, (1, 0, 0, 1)),            # conv_tile_kernel<3,32,32,2,upbwd>
    (2, 16, 32, 32, 32, 1, (1, 1)),                     # conv_tile_kernel<3,32,32,1,upbwd>, skip group 0 unread (zeros)
    (2, 32, 32, 32, 16, 0, ()),                         # conv_tile_kernel<3,16,32,1,upbwd>
    (3, 48, 32, 64, 40, 0, ()),                         # ragged: 48 x 48 map, cout 40 (cin_pad 48), c1 = 64
    (4, 256, 32, 64, 16, 1, (1, 0, 0, 1)),              # conv_tile_wres_kernel<3,16,32,1,upbwd> (c1 = 64: two-block form)
    (4, 256, 32, 32, 16, 1, (0, 0, 0, 1)),              # upboth: three sources / one source
    (4, 256, 32, 32, 16, 1, (1, 1, 1, 1)),              # upboth: skip image 0 unread (zeros), image 1 read four times
    (6, 256, 32, 32, 16, 0, ()),                        # upboth without groups
]
UPBWD_KERNELS = ['conv_tile_wres_kernel<3,16,64,1,upboth>', 'conv_tile_wres_kernel<3,32,64,1,upbwd>', 'conv_tile_wres_kernel<3,32,32,1,upbwd>',
                 'conv_tile_wres_kernel<3,16,64,1,upbwd>', 'conv_tile_kernel<3,16,64,1,upbwd>', 'conv_tile_kernel<3,32,64,2,upbwd>',
                 'conv_tile_kernel<3,32,64,1,upbwd>', 'conv_tile_kernel<3,32,32,2,upbwd>', 'conv_tile_kernel<3,32,32,1,upbwd>',
                 'conv_tile_kernel<3,16,32,1,upbwd>', None, 'conv_tile_wres_kernel<3,16,32,1,upbwd>',
                 'conv_tile_wres_kernel<3,16,64,1,upboth>', 'conv_tile_wres_kernel<3,16,64,1,upboth>',
                 'conv_tile_wres_kernel<3,16,64,1,upboth>']


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('case', range(len(UPBWD_CASES)))
def test_upcat_backward_data_epilogue(ops, dtype, case):
  """tg_conv2d_upcat_bwd_data: the input gradient of conv3x3(concat(up2(x0), skip)) (nets/pggan.py:69-78,
  nets/pggan_utils.py:281-298,349-350) written by the backward-data kernel itself -- 2x2 sums of the first c0 channels into
  g0, the rest summed over the groups that read a skip image into g1 -- for every kernel variant the dispatch can pick, with
  and without group permutations (incl. a skip group nobody reads and one read three times), against float64 sums of the
  float64 backward-data of the same rounded operands (one rounding: <= 2e-3; fp16 3e-4) and against the composed path."""
  from twingan_amd import ops as O, _lib
  n, hw, c0, c1, cout, gsz, perm = UPBWD_CASES[case]
  g = torch.Generator().manual_seed(100 + case)
  n1 = (max(perm) + 1) * gsz if gsz else n
  gy = torch.randn(n, hw, hw, cout, generator=g).to(dtype)
  w = (torch.randn(3, 3, c0 + c1, cout, generator=g) * (2.0 / (9 * cout)) ** 0.5).to(dtype).float()
  gyd, wd = gy.to(dev()), w.to(dev())
  spec = O.ConvSpec(3, 'SAME')
  d = O._desc((n, hw, hw, c0 + c1), cout, spec, dtype, 0)
  g0 = torch.full((n, hw // 2, hw // 2, c0), float('nan'), dtype=dtype, device=dev())
  g1 = torch.full((n1, hw, hw, c1), float('nan'), dtype=dtype, device=dev())
  wpack = O.PackCache.get(wd, d, 1)      # held in a local: an uncached pack must outlive the launch
  O.call('tg_conv2d_upcat_bwd_data', gyd.data_ptr(), wpack.data_ptr(), g0.data_ptr(), g1.data_ptr(), n, hw, hw,
         c0, c1, cout, gsz, O._pack_perm(perm), O._dt(gyd), O._stream())
  sym = _lib.load().tg_last_kernel().decode()
  want = UPBWD_KERNELS[case]
  if want is not None:
    assert sym == (want if dtype == torch.bfloat16 else want.replace('upbwd', 'upbwd,f16').replace('upboth', 'upboth,f16')), sym
  assert bool(torch.isfinite(g0.float()).all()) and bool(torch.isfinite(g1.float()).all())
  # composed path on the device: same values up to the extra rounding of the concat-layout tensor
  gcat = O.conv_bwd_data_raw(gyd, wd, (n, hw, hw, c0 + c1), spec)
  r0, r1 = torch.empty_like(g0), torch.empty_like(g1)
  O.call('tg_upsample2x_concat_bwd', gcat.data_ptr(), r0.data_ptr(), r1.data_ptr(), n, hw // 2, hw // 2, c0, c1, gsz,
         O._pack_perm(perm), O._dt(gcat), O._stream())
  ctol = 4e-3 if dtype == torch.bfloat16 else 6e-4
  unread = [sg for sg in range(n1 // gsz) if sg not in perm] if gsz else []
  for sg in unread:      # nobody read this skip group: exact zeros from both
    assert float(g1[sg * gsz:(sg + 1) * gsz].abs().max()) == 0.0 and float(r1[sg * gsz:(sg + 1) * gsz].abs().max()) == 0.0
  assert rel_l2(host(g0), host(r0)) < ctol, ('g0 vs composed', rel_l2(host(g0), host(r0)))
  assert rel_l2(host(g1), host(r1)) < ctol, ('g1 vs composed', rel_l2(host(g1), host(r1)))
  # float64 reference for one generator image (g0) and one skip image (g1)
  wn, gyn = host(wd), host(gyd)
  tol = 2e-3 if dtype == torch.bfloat16 else 3e-4
  i = n - 1
  gc = N.conv2d_bwd_data_gemm(gyn[i:i + 1], wn, (hw, hw))
  e = rel_l2(host(g0[i:i + 1]), gc[..., :c0].reshape(1, hw // 2, 2, hw // 2, 2, c0).sum(axis=(2, 4)))
  assert e < tol, ('g0 vs float64', e)
  j = n1 - 1
  if gsz:
    srcs = [og * gsz + j % gsz for og in range(n // gsz) if perm[og] == j // gsz]
  else:
    srcs = [j]
  tot = sum(N.conv2d_bwd_data_gemm(gyn[k:k + 1], wn, (hw, hw))[..., c0:] for k in srcs)
  e = rel_l2(host(g1[j:j + 1]), tot)
  assert e < tol, ('g1 vs float64', e, srcs)
  # one output not wanted: the other is unchanged
  g0b = torch.empty_like(g0)
  wpack = O.PackCache.get(wd, d, 1)      # held in a local: an uncached pack must outlive the launch
  O.call('tg_conv2d_upcat_bwd_data', gyd.data_ptr(), wpack.data_ptr(), g0b.data_ptr(), None, n, hw, hw,
         c0, c1, cout, gsz, O._pack_perm(perm), O._dt(gyd), O._stream())
  assert torch.equal(g0b, g0)


@pytest.mark.parametrize('hw,cin,cout,na,nb', [(16, 64, 32, 2, 3), (8, 256, 64, 3, 2), (32, 16, 16, 1, 4)])
def test_paired_filter_gradient_matches_two_launches(ops, hw, cin, cout, na, nb):
  """tg_conv2d_bwd_weight2 (two batches of one layer in one launch) == two tg_conv2d_bwd_weight launches."""
  g = torch.Generator().manual_seed(7)
  mk = lambda n, c: torch.randn(n, hw, hw, c, generator=g).to(dev()).bfloat16()
  xa, gya, xb, gyb = mk(na, cin), mk(na, cout), mk(nb, cin), mk(nb, cout)
  spec = ops.ConvSpec(3, 'SAME')
  ref = torch.zeros(3, 3, cin, cout, device=dev())
  ops.conv_bwd_weight_raw(xa, gya, spec, out=ref)
  ops.conv_bwd_weight_raw(xb, gyb, spec, out=ref)
  out = torch.zeros_like(ref)
  assert ops.conv_bwd_weight2_raw(xa, gya, xb, gyb, spec, out)
  assert rel_l2(host(out), host(ref)) < 1e-5


@pytest.mark.parametrize('dtype,hw,c1,c2', [(torch.bfloat16, 16, 32, 64), (torch.bfloat16, 32, 16, 16), (torch.float32, 8, 8, 8),
                                            (torch.bfloat16, 8, 64, 32)])
def test_lrelu_backward_folded_into_next_backward_data(ops, dtype, hw, c1, c2):
  """conv+bias+lrelu -> conv+bias+lrelu: with fuse_input_lrelu the first layer's LeakyReLU backward runs in the second
  layer's backward-data epilogue and its bias gradient in its own filter-gradient kernel (gradient sinks); all
  gradients must match the unfused chain."""
  g = torch.Generator().manual_seed(21)
  x = torch.randn(3, hw, hw, 16, generator=g).to(dev()).to(dtype)
  w1 = (torch.randn(3, 3, 16, c1, generator=g) * 0.1).to(dev())
  b1 = (torch.randn(c1, generator=g) * 0.1).to(dev())
  w2 = (torch.randn(3, 3, c1, c2, generator=g) * 0.1).to(dev())
  b2 = (torch.randn(c2, generator=g) * 0.1).to(dev())
  gy = torch.randn(3, hw, hw, c2, generator=g).to(dev()).to(dtype)
  res = []
  for fuse, sinks in ((False, False), (True, False), (True, True)):
    ops.GradSink.clear()
    ps = [t.clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
    xin = x.clone().requires_grad_(True)
    bufs = [torch.zeros_like(p) for p in ps]
    if sinks:
      for p, b in zip(ps, bufs):
        ops.GradSink.register(p, b)
    z1 = ops.conv2d(xin, ps[0], ps[1], 3, 'SAME', lrelu=True)
    z2 = ops.conv2d(z1, ps[2], ps[3], 3, 'SAME', lrelu=True, fuse_input_lrelu=fuse)
    z2.backward(gy)
    ops.GradSink.flush()
    grads = [b if sinks else p.grad for p, b in zip(ps, bufs)]
    res.append([host(xin.grad)] + [host(t) for t in grads])
  ops.GradSink.clear()
  tol = 1e-5 if dtype == torch.float32 else 2e-2      # bf16: the fused path rounds gx once instead of twice
  for other in res[1:]:
    for a, b in zip(other, res[0]):
      assert rel_l2(a, b) < tol


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_gdrop_op(ops, dtype):
  """tg_gdrop (libs/gdrop.py:20-36, mode 'prop'): x * (noise[N,1,1,C] * strength * sqrt(C) + 1) with a host or a device
  strength, a channel-padded tensor (c_logical < C), and the node differentiated twice (it is linear: its backward is
  itself)."""
  rng = np.random.RandomState(8)
  n, hw, c, cl = 3, 4, 16, 9
  x = rng.randn(n, hw, hw, c)
  x[..., cl:] = 0.0
  if dtype == torch.bfloat16:
    x = bf16_round(x)
  noise = rng.randn(n, c).astype(np.float32)
  strength = 0.37
  f = noise.astype(np.float64) * (strength * float(np.sqrt(np.float32(cl)))) + 1.0
  ref = x * f[:, None, None, :]
  xd = to_dev(x, dtype).requires_grad_(True)
  nd = torch.from_numpy(noise).to(dev())
  y = ops.gdrop(xd, strength, noise=nd, c_logical=cl)
  assert rel_l2(host(y), ref) < tol_for(dtype)
  sdev = torch.tensor([strength], dtype=torch.float32, device=dev())      # the gdrop_strength variable
  assert torch.equal(ops.gdrop(xd.detach(), sdev, noise=nd, c_logical=cl), y.detach())
  gy = rng.randn(n, hw, hw, c)
  gyd = to_dev(gy, dtype).requires_grad_(True)
  gx, = torch.autograd.grad(y, xd, gyd, create_graph=True)
  assert rel_l2(host(gx), gy * f[:, None, None, :]) < tol_for(dtype, True)
  v = rng.randn(n, hw, hw, c)
  ggy, = torch.autograd.grad(gx, gyd, to_dev(v, dtype))      # d/d gy of gy * f, contracted with v
  assert rel_l2(host(ggy), v * f[:, None, None, :]) < tol_for(dtype, True)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('inner', [(4, 4, 8), (5,), (3, 3)])      # 16-byte rows, and rows only the scalar path can take
def test_rows_and_cat_rows(ops, record_calls, dtype, inner):
  """ops.rows / ops.cat_rows (tg_rows_assemble): the batched towers' tf.concat / split glue (twingan.py:233-288).  Views
  forward; repeats and concatenations one launch; the backward ONE launch that writes each row block once -- the fp32 sum
  of every gradient covering it, zeros where none does -- against the framework's chunk / cat / add arithmetic in fp64."""
  rng = np.random.RandomState(5)
  b = 3
  x = to_dev(rng.randn(*((2 * b,) + inner)), dtype).double().cpu().numpy()      # values the dtype holds exactly
  xd = to_dev(x, dtype).requires_grad_(True)
  s_rows, t_rows = (0, b), (b, 2 * b)
  record_calls.clear()
  es, et, rep, unused, mid = ops.rows(xd, [s_rows, t_rows, (t_rows, s_rows, s_rows, t_rows), (1, 4), (2, 5)])
  assert [c[0] for c in record_calls] == ['tg_rows_assemble']
  assert es.data_ptr() == xd.data_ptr() and et.data_ptr() == xd.data_ptr() + b * xd[0].numel() * xd.element_size()
  np.testing.assert_array_equal(host(rep), np.concatenate([x[b:], x[:b], x[:b], x[b:]]))
  np.testing.assert_array_equal(host(mid), x[2:5])
  g_es, g_rep, g_mid = rng.randn(*es.shape), rng.randn(*rep.shape), rng.randn(*mid.shape)
  g_es, g_rep, g_mid = (to_dev(g, dtype).double().cpu().numpy() for g in (g_es, g_rep, g_mid))
  want = np.zeros_like(x)
  want[:b] += g_es
  want[b:] += g_rep[:b] + g_rep[3 * b:]
  want[:b] += g_rep[b:2 * b] + g_rep[2 * b:3 * b]
  want[2:5] += g_mid
  record_calls.clear()
  gx, = torch.autograd.grad([es, rep, mid], xd, [to_dev(g_es, dtype), to_dev(g_rep, dtype), to_dev(g_mid, dtype)])
  assert [c[0] for c in record_calls] == ['tg_rows_assemble']      # et and `unused` bring no gradient and no zero tensor
  assert rel_l2(host(gx), want) < (1e-6 if dtype == torch.float32 else 4e-3)
  # a lone gradient over all rows is passed through; no gradient at all is None
  whole, = ops.rows(xd, [(0, 2 * b)])
  record_calls.clear()
  g1, = torch.autograd.grad(whole, xd, to_dev(x, dtype))
  assert record_calls == [] and torch.equal(g1, to_dev(x, dtype))
  # concatenation: a copy with a tape (the gradients are views), a view of adjacent rows without one
  a, c = to_dev(x[:2], dtype).requires_grad_(True), to_dev(x[2:], dtype)
  both = ops.cat_rows([a, c, a])
  np.testing.assert_array_equal(host(both), np.concatenate([x[:2], x[2:], x[:2]]))
  ga, = torch.autograd.grad(both, a, both.detach())
  assert rel_l2(host(ga), 2 * x[:2]) < (1e-6 if dtype == torch.float32 else 4e-3)
  base = to_dev(x, dtype)
  record_calls.clear()
  view = ops.cat_rows([base[:2], base[2:]])
  assert record_calls == [] and view.data_ptr() == base.data_ptr() and torch.equal(view, base)
  apart = ops.cat_rows([base[:2], base[3:]])
  np.testing.assert_array_equal(host(apart), np.concatenate([x[:2], x[3:]]))
  # more blocks than one launch takes, more sources than one job takes
  many = ops.cat_rows([base[k:k + 1] for k in (0, 1, 2, 3, 4, 5, 0, 1, 2, 3)])
  np.testing.assert_array_equal(host(many), x[[0, 1, 2, 3, 4, 5, 0, 1, 2, 3]])
  dst = torch.empty_like(base[:1])
  ops.assemble_rows(dst, [(0, 1, [base[k:k + 1] for k in range(6)])])
  assert rel_l2(host(dst), x.sum(0, keepdims=True)) < (1e-6 if dtype == torch.float32 else 8e-3)


def test_uniform_draws(ops):
  """tg_uniform: Philox4x32-10 keyed by (seed, draw counter) -- U[0,1) values, a new draw per launch (the counter is
  advanced on the device, ticket word back at 0), the same stream for the same seed, ragged lengths, and sane moments."""
  state = torch.zeros(2, dtype=torch.int32, device=dev())
  a = ops.uniform(32, 7, state)
  b = ops.uniform(32, 7, state)
  assert state.tolist() == [2, 0]
  assert float(a.min()) >= 0.0 and float(a.max()) < 1.0 and not torch.equal(a, b)
  again = torch.zeros(2, dtype=torch.int32, device=dev())
  assert torch.equal(ops.uniform(32, 7, again), a) and torch.equal(ops.uniform(32, 7, again), b)
  assert not torch.equal(ops.uniform(32, 8, torch.zeros(2, dtype=torch.int32, device=dev())), a)
  # a prefix of a longer draw is the shorter draw (counter = block index), whatever the grid
  long = ops.uniform(300001, 7, torch.zeros(2, dtype=torch.int32, device=dev()))
  assert torch.equal(long[:32], a)
  x = host(long)
  assert abs(x.mean() - 0.5) < 3e-3 and abs(x.var() - 1.0 / 12.0) < 2e-3 and x.min() >= 0.0 and x.max() < 1.0
  assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 1e-2
  r = host(ops.uniform(1000, 3, torch.zeros(2, dtype=torch.int32, device=dev()), lo=-1.0, hi=1.0))
  assert r.min() >= -1.0 and r.max() < 1.0 and abs(r.mean()) < 0.1
  # the known-answer vector of Philox4x32-10 (Random123 kat_vectors: counter 0, key 0)
  z = ops.uniform(4, 0, torch.zeros(2, dtype=torch.int32, device=dev()))
  words = [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
  assert host(z).tolist() == [float(np.float32(w >> 8) * np.float32(1.0 / 16777216.0)) for w in words]


def test_no_backward_work_without_a_gradient(ops, record_calls):
  """A conv / minibatch-stddev node the backward reaches WITHOUT a gradient (the gradient penalty's second backward
  reaches the layers after the minibatch stddev only through LeakyReLU masks) launches nothing and passes None on --
  the default would materialise a tensor of zeros and run backward-data, filter-gradient and mbstd kernels on it."""
  class Drop(torch.autograd.Function):      # a consumer whose backward has nothing to say about its input
    @staticmethod
    def forward(ctx, t):
      return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
      return None

  rng = np.random.RandomState(2)
  x = to_dev(rng.randn(4, 4, 4, 16), torch.bfloat16).requires_grad_(True)
  w1, w2 = (to_dev(rng.randn(3, 3, c, 16) * 0.1).requires_grad_(True) for c in (16, 24))
  b = to_dev(rng.randn(16) * 0.1).requires_grad_(True)
  h = ops.conv2d(x, w1, b, 3, 'SAME', lrelu=True)
  y = ops.conv2d(ops.minibatch_state_concat(h, 24, 2), w2, None, 3, 'SAME', lrelu=False)
  record_calls.clear()
  (Drop.apply(y).float().sum() + x.float().sum()).backward()
  assert [c[0] for c in record_calls if c[0].startswith(('tg_conv2d', 'tg_mbstd', 'tg_lrelu'))] == []
  assert w1.grad is None and w2.grad is None and b.grad is None
  assert torch.equal(x.grad, torch.ones_like(x))


def test_first_order_only_input(ops, record_calls):
  """ops.first_order_only: the gradient penalty differentiates D with respect to the interpolates under create_graph;
  the FINAL backward then wants parameter gradients only -- the first layer skips its input gradient (a full-resolution
  backward-data launch and a copy into .grad), the parameter gradients are those of the unmarked run."""
  rng = np.random.RandomState(3)
  xv = to_dev(rng.randn(2, 8, 8, 3), torch.bfloat16)
  wv, bv = to_dev(rng.randn(1, 1, 3, 16) * 0.5), to_dev(rng.randn(16) * 0.1)
  res = []
  for mark in (False, True):
    x = xv.clone().requires_grad_(True)
    if mark:
      ops.first_order_only(x)
    w, b = wv.clone().requires_grad_(True), bv.clone().requires_grad_(True)
    with ops.second_order():
      y = ops.pointwise_conv(ops.scale(x, 0.7), w, b, lrelu=True)
    gx, = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=True)      # first order: wanted either way
    record_calls.clear()
    (gx.float().pow(2).sum() + y.float().sum()).backward()
    res.append((gx.detach(), w.grad, b.grad, x.grad, [c[0] for c in record_calls]))
  (gx0, w0, b0, x0, calls0), (gx1, w1, b1, x1, calls1) = res
  assert torch.equal(gx0, gx1) and torch.equal(w0, w1) and torch.equal(b0, b1)
  assert x0 is not None and x1 is None
  assert calls0.count('tg_pointwise_conv_fwd') == calls1.count('tg_pointwise_conv_fwd') + 1, (calls0, calls1)
  assert calls0.count('tg_axpby') == calls1.count('tg_axpby') + 1, (calls0, calls1)


SMALL_MASK_CASES = [
    # n, hw, cin, cout, k, padding, kernel family of the backward-data
    (5, 8, 256, 256, 3, 'SAME', 'conv_img'),
    (3, 8, 512, 256, 3, 'SAME', 'conv_img'),
    (5, 4, 256, 256, 3, 'SAME', 'conv_small'),
    (6, 4, 264, 256, 3, 'SAME', 'conv_small'),      # the minibatch-stddev layer: 264 input channels (a partial last block)
    (6, 4, 64, 64, 4, 'VALID', 'conv_small'),       # the dense rewrite of the discriminator's 4x4 VALID conv
]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('n,hw,cin,cout,k,padding,family', SMALL_MASK_CASES)
def test_masked_backward_data_on_the_small_maps(ops, dtype, n, hw, cin, cout, k, padding, family):
  """tg_conv2d_bwd_data_masked at the 8x8 / 4x4 maps and the dense 4x4-VALID layer: the LeakyReLU backward of the layer
  below rides in the epilogue of conv_img / conv_small (as it does in conv_tile's from 16x16 up) -- ONE launch, no
  tg_lrelu_bwd pass over the result -- against the float64 oracle on rounded operands (one rounding of the product)."""
  from twingan_amd import _lib
  rng = np.random.RandomState(31)
  rnd = bf16_round if dtype == torch.bfloat16 else f16_round
  x = rnd(rng.randn(n, hw, hw, cin))
  wt = rnd(rng.randn(k, k, cin, cout) / np.sqrt(k * k * cin))
  spec = ops.ConvSpec(k, padding)
  ho, wo = spec.out_hw(hw, hw)
  gy = rnd(rng.randn(n, ho, wo, cout))
  xd, gyd, wd = to_dev(x, dtype), to_dev(gy, dtype), to_dev(wt)
  names = []
  import twingan_amd.ops as O
  real = O.call

  def spy(name, *a, **kw):
    names.append(name)
    return real(name, *a, **kw)
  O.call = spy
  try:
    gxm = ops.conv_bwd_data_masked_raw(gyd, wd, xd, spec)
  finally:
    O.call = real
  kern = _lib.load().tg_last_kernel().decode()
  assert names[-1] == 'tg_conv2d_bwd_data_masked' and 'tg_lrelu_bwd' not in names and family in kern, (names, kern)      # the conv kernel ran LAST: no mask pass after it
  ref = N.conv2d_bwd_data(gy, wt, (hw, hw), padding) * np.where(x > 0, 1.0, 0.2)
  assert rel_l2(host(gxm), ref) < (8e-4 if dtype == torch.float16 else 5e-3)
  plain = ops.conv_bwd_data_raw(gyd, wd, (n, hw, hw, cin), spec)      # the same kernel without the mask: untouched
  assert rel_l2(host(plain), N.conv2d_bwd_data(gy, wt, (hw, hw), padding)) < (8e-4 if dtype == torch.float16 else 5e-3)
  # ... and the forward conv with the mask of its OUTPUT's shape (tg_conv2d_fwd_masked: the gradient penalty's second pass)
  msrc = rnd(rng.randn(n, ho, wo, cout))
  ym = ops.conv_fwd_masked_raw(xd, wd, to_dev(msrc, dtype), spec)
  assert family in _lib.load().tg_last_kernel().decode()
  assert rel_l2(host(ym), N.conv2d(x, wt, padding) * np.where(msrc > 0, 1.0, 0.2)) < (8e-4 if dtype == torch.float16 else 5e-3)


@pytest.mark.parametrize('dtype,hw,c1,c2', [(torch.bfloat16, 16, 32, 64), (torch.float32, 8, 8, 8), (torch.bfloat16, 32, 16, 16)])
def test_lrelu_fold_under_create_graph_matches_unfused(ops, dtype, hw, c1, c2):
  """Gradient-penalty shaped double backward through conv+bias+lrelu -> conv+bias+lrelu -> conv: with fuse_input_lrelu
  the first backward runs MaskedDgradFn (mask in the backward-data epilogue) instead of LeakyReLU-backward + backward-
  data nodes (and the pooled last layer one unpool+mask node); the input gradient, the penalty and every parameter
  gradient of the penalty must match the unfused chain and a float64 torch reference."""
  g = torch.Generator().manual_seed(33)
  x = torch.randn(2, hw, hw, 16, generator=g).to(dev()).to(dtype)
  shapes = [(3, 3, 16, c1), (c1,), (3, 3, c1, c2), (c2,), (3, 3, c2, 16), (16,)]
  ws = [(torch.randn(*s, generator=g) * (0.1 if len(s) == 4 else 0.05)).to(dev()) for s in shapes]
  res = []
  for fuse in (False, True):
    ops.GradSink.clear()
    ps = [t.clone().requires_grad_(True) for t in ws]
    xin = x.clone().requires_grad_(True)
    z1 = ops.conv2d(xin, ps[0], ps[1], 3, 'SAME', lrelu=True)
    z2 = ops.conv2d(z1, ps[2], ps[3], 3, 'SAME', lrelu=True, fuse_input_lrelu=fuse)
    _, z3 = ops.conv2d(z2, ps[4], ps[5], 3, 'SAME', lrelu=True, fuse_input_lrelu=fuse, pool=True)      # block end: pooled
    gx, = torch.autograd.grad(z3.float().sum(), xin, create_graph=True)
    pen = ((gx.float().pow(2).sum(dim=(1, 2, 3)).sqrt() - 1.0) ** 2).mean()
    grads = torch.autograd.grad(pen, [ps[0], ps[2], ps[4]])
    res.append([host(gx.detach()), host(pen.detach().reshape(1))] + [host(t) for t in grads])
  tol = 1e-5 if dtype == torch.float32 else 3e-2
  for a, b in zip(res[1], res[0]):
    assert rel_l2(a, b) < tol, (rel_l2(a, b))
  # independent reference: the same graph in float64 torch on the host
  import torch.nn.functional as F
  xr = x.double().cpu().requires_grad_(True)
  pr = [t.double().cpu().requires_grad_(True) for t in ws]
  def layer(a, w, b):
    y = F.conv2d(a.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), b, padding=1).permute(0, 2, 3, 1)
    return torch.maximum(y, 0.2 * y)
  a3 = layer(layer(layer(xr, pr[0], pr[1]), pr[2], pr[3]), pr[4], pr[5])
  z3r = F.avg_pool2d(a3.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
  gxr, = torch.autograd.grad(z3r.sum(), xr, create_graph=True)
  penr = ((gxr.pow(2).sum(dim=(1, 2, 3)).sqrt() - 1.0) ** 2).mean()
  gr = torch.autograd.grad(penr, [pr[0], pr[2], pr[4]])
  want = [gxr.detach().numpy(), penr.detach().reshape(1).numpy()] + [t.numpy() for t in gr]
  rtol = 1e-4 if dtype == torch.float32 else 6e-2
  for a, b in zip(res[1], want):
    assert rel_l2(a, b) < rtol, rel_l2(a, b)


@pytest.mark.parametrize('dtype,hw,c1,c2', [(torch.bfloat16, 16, 32, 64), (torch.float32, 8, 8, 8), (torch.bfloat16, 32, 16, 16),
                                            (torch.float16, 64, 16, 32)])
def test_gradient_penalty_second_pass_masks_in_the_conv_epilogue(ops, monkeypatch, dtype, hw, c1, c2):
  """The trainer's gradient-penalty flow (image_generation.py:414-439): inner gradient under ops.no_param_grads with
  create_graph, parameters through the double backward.  With USE_GP_PREMASK the LeakyReLU mask every node of the second
  pass applies to its incoming cotangent sits in the epilogue of the conv that produces it (tg_conv2d_fwd_masked; the
  producer is flagged and skips its tg_lrelu_bwd launch): fewer launches, same penalty gradients as the unflagged chain and
  as a float64 torch reference; the launch counts are checked through the profiler hook."""
  from twingan_amd import _lib
  g = torch.Generator().manual_seed(34)
  x = torch.randn(2, hw, hw, 16, generator=g).to(dev()).to(dtype)
  shapes = [(3, 3, 16, c1), (c1,), (3, 3, c1, c2), (c2,), (3, 3, c2, c2), (c2,), (3, 3, c2, 16), (16,)]
  ws = [(torch.randn(*s, generator=g) * (0.1 if len(s) == 4 else 0.05)).to(dev()) for s in shapes]
  res, launches = [], []
  # third run: the block end's unpool + mask + masked backward-data as ONE differentiable node (UnpoolMaskedDgradFn)
  for premask, one_node in ((False, False), (True, False), (True, True)):
    monkeypatch.setattr(ops, 'USE_GP_PREMASK', premask)
    monkeypatch.setattr(ops, 'USE_DGRAD_UNPOOL_GP', one_node)
    ops.GradSink.clear()
    ps = [t.clone().requires_grad_(True) for t in ws]
    xin = x.clone().requires_grad_(True)
    with ops.second_order():
      z1 = ops.conv2d(xin, ps[0], ps[1], 3, 'SAME', lrelu=True)
      z2 = ops.conv2d(z1, ps[2], ps[3], 3, 'SAME', lrelu=True, fuse_input_lrelu=True)
      _, z3 = ops.conv2d(z2, ps[4], ps[5], 3, 'SAME', lrelu=True, fuse_input_lrelu=True, pool=True, pool_only=True)      # block end
      z4 = ops.conv2d(z3, ps[6], ps[7], 3, 'SAME', lrelu=True, fuse_input_lrelu=True)
    with ops.no_param_grads():
      gx, = torch.autograd.grad(z4.float().sum(), xin, create_graph=True)
    pen = ((gx.float().pow(2).sum(dim=(1, 2, 3)).sqrt() - 1.0) ** 2).mean()
    _lib.profiler = []
    try:
      grads = torch.autograd.grad(pen, [ps[0], ps[2], ps[4], ps[6]])
      torch.cuda.synchronize()
      names = [r[0] for r in _lib.profiler]
    finally:
      _lib.profiler = None
    launches.append((names.count('tg_lrelu_bwd'), names.count('tg_conv2d_fwd_masked')))
    res.append([host(gx.detach()), host(pen.detach().reshape(1))] + [host(t) for t in grads])
    node = gx.grad_fn
    seen = set()
    stack = [node]
    while stack:      # is the one-node form in the graph of the inner gradient exactly when it should be?
      nd = stack.pop()
      if nd is None or nd in seen:
        continue
      seen.add(nd)
      stack.extend(f for f, _ in nd.next_functions)
    has = any('UnpoolMaskedDgradFn' in type(nd).__name__ for nd in seen)
    assert has == (one_node and dtype != torch.float32 and c2 % 32 == 0), (has, one_node)
  # three masks move into conv epilogues: z1's (into conv 2's node), z2's (conv 3's node), z3-block-end's (the unpool node)
  assert launches[1][1] >= 2 and launches[1][0] <= launches[0][0] - launches[1][1], launches
  tol = 1e-5 if dtype == torch.float32 else 3e-2
  for a, b in zip(res[1], res[0]):
    assert rel_l2(a, b) < tol, (rel_l2(a, b))
  for a, b in zip(res[2], res[0]):
    assert rel_l2(a, b) < tol, (rel_l2(a, b))
  import torch.nn.functional as F
  xr = x.double().cpu().requires_grad_(True)
  pr = [t.double().cpu().requires_grad_(True) for t in ws]
  def layer(a, w, b):
    y = F.conv2d(a.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), b, padding=1).permute(0, 2, 3, 1)
    return torch.maximum(y, 0.2 * y)
  a3 = layer(layer(layer(xr, pr[0], pr[1]), pr[2], pr[3]), pr[4], pr[5])
  z3r = F.avg_pool2d(a3.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
  z4r = layer(z3r, pr[6], pr[7])
  gxr, = torch.autograd.grad(z4r.sum(), xr, create_graph=True)
  penr = ((gxr.pow(2).sum(dim=(1, 2, 3)).sqrt() - 1.0) ** 2).mean()
  gr = torch.autograd.grad(penr, [pr[0], pr[2], pr[4], pr[6]])
  want = [gxr.detach().numpy(), penr.detach().reshape(1).numpy()] + [t.numpy() for t in gr]
  rtol = 1e-4 if dtype == torch.float32 else 1e-1      # four 16-bit layers (the three-layer test above: 6e-2; measured 7.4e-2)
  for r in (res[1], res[2]):
    for a, b in zip(r, want):
      assert rel_l2(a, b) < rtol, rel_l2(a, b)


@pytest.mark.parametrize('k,cin,cout', [(3, 16, 32), (1, 3, 16), (4, 64, 64), (3, 264, 256), (3, 5, 7)])
def test_spectral_norm_matches_oracle(ops, k, cin, cout):
  """tg_spectral_norm_fwd / _bwd (libs/sn.py:38-101): w_bar, the next power-iteration vector, and d L / d w with the
  gradient flowing through sigma, v and u' -- against float64 autograd of the literal formulas."""
  rng = np.random.RandomState(6)
  w = rng.randn(k, k, cin, cout) * 0.1
  u = rng.randn(1, cout)
  gq = rng.randn(k, k, cin, cout)
  wd = to_dev(w).requires_grad_(True)
  w_bar, u_new = ops.spectral_norm(wd, to_dev(u))
  (w_bar * to_dev(gq)).sum().backward()
  wt = torch.from_numpy(w).requires_grad_(True)
  w2 = wt.reshape(-1, cout)
  ut = torch.from_numpy(u)
  v = R.l2_normalize(ut @ w2.t())
  u1 = R.l2_normalize(v @ w2)
  sigma = (v @ w2 @ u1.t()).reshape(())
  ref = (w2 / sigma).reshape(wt.shape)
  (ref * torch.from_numpy(gq)).sum().backward()
  assert rel_l2(host(w_bar), ref.detach().numpy()) < F32_TOL
  assert rel_l2(host(u_new), u1.detach().numpy()) < F32_TOL
  assert rel_l2(host(wd.grad), wt.grad.numpy()) < 5 * F32_TOL


def test_spectral_norm_of_many_kernels_in_three_launches(ops):
  """tg_spectral_norm_fwd_multi (ops.spectral_norm_multi, what pggan.prepare_run uses): the power iterations of several kernels of
  different shapes from one job table -- w_bar, u', and the gradients through the per-kernel nodes equal the one-kernel
  entry point's bit for bit (the same kernel bodies in the same order), also when the table is reused for a second run."""
  g = torch.Generator().manual_seed(37)
  shapes = [(3, 3, 16, 32), (1, 1, 3, 16), (4, 4, 64, 64), (3, 3, 264, 256), (3, 3, 5, 7), (3, 3, 128, 40)]
  ws = [(torch.randn(*sh, generator=g) * 0.1).to(dev()) for sh in shapes]
  us = [torch.randn(1, sh[3], generator=g).to(dev()) for sh in shapes]
  gq = [torch.randn(*sh, generator=g).to(dev()) for sh in shapes]
  outs = [torch.empty(w.numel(), dtype=torch.float32, device=w.device) for w in ws]
  table = None
  for run in range(2):
    single = []
    for w, u, q in zip(ws, us, gq):
      wd = w.clone().requires_grad_(True)
      wb, un = ops.spectral_norm(wd, u)
      (wb * q).sum().backward()
      single.append((wb.detach().clone(), un.detach().clone(), wd.grad.clone()))
    wds = [w.clone().requires_grad_(True) for w in ws]
    if run == 1:      # the same addresses: the table of run 0 is reused
      for wd, keep in zip(wds, kept):
        keep.data.copy_(wd.data)
      wds = kept
      for wd in wds:
        wd.grad = None
    res, table2 = ops.spectral_norm_multi(list(zip(wds, us, outs)), table)
    assert run == 0 or table2 is table
    table, kept = table2, wds
    loss = sum((wb * q).sum() for (wb, _), q in zip(res, gq))
    loss.backward()
    for (wb, un), wd, (wb1, un1, g1) in zip(res, wds, single):
      assert torch.equal(wb.detach(), wb1) and torch.equal(un.detach().reshape(-1), un1.reshape(-1))
      assert torch.equal(wd.grad, g1)
    for u, (_, un) in zip(us, res):      # the next run starts from u', assigned in place as pggan.end_run does
      u.copy_(un.detach().reshape(1, -1))
  # the nodes save the table's PERSISTENT buffers: a backward after the table has run again must refuse, not use them
  for wd in wds:
    wd.grad = None
  res, _ = ops.spectral_norm_multi(list(zip(wds, us, outs)), table)
  stale = sum((wb * q).sum() for (wb, _), q in zip(res, gq))
  ops.spectral_norm_multi(list(zip(wds, us, outs)), table)
  with pytest.raises(RuntimeError, match='one outstanding graph'):
    stale.backward()


BGEMM_CASES = [
    # batch, m, n, k     (attention: s = f g^T is (N, N, c/8); o = beta h is (N, c, N); their gradients transpose them)
    (2, 64, 64, 2), (2, 64, 16, 64), (3, 256, 256, 8), (2, 256, 64, 256), (1, 130, 70, 36), (2, 33, 9, 5),
    (2, 1024, 1024, 8), (2, 1024, 64, 1024),
]


@pytest.mark.parametrize('batch,m,n,k', BGEMM_CASES)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_batched_gemm_all_transposes(ops, batch, m, n, k, dtype):
  """tg_batched_gemm (csrc/attention.hip): all four operand layouts of the MFMA kernel (ds_read_b128 for a K-contiguous
  operand, the LDS transpose read for the other kind), ragged tiles and unaligned leading dimensions, against float64."""
  rng = np.random.RandomState(11)
  for ta in (False, True):
    for tb in (False, True):
      a = rng.randn(batch, k, m) if ta else rng.randn(batch, m, k)
      b = rng.randn(batch, n, k) if tb else rng.randn(batch, k, n)
      if dtype == torch.bfloat16:
        a, b = bf16_round(a), bf16_round(b)
      ref = np.matmul(a.transpose(0, 2, 1) if ta else a, b.transpose(0, 2, 1) if tb else b) * 0.5
      c = ops.bgemm(to_dev(a, dtype), to_dev(b, dtype), ta, tb, 0.5)
      assert rel_l2(host(c), ref) < (F32_TOL if dtype == torch.float32 else 4e-3), (ta, tb)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('rows,cols', [(6, 64), (10, 100), (4, 4096)])
def test_softmax_rows_first_and_second_order(ops, rows, cols, dtype):
  """Row softmax, its backward and the backward OF that backward (the gradient-penalty pass differentiates the
  discriminator's attention twice) against float64 autograd."""
  rng = np.random.RandomState(12)
  s = rng.randn(rows, cols) * 2.0
  w1 = rng.randn(rows, cols)
  w2 = rng.randn(rows, cols)
  if dtype == torch.bfloat16:
    s, w1, w2 = bf16_round(s), bf16_round(w1), bf16_round(w2)
  sd = to_dev(s, dtype).requires_grad_(True)
  p = ops.softmax_rows(sd)
  gs, = torch.autograd.grad(p, sd, grad_outputs=to_dev(w1, dtype), create_graph=True)
  (gs * to_dev(w2, dtype)).sum().backward()
  st = torch.from_numpy(s).requires_grad_(True)
  pt = torch.softmax(st, dim=-1)
  gst, = torch.autograd.grad(pt, st, grad_outputs=torch.from_numpy(w1), create_graph=True)
  (gst * torch.from_numpy(w2)).sum().backward()
  tol = F32_TOL if dtype == torch.float32 else 2e-2
  assert rel_l2(host(p), pt.detach().numpy()) < (F32_TOL if dtype == torch.float32 else 4e-3)
  assert rel_l2(host(gs), gst.detach().numpy()) < tol
  assert rel_l2(host(sd.grad), st.grad.numpy()) < (5 * F32_TOL if dtype == torch.float32 else 5e-2)


def test_tanh_and_scale_second_order(ops):
  rng = np.random.RandomState(13)
  x, w1, w2 = rng.randn(4, 33), rng.randn(4, 33), rng.randn(4, 33)
  gam = np.array([0.7])
  xd = to_dev(x).requires_grad_(True)
  gd = to_dev(gam).requires_grad_(True)
  y = ops.scale_dev(ops.tanh(xd), gd)
  gx, = torch.autograd.grad(y, xd, grad_outputs=to_dev(w1), create_graph=True)
  (gx * to_dev(w2)).sum().backward()
  xt = torch.from_numpy(x).requires_grad_(True)
  gt = torch.from_numpy(gam).requires_grad_(True)
  yt = torch.tanh(xt) * gt
  gxt, = torch.autograd.grad(yt, xt, grad_outputs=torch.from_numpy(w1), create_graph=True)
  (gxt * torch.from_numpy(w2)).sum().backward()
  assert rel_l2(host(y), yt.detach().numpy()) < F32_TOL and rel_l2(host(gx), gxt.detach().numpy()) < F32_TOL
  assert rel_l2(host(xd.grad), xt.grad.numpy()) < 5 * F32_TOL and rel_l2(host(gd.grad), gt.grad.numpy()) < 5 * F32_TOL


@pytest.mark.parametrize('hw,c,n', [(64, 64, 4), (16, 32, 3)])
def test_self_attention_layer_bf16_at_config4_shape(hw, c, n):
  """The whole SAGAN layer as the discriminator of BASELINE configs[4] runs it (64x64 map, 64 channels: N = 4096,
  d = 8) on the MFMA batched GEMM, bf16, forward and first-order backward, against the float64 oracle on the same
  bf16-rounded inputs and weights."""
  from oracle import torch_ref as R2
  from twingan_amd import Config, pggan
  from twingan_amd.params import ParamStore
  rng = np.random.RandomState(14)
  sc = 'discriminator_s/self_attention_%dx%dx%d' % (hw, hw, c)
  st = ParamStore('cuda:0')
  for nm, co in (('sa_f', c // 8), ('sa_g', c // 8), ('sa_h', c)):
    st.add_conv('%s/%s' % (sc, nm), 1, c, co, 'd', True, ())
  st.add(sc + '/sa_gamma', (1,), 'd', 'beta')
  st.build(0)
  vals = {}
  for k, spec in st.specs.items():
    v = rng.randn(*spec['shape']) * (0.6 if k.endswith('weights') else 0.1)
    vals[k] = bf16_round(v) if k.endswith('weights') else np.float32(v).astype(np.float64)
  vals[sc + '/sa_gamma'] = np.array([0.8])
  st.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})
  x = bf16_round(rng.randn(n, hw, hw, c) * 0.5)
  cfg = Config(hw=hw, max_ch=c, precision='bf16', do_self_attention=True, self_attention_hw=hw)
  for p_ in st.P.values():
    p_.requires_grad_(True)
  xd = to_dev(x, torch.bfloat16).requires_grad_(True)
  y = pggan.self_attention_layer(st.P, sc, xd, None, cfg, True)
  gy = bf16_round(rng.randn(*x.shape))
  y.backward(to_dev(gy, torch.bfloat16))
  Pt = {k: torch.from_numpy(v).requires_grad_(True) for k, v in vals.items()}
  xt = torch.from_numpy(x).requires_grad_(True)
  rcfg = R2.Config(hw=hw, max_ch=c, do_self_attention=True, self_attention_hw=hw)
  yt = R2.self_attention(Pt, sc, xt, None, rcfg, True)
  yt.backward(torch.from_numpy(gy))
  assert rel_l2(host(y), yt.detach().numpy()) < 1e-2
  assert rel_l2(host(xd.grad), xt.grad.numpy()) < 3e-2
  gd = st.grad_dict()
  for k in vals:
    ref = Pt[k].grad.numpy()
    assert rel_l2(gd[k].double().cpu().numpy(), ref) < 3e-2, k


# ------------------------------------------------------------------------- conv with the statistics epilogue
STATS_CASES = [
    # n, hw, cin, cout  -> kernel family
    (4, 256, 16, 16),     # weight-resident thin kernel, several tiles per workgroup
    (4, 256, 16, 32),
    (8, 128, 32, 64),     # weight-resident, 64-channel blocks
    (3, 64, 64, 64),      # tile kernel, two sub-tiles per wave
    (2, 32, 128, 256),    # tile kernel, channel blocks along grid y
    (3, 16, 256, 256),    # tile kernel, 2 tiles per image
    (2, 16, 40, 24),      # ragged channel counts (partial 32-channel blocks; no pixel norm at c = 24)
    (5, 8, 256, 256),     # conv_img: a whole 8x8 image per workgroup, ONE chunk per image
    (3, 8, 512, 256),
    (6, 4, 256, 256),     # conv_small: a 4x4 image is 16 lanes of a column block, ONE chunk per image
    (64, 4, 256, 64),     # ... two column blocks per workgroup
    (200, 4, 64, 32),     # ... four
]


@pytest.mark.parametrize('n,hw,cin,cout', STATS_CASES)
def test_conv_statistics_epilogue(ops, n, hw, cin, cout):
  """tg_conv2d_fwd_stats: the tensor is bit-identical to tg_conv2d_fwd's, and the per-workgroup partial sums add up
  to the sums of THAT tensor (value and square, per image and channel) -- fp32 summation error only.  Then the
  normaliser fed by those partials (tg_norm_act_fwd_conv_stats) equals the one that reads the tensor itself."""
  g = torch.Generator().manual_seed(5)
  x = (torch.randn(n, hw, hw, cin, generator=g) + 0.3).to(dev()).bfloat16()
  w = (torch.randn(3, 3, cin, cout, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev())
  spec = ops.ConvSpec(3, 'SAME')
  y_ref = ops.conv_fwd_raw(x, w, None, spec, 0)
  y, st = ops.conv_fwd_stats_raw(x, w, spec)
  assert st is not None, 'this shape should dispatch a kernel with the statistics epilogue'
  assert torch.equal(y, y_ref)
  part = st.part.view(n, st.chunks, 2, cout).double().sum(dim=1).cpu().numpy()
  yd = y.double().cpu().numpy().reshape(n, hw * hw, cout)
  s1, s2 = yd.sum(axis=1), (yd * yd).sum(axis=1)
  assert np.abs(part[:, 0] - s1).max() <= 2e-6 * np.sqrt(s2 * hw * hw).max()
  assert np.abs(part[:, 1] / s2 - 1).max() <= 2e-6
  gamma = (1 + 0.1 * torch.randn(cout, generator=g)).to(dev())
  beta = (0.1 * torch.randn(cout, generator=g)).to(dev())
  pn = cout % 8 == 0 and (cout // 8) & (cout // 8 - 1) == 0      # the pixel-norm kernel wants c = 8 * 2^k
  z_ref = ops.norm_act(y, gamma, beta, pixel_norm=pn)
  z = ops.norm_act(y, gamma, beta, pixel_norm=pn, conv_stats=st)
  assert rel_l2(host(z), host(z_ref)) < 2e-3      # one bf16 rounding of the output is 1.1e-3; statistics differ by ~1e-6
  zp_ref = ops.norm_act(y, gamma, beta, pixel_norm=pn, pool=True)
  zp = ops.norm_act(y, gamma, beta, pixel_norm=pn, pool=True, conv_stats=st)
  assert rel_l2(host(zp[1]), host(zp_ref[1])) < 2e-3


def test_upcat_conv_statistics_epilogue(ops):
  g = torch.Generator().manual_seed(6)
  n, h, c0, c1, cout = 4, 32, 32, 32, 16
  x0 = torch.randn(n, h, h, c0, generator=g).to(dev()).bfloat16()
  x1 = torch.randn(2, 2 * h, 2 * h, c1, generator=g).to(dev()).bfloat16()
  w = (torch.randn(3, 3, c0 + c1, cout, generator=g) * (2.0 / (9 * (c0 + c1))) ** 0.5).to(dev())
  y_ref = ops.upcat_conv(x0, x1, w, 1, (1, 0, 0, 1))
  y, st = ops.upcat_conv_stats(x0, x1, w, 1, (1, 0, 0, 1))
  assert st is not None and torch.equal(y, y_ref)
  part = st.part.view(n, st.chunks, 2, cout).double().sum(dim=1).cpu().numpy()
  yd = y.double().cpu().numpy().reshape(n, 4 * h * h, cout)
  s2 = (yd * yd).sum(axis=1)
  assert np.abs(part[:, 0] - yd.sum(axis=1)).max() <= 2e-6 * np.sqrt(s2 * 4 * h * h).max()
  assert np.abs(part[:, 1] / s2 - 1).max() <= 2e-6


# ------------------------------------------------------------------------- conv that also writes its 2x2 average pool
@pytest.mark.parametrize('n,hw,cin,cout', [(4, 256, 16, 32), (5, 128, 32, 64), (3, 64, 64, 128), (2, 32, 128, 256),
                                           (3, 16, 256, 256), (2, 16, 40, 24)])
def test_conv_with_pooled_output(ops, monkeypatch, n, hw, cin, cout):
  """tg_conv2d_fwd_pool (last conv of a discriminator block + the avg_pool after it): z is bit-identical to
  tg_conv2d_fwd's, the pooled tensor equals tg_pool2x2_fwd of z (the 4 rounded values are added in another order: at
  most an ulp of bf16 on rare elements)."""
  from twingan_amd._lib import TG_EPI_BIAS, TG_EPI_LRELU
  g = torch.Generator().manual_seed(9)
  x = torch.randn(n, hw, hw, cin, generator=g).to(dev()).bfloat16()
  w = (torch.randn(3, 3, cin, cout, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev())
  b = (0.1 * torch.randn(cout, generator=g)).to(dev())
  spec = ops.ConvSpec(3, 'SAME')
  epi = TG_EPI_BIAS | TG_EPI_LRELU
  monkeypatch.setattr(ops, 'USE_CONV_POOL', False)
  z_ref, zp_ref = ops.conv_fwd_pool_raw(x, w, b, spec, epi)
  monkeypatch.setattr(ops, 'USE_CONV_POOL', True)
  z, zp = ops.conv_fwd_pool_raw(x, w, b, spec, epi)
  from twingan_amd import _lib
  assert 'pool' in _lib.load().tg_last_kernel().decode()
  assert torch.equal(z, z_ref)
  assert rel_l2(host(zp), host(zp_ref)) < 1e-4
  assert float((zp.float() - zp_ref.float()).abs().max()) <= 2.0 ** -7 * float(zp_ref.float().abs().max())


def test_rccl_allreduce_wrapper_single_rank():
  """tg_comm_* / tg_allreduce (the C-ABI RCCL wrapper for callers without torch.distributed): a one-rank communicator on
  this GPU -- all a 1-GPU box allows -- initialises, sums in place (= identity for one rank) on a side stream, for fp32
  and bf16, and tears down.  RCCL is bound lazily: the library has no link dependency on it."""
  import ctypes
  from twingan_amd import _lib
  lib = _lib.load()
  nbytes = lib.tg_comm_unique_id_bytes()
  assert nbytes == 128
  uid = ctypes.create_string_buffer(nbytes)
  rc = lib.tg_comm_unique_id(ctypes.cast(uid, ctypes.c_void_p))
  assert rc == 0, lib.tg_last_error().decode()
  comm = ctypes.c_void_p()
  with torch.cuda.device(0):
    rc = lib.tg_comm_init(ctypes.cast(uid, ctypes.c_void_p), 1, 0, ctypes.byref(comm))
    assert rc == 0 and comm.value, lib.tg_last_error().decode()
    st = torch.cuda.Stream()
    for dt, code in ((torch.float32, _lib.TG_F32), (torch.bfloat16, _lib.TG_BF16)):
      x = torch.randn(1 << 20, device='cuda:0').to(dt)
      want = x.clone()
      st.wait_stream(torch.cuda.current_stream())
      rc = lib.tg_allreduce(comm, x.data_ptr(), x.numel(), code, st.cuda_stream)
      assert rc == 0, lib.tg_last_error().decode()
      st.synchronize()
      assert torch.equal(x, want)
    assert lib.tg_allreduce(comm, 0, 16, _lib.TG_F32, 0) != 0 and b'bad arguments' in lib.tg_last_error()
    assert lib.tg_comm_destroy(comm) == 0


# ------------------------------------------------------------------------- fp16 storage (TG_F16)
def f16_round(a):
  return np.asarray(a, np.float32).astype(np.float16).astype(np.float64)


F16_CASES = [
    # n, hw, cin, cout, k, padding   -> kernel family
    (4, 128, 16, 32, 3, 'SAME'),      # weight-resident thin kernel; thin (16x16x32 MFMA) filter gradient
    (4, 64, 32, 64, 3, 'SAME'),       # weight-resident, 64-channel blocks
    (3, 32, 64, 128, 3, 'SAME'),      # tile kernel
    (2, 16, 256, 256, 3, 'SAME'),     # tile kernel, two tiles per image
    (5, 8, 256, 256, 3, 'SAME'),      # small-map kernel
    (5, 4, 264, 256, 3, 'SAME'),
    (6, 4, 64, 64, 4, 'VALID'),       # dense rewrite
    (1, 20, 24, 40, 3, 'SAME'),       # (20 x 12 map) first-generation fallback kernels
    (2, 12, 32, 48, 1, 'SAME'),
]


@pytest.mark.parametrize('n,hw,cin,cout,k,padding', F16_CASES)
def test_fp16_conv_kernels_vs_oracle(ops, n, hw, cin, cout, k, padding):
  """TG_F16 through the same MFMA kernels as bf16 (v_mfma_f32_*_f16 instead of *_bf16): forward with bias + LeakyReLU,
  backward-data plain and with the producer's mask, filter gradient with and without the bias gradient, against the
  float64 oracle on fp16-rounded operands.  fp16 keeps 11 significand bits: outputs within 6e-4 (one rounding is
  2.4e-4), fp32-accumulated filter / bias gradients within 1e-4."""
  import twingan_amd.ops as O
  from twingan_amd import _lib
  from twingan_amd._lib import TG_EPI_BIAS, TG_EPI_LRELU
  rng = np.random.RandomState(4)
  h = w = hw
  if (n, hw, cin) == (1, 20, 24):
    w = 12
  x = f16_round(rng.randn(n, h, w, cin))
  wt = f16_round(rng.randn(k, k, cin, cout) / np.sqrt(k * k * cin))
  b = rng.randn(cout) * 0.1
  spec = O.ConvSpec(k, padding)
  ho, wo = spec.out_hw(h, w)
  gy = f16_round(rng.randn(n, ho, wo, cout))
  xd, gyd = to_dev(x, torch.float16), to_dev(gy, torch.float16)
  wd, bd = to_dev(wt), to_dev(b)
  assert O._mfma_ok(torch.float16, cin, cout, spec, h, w)
  y = O.conv_fwd_raw(xd, wd, bd, spec, TG_EPI_BIAS | TG_EPI_LRELU)
  kern = _lib.load().tg_last_kernel().decode()
  assert 'f16' in kern, kern
  assert rel_l2(host(y), N.leaky_relu(N.conv2d(x, wt, padding) + b)) < 6e-4
  gx = O.conv_bwd_data_raw(gyd, wd, (n, h, w, cin), spec)
  ref_gx = N.conv2d_bwd_data(gy, wt, (h, w), padding)
  assert rel_l2(host(gx), ref_gx) < 6e-4
  if k == 3 and padding == 'SAME':
    gxm = O.conv_bwd_data_masked_raw(gyd, wd, xd, spec)
    assert rel_l2(host(gxm), ref_gx * np.where(x > 0, 1.0, 0.2)) < 8e-4
  gw = O.conv_bwd_weight_raw(xd, gyd, spec)
  assert rel_l2(host(gw), N.conv2d_bwd_weight(x, gy, (k, k), padding)) < 1e-4
  if k == 3 and padding == 'SAME' and hw >= 16 and h == w:      # layers whose filter-gradient kernel also sums gy (bias gradient)
    gb = torch.zeros(cout, device=dev())
    gw2 = O.conv_bwd_weight_raw(xd, gyd, spec, gbias=gb)
    assert rel_l2(host(gw2), host(gw)) < 1e-6 and rel_l2(host(gb), gy.sum(axis=(0, 1, 2))) < 1e-4
  # the bf16 instantiation is untouched by the element format of the previous calls
  yb = O.conv_fwd_raw(xd.bfloat16(), wd, bd, spec, TG_EPI_BIAS | TG_EPI_LRELU)
  assert 'f16' not in _lib.load().tg_last_kernel().decode()
  assert rel_l2(host(yb), host(y)) < 6e-3


@pytest.mark.parametrize('n,hw,c', [(3, 16, 32), (2, 64, 16), (4, 4, 256)])
def test_fp16_norm_pointwise_and_attention_pieces(ops, n, hw, c):
  rng = np.random.RandomState(6)
  y = f16_round(rng.randn(n, hw, hw, c) * 1.5 + 0.3)
  g, b = 1 + 0.1 * rng.randn(c), 0.1 * rng.randn(c)
  yd = to_dev(y, torch.float16).requires_grad_(True)
  gd, bd = to_dev(g).requires_grad_(True), to_dev(b).requires_grad_(True)
  z = ops.norm_act(yd, gd, bd)
  ref = N.pixel_norm(N.leaky_relu(N.instance_norm(y, g, b)))
  assert z.dtype == torch.float16 and rel_l2(host(z), ref) < 8e-4
  gz = f16_round(rng.randn(*y.shape))
  z.backward(to_dev(gz, torch.float16))
  yt = torch.tensor(y, dtype=torch.float64, requires_grad=True)
  gt, bt = torch.tensor(g, requires_grad=True), torch.tensor(b, requires_grad=True)
  from oracle import torch_ref as R
  zr = R.pixel_norm(torch.nn.functional.leaky_relu(R.instance_norm(yt, gt, bt), 0.2))
  zr.backward(torch.tensor(gz))
  assert rel_l2(host(yd.grad), yt.grad.numpy()) < 2e-3
  assert rel_l2(host(gd.grad), gt.grad.numpy()) < 1e-3 and rel_l2(host(bd.grad), bt.grad.numpy()) < 1e-3
  # fromRGB / toRGB and the batched GEMM of the attention layer
  x3 = to_dev(f16_round(rng.rand(n, hw, hw, 3)), torch.float16)
  w3 = to_dev(rng.randn(1, 1, 3, c) * 0.5)
  assert rel_l2(host(ops.pointwise_conv(x3, w3)), N.conv2d(host(x3), host(w3), 'SAME')) < 6e-4
  a = to_dev(f16_round(rng.randn(n, 40, 24)), torch.float16)
  bm = to_dev(f16_round(rng.randn(n, 24, 56)), torch.float16)
  assert rel_l2(host(ops.bgemm(a, bm)), host(a) @ host(bm)) < 6e-4


# ------------------------------------------------------------------------- flash attention
def _attention_ref(q, k, v):
  s = np.einsum('nid,njd->nij', q, k)
  s = s - s.max(axis=-1, keepdims=True)
  p = np.exp(s)
  p /= p.sum(axis=-1, keepdims=True)
  return np.einsum('nij,njd->nid', p, v)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('n,ln,dk,dv', [(2, 256, 8, 64), (1, 1024, 16, 128), (3, 128, 8, 256)])
def test_flash_attention_forward(ops, dtype, n, ln, dk, dv):
  """tg_flash_attention_fwd against softmax(q k^T) v in float64 on the 16-bit-rounded operands: the map is never
  written, the softmax statistics stay fp32 (the composed path rounds the scores to 16 bit first)."""
  rng = np.random.RandomState(7)
  rnd = bf16_round if dtype == torch.bfloat16 else f16_round
  q, k, v = rnd(np.tanh(rng.randn(n, ln, dk))), rnd(np.tanh(rng.randn(n, ln, dk) * 2)), rnd(rng.randn(n, ln, dv))
  qd, kd, vd = (to_dev(t, dtype) for t in (q, k, v))
  assert ops.flash_attention_supported(qd, vd)
  o, lse = ops.flash_attention_fwd_raw(qd, kd, vd)
  ref = _attention_ref(q, k, v)
  tol = 6e-3 if dtype == torch.bfloat16 else 8e-4      # P is rounded to the storage type before the second product
  assert rel_l2(host(o), ref) < tol
  s = np.einsum('nid,njd->nij', q, k)
  want = np.log(np.exp(s - s.max(-1, keepdims=True)).sum(-1)) + s.max(-1)
  # the row sums are sums of the 16-bit-rounded probabilities (the MFMA unit adds them up), as the numerator's are
  assert np.abs(host(lse) - want).max() < (2e-3 if dtype == torch.bfloat16 else 3e-4)


def _attention_grads_ref(q, k, v, go):
  s = np.einsum('nid,njd->nij', q, k)
  p = np.exp(s - s.max(-1, keepdims=True))
  p /= p.sum(-1, keepdims=True)
  gv = np.einsum('nij,nid->njd', p, go)
  gp = np.einsum('nid,njd->nij', go, v)
  gs = p * (gp - (gp * p).sum(-1, keepdims=True))
  return np.einsum('nij,njd->nid', gs, k), np.einsum('nij,nid->njd', gs, q), gv


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('n,ln,dk,dv', [(2, 256, 8, 64), (1, 512, 16, 128), (3, 128, 16, 64)])
def test_flash_attention_backward(ops, dtype, n, ln, dk, dv):
  """tg_flash_attention_bwd (dq, dk, dv) against the float64 softmax-attention gradients, and against the composed
  batched-GEMM / softmax path's own gradients on the same inputs (both 16-bit; the flash path keeps fp32 scores)."""
  rng = np.random.RandomState(11)
  rnd = bf16_round if dtype == torch.bfloat16 else f16_round
  q, k, v = rnd(np.tanh(rng.randn(n, ln, dk))), rnd(np.tanh(rng.randn(n, ln, dk) * 2)), rnd(rng.randn(n, ln, dv))
  go = rnd(rng.randn(n, ln, dv))
  qd, kd, vd = (to_dev(t, dtype).requires_grad_(True) for t in (q, k, v))
  assert ops.flash_attention_trainable(qd, vd)
  o = ops.flash_attention(qd, kd, vd)
  gq, gk, gv = torch.autograd.grad(o, (qd, kd, vd), to_dev(go, dtype))
  rq, rk, rv = _attention_grads_ref(q, k, v, go)
  tol = 1.2e-2 if dtype == torch.bfloat16 else 2e-3
  for got, ref, nm in ((gq, rq, 'dq'), (gk, rk, 'dk'), (gv, rv, 'dv')):
    assert rel_l2(host(got), ref) < tol, nm
  oc = ops.bgemm(ops.softmax_rows(ops.bgemm(qd, kd, False, True)), vd, False, False)
  cq, ck, cv = torch.autograd.grad(oc, (qd, kd, vd), to_dev(go, dtype))
  for got, ref, cmp_, nm in ((gq, rq, cq, 'dq'), (gk, rk, ck, 'dk'), (gv, rv, cv, 'dv')):
    assert rel_l2(host(got), ref) <= rel_l2(host(cmp_), ref) * 1.5 + 1e-4, nm      # at least as close as the composed path


def _attention_second_order_ref(q, k, v, go, aq, ak, av):
  """Gradients of <dq, aq> + <dk, ak> + <dv, av> (dq, dk, dv = the attention backward) wrt q, k, v, go: torch autograd
  in float64 on the CPU (oracle/np_ops.attention_backward_backward is the closed form of the same, pinned against this
  in tests/test_oracle.py)."""
  t = [torch.tensor(x, dtype=torch.float64, requires_grad=True) for x in (q, k, v, go)]
  tq, tk, tv, tg = t
  o = torch.softmax(tq @ tk.transpose(1, 2), -1) @ tv
  gq, gk, gv = torch.autograd.grad(o, (tq, tk, tv), tg, create_graph=True)
  loss = (gq * torch.tensor(aq)).sum() + (gk * torch.tensor(ak)).sum() + (gv * torch.tensor(av)).sum()
  return [x.numpy() for x in torch.autograd.grad(loss, t)]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('n,ln,dk,dv', [(2, 256, 8, 64), (1, 128, 16, 128)])
def test_flash_attention_second_order(ops, dtype, n, ln, dk, dv):
  """The gradient-penalty pattern through a flash node: create_graph backward (FlashAttnBwdFn) and then the backward of
  that (tg_flash_attention_bwd_bwd), against float64 autograd of the same composition and against the batched-GEMM /
  softmax path's own double backward."""
  rng = np.random.RandomState(5)
  rnd = bf16_round if dtype == torch.bfloat16 else f16_round
  q, k = rnd(np.tanh(rng.randn(n, ln, dk))), rnd(np.tanh(rng.randn(n, ln, dk) * 2))
  v, go = rnd(rng.randn(n, ln, dv)), rnd(rng.randn(n, ln, dv))
  aq, ak, av = rnd(rng.randn(n, ln, dk)), rnd(rng.randn(n, ln, dk)), rnd(rng.randn(n, ln, dv))
  ref = _attention_second_order_ref(q, k, v, go, aq, ak, av)
  res = {}
  for flash in (True, False):
    qd, kd, vd, gd = (to_dev(t, dtype).requires_grad_(True) for t in (q, k, v, go))
    if flash:
      with ops.second_order():
        assert ops.flash_attention_trainable(qd, vd) == ops.USE_FLASH_BWD_BWD
      o = ops.flash_attention(qd, kd, vd)
    else:
      o = ops.bgemm(ops.softmax_rows(ops.bgemm(qd, kd, False, True)), vd, False, False)
    gq, gk, gv = torch.autograd.grad(o, (qd, kd, vd), gd, create_graph=True)
    loss = (gq.float() * to_dev(aq, torch.float32)).sum() + (gk.float() * to_dev(ak, torch.float32)).sum() + \
        (gv.float() * to_dev(av, torch.float32)).sum()
    res[flash] = [host(t) for t in torch.autograd.grad(loss, (qd, kd, vd, gd))]
  tol = 2.5e-2 if dtype == torch.bfloat16 else 4e-3
  for got, cmp_, want, nm in zip(res[True], res[False], ref, ('adj q', 'adj k', 'adj v', 'adj dO')):
    e, ec = rel_l2(got, want), rel_l2(cmp_, want)
    print('[flash2] %s %s: flash %.2e composed %.2e' % (dtype, nm, e, ec))
    assert e < tol and e <= ec * 1.5 + 1e-4, (nm, e, ec)


# --------------------------------------------------------------- thin-output kernel (<= 16 output channels, TG_THIN16=1)
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('cin,cout', [(16, 16), (32, 16), (16, 8)])
def test_thin_output_kernel_matches_the_wide_block_kernels(ops, monkeypatch, dtype, cin, cout):
  """conv_thin16_kernel (v_mfma_f32_16x16x32, weights in registers; an A/B switch, off by default): forward with bias +
  LeakyReLU, forward with the statistics epilogue, backward-data with and without the LeakyReLU mask -- against the
  32-wide-block kernels the dispatch uses otherwise (same products, another summation order inside the MFMA: <= 2e-3) and
  against the float64 oracle on the same rounded operands."""
  import twingan_amd.ops as O
  from twingan_amd import _lib
  from twingan_amd._lib import TG_EPI_BIAS, TG_EPI_LRELU
  n, hw = 16, 128      # 2048 tiles: where the dispatch goes to the thin kernels
  g = torch.Generator().manual_seed(17)
  x = torch.randn(n, hw, hw, cin, generator=g).to(dtype).to(dev())
  w = (torch.randn(3, 3, cin, cout, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev())
  b = (torch.randn(cout, generator=g) * 0.1).to(dev())
  spec = O.ConvSpec(3, 'SAME')
  f16 = ',f16' if dtype == torch.float16 else ''

  def both(fn):
    out = []
    for on in ('0', '1'):
      monkeypatch.setenv('TG_THIN16', on)
      out.append((fn(), _lib.load().tg_last_kernel().decode()))
    monkeypatch.setenv('TG_THIN16', '0')
    return out
  (ya, ka), (yb, kb) = both(lambda: O.conv_fwd_raw(x, w, b, spec, TG_EPI_BIAS | TG_EPI_LRELU))
  assert 'thin16' not in ka and kb == 'conv_thin16_kernel<%d%s>' % (cin, f16), (ka, kb)
  assert rel_l2(host(yb), host(ya)) < 2e-3
  rnd = bf16_round if dtype == torch.bfloat16 else f16_round
  sub = slice(0, 2)      # the oracle on two images
  ref = N.leaky_relu(N.conv2d(host(x[sub]), rnd(host(w)), 'SAME') + host(b))
  assert rel_l2(host(yb[sub]), ref) < (6e-3 if dtype == torch.bfloat16 else 8e-4)
  if cout % 8 == 0:
    ((y1, s1), k1), ((y2, s2), k2) = both(lambda: O.conv_fwd_stats_raw(x, w, spec))
    assert k2 == 'conv_thin16_kernel<%d,stats%s>' % (cin, f16), k2
    assert rel_l2(host(y2), host(y1)) < 2e-3
    part = s2.part.view(n, s2.chunks, 2, cout).double().sum(dim=1).cpu().numpy()
    yd = y2.double()
    want = torch.stack([yd.sum(dim=(1, 2)), (yd * yd).sum(dim=(1, 2))], dim=1).cpu().numpy()
    assert rel_l2(part, want) < 1e-5      # the partials are sums of THIS tensor
  if cin <= 16:      # backward-data of this layer writes cin <= 16 channels: thin as well
    gy = torch.randn(n, hw, hw, cout, generator=g).to(dtype).to(dev())
    if cout == 16:
      (ga, _), (gb, k3) = both(lambda: O.conv_bwd_data_masked_raw(gy, w, x, spec))
      assert k3 == 'conv_thin16_kernel<16%s>' % f16, k3
      assert rel_l2(host(gb), host(ga)) < 2e-3
      want_g = N.conv2d_bwd_data(host(gy[sub]), rnd(host(w)), (hw, hw), 'SAME') * np.where(host(x[sub]) > 0, 1.0, 0.2)
      assert rel_l2(host(gb[sub]), want_g) < (6e-3 if dtype == torch.bfloat16 else 8e-4)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_thin_output_kernel_over_the_concat_input(ops, monkeypatch, dtype):
  """conv_thin16_upcat_kernel (TG_THIN16=1): the generator's concat conv with 16 outputs -- concat(nearest_up2(x0), skip),
  32 + 32 channels read in place, groups permuted as the trainer does -- forward with and without the statistics epilogue
  against the 32-wide-block UPCAT kernels (<= 2e-3: another summation order inside the MFMA)."""
  from twingan_amd import _lib
  g = torch.Generator().manual_seed(19)
  b, h, c0, c1, cout = 4, 64, 32, 32, 16      # 4 groups of 4 images at 128 x 128: 2048 tiles
  n = 4 * b
  x0 = torch.randn(n, h, h, c0, generator=g).to(dtype).to(dev())
  x1 = torch.randn(2 * b, 2 * h, 2 * h, c1, generator=g).to(dtype).to(dev())
  w = (torch.randn(3, 3, c0 + c1, cout, generator=g) * (2.0 / (9 * (c0 + c1))) ** 0.5).to(dev())
  f16 = ',f16' if dtype == torch.float16 else ''
  res = {}
  for on in ('0', '1'):
    monkeypatch.setenv('TG_THIN16', on)
    y_plain = ops.upcat_conv(x0, x1, w, b, (1, 0, 0, 1))
    k_plain = _lib.load().tg_last_kernel().decode()
    y, st = ops.upcat_conv_stats(x0, x1, w, b, (1, 0, 0, 1))
    res[on] = (y_plain, k_plain, y, st, _lib.load().tg_last_kernel().decode())
  monkeypatch.setenv('TG_THIN16', '0')
  assert 'thin16' not in res['0'][1] and res['1'][1] == 'conv_thin16_upcat_kernel<plain%s>' % f16, (res['0'][1], res['1'][1])
  assert res['1'][4] == 'conv_thin16_upcat_kernel<stats%s>' % f16
  y_plain, _, y, st, _ = res['1']
  assert st is not None and torch.equal(y, y_plain)
  assert rel_l2(host(y), host(res['0'][2])) < 2e-3
  part = st.part.view(n, st.chunks, 2, cout).double().sum(dim=1).cpu().numpy()
  yd = y.double()
  want = torch.stack([yd.sum(dim=(1, 2)), (yd * yd).sum(dim=(1, 2))], dim=1).cpu().numpy()
  assert rel_l2(part, want) < 1e-5
  # the float64 oracle on the first image of the last group (skip group 1)
  up = host(x0[3 * b:3 * b + 1]).repeat(2, axis=1).repeat(2, axis=2)
  cat = np.concatenate([up, host(x1[b:b + 1])], axis=-1)
  rnd = bf16_round if dtype == torch.bfloat16 else f16_round
  ref = N.conv2d(cat, rnd(host(w)), 'SAME')
  assert rel_l2(host(y[3 * b:3 * b + 1]), ref) < (6e-3 if dtype == torch.bfloat16 else 8e-4)


# ------------------------------------------------ backward-data of a block's last conv from the pooled gradient + sign bytes
UNPOOL_CASES = [
    # n, hw, cin, cout, masked, kernel the dispatch picks (bf16 name)
    (2, 16, 16, 32, True, 'conv_tile_kernel<3,32,32,1,unpool>'),
    (1, 32, 32, 64, False, 'conv_tile_kernel<3,32,32,1,unpool>'),
    (2, 16, 128, 256, True, 'conv_tile_kernel<3,32,32,1,unpool>'),
    (16, 64, 64, 128, True, 'conv_tile_kernel<3,32,32,2,unpool>'),     # two sub-tiles per wave (>= 1024 tiles, cin_pad >= 64)
    (32, 64, 64, 32, True, 'conv_tile_kernel<3,32,64,1,unpool>'),      # 64-channel output blocks
    (32, 64, 64, 128, False, 'conv_tile_kernel<3,32,64,2,unpool>'),
    (16, 128, 16, 32, True, 'conv_tile_wres_kernel<3,32,32,1,unpool>'),      # weight-resident thin kernel (the 256 x 256 block end)
    (1, 48, 16, 32, True, 'conv_tile_kernel<3,32,32,1,unpool>'),       # 48 rows / columns: three column tiles
]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('case', range(len(UNPOOL_CASES)))
def test_unpool_backward_data_equals_the_two_launch_path(ops, dtype, case):
  """tg_conv2d_bwd_data_unpool: the backward-data of a discriminator block's last conv (nets/pggan.py:304-306) with
  AvgPoolGrad + LeakyReluGrad applied while its tiles are staged -- bit-identical to tg_lrelu_pool_bwd_signs followed by
  tg_conv2d_bwd_data(_masked) for every kernel variant the dispatch picks, and within the rounding bound of the float64
  oracle evaluated on the same rounded operands."""
  import twingan_amd.ops as O
  from twingan_amd import _lib
  n, hw, cin, cout, masked, want = UNPOOL_CASES[case]
  g = torch.Generator().manual_seed(300 + case)
  spec = O.ConvSpec(3, 'SAME')
  x = torch.randn(n, hw, hw, cin, generator=g).to(dtype).to(dev())               # the conv's forward input: the mask source
  w = (torch.randn(3, 3, cin, cout, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev())
  gzp = torch.randn(n, hw // 2, hw // 2, cout, generator=g).to(dtype).to(dev())
  gzp[0, 0, 0, :8] = 0.0                                                         # zeros stay zeros whatever their sign bit
  signs = torch.randint(0, 256, (n, hw, hw, cout // 8), generator=g, dtype=torch.uint8).to(dev())
  g2, _ = O.lrelu_pool_bwd_signs(gzp, signs, spec.alpha, None, False)
  ref = O.conv_bwd_data_masked_raw(g2, w, x, spec) if masked else O.conv_bwd_data_raw(g2, w, (n, hw, hw, cin), spec)
  got = O.conv_bwd_data_unpool_raw(gzp, signs, w, x if masked else None, (n, hw, hw, cin), spec)
  assert got is not None, 'the tile kernels take this layer'
  sym = _lib.load().tg_last_kernel().decode()
  assert sym == (want if dtype == torch.bfloat16 else want.replace('unpool', 'unpool,f16')), sym
  assert torch.equal(got, ref), float((got.float() - ref.float()).abs().max())
  # keep mode (a discriminator step): the same kernel also writes the gradient tensor itself, each element exactly once
  got2, g2w = O.conv_bwd_data_unpool_raw(gzp, signs, w, x if masked else None, (n, hw, hw, cin), spec, True)
  assert _lib.load().tg_last_kernel().decode() == sym
  assert torch.equal(got2, ref) and torch.equal(g2w, g2), float((g2w.float() - g2.float()).abs().max())
  # the signs taken from the activation tensor itself (a pass that kept it): same kernels, 'unpoolz'
  zact = torch.randn(n, hw, hw, cout, generator=g).to(dtype).to(dev())
  zact[0, 0, :, :4] = 0.0                                                        # exact zeros count as "not positive"
  g3, _ = O.lrelu_pool_bwd(None, gzp, zact, spec.alpha, None, False)
  ref3 = O.conv_bwd_data_masked_raw(g3, w, x, spec) if masked else O.conv_bwd_data_raw(g3, w, (n, hw, hw, cin), spec)
  got3, g3w = O.conv_bwd_data_unpool_raw(gzp, zact, w, x if masked else None, (n, hw, hw, cin), spec, True)
  assert _lib.load().tg_last_kernel().decode() == sym.replace('unpool', 'unpoolz')
  assert torch.equal(got3, ref3) and torch.equal(g3w, g3)
  if n * hw * hw * cout <= 1 << 21:      # the float64 oracle on the small cases
    bits = ((signs.to(torch.int32).unsqueeze(-1) >> torch.arange(8, dtype=torch.int32, device=signs.device)) & 1).reshape(n, hw, hw, cout)
    up = host(gzp).repeat(2, axis=1).repeat(2, axis=2) * 0.25 * np.where(host(bits) > 0, 1.0, 0.2)
    rnd = bf16_round if dtype == torch.bfloat16 else f16_round
    wr = rnd(host(w))
    want_gx = N.conv2d_bwd_data(rnd(up), wr, (hw, hw), 'SAME')
    if masked:
      want_gx = want_gx * np.where(host(x) > 0, 1.0, 0.2)
    assert rel_l2(host(got), want_gx) < (4e-3 if dtype == torch.bfloat16 else 5e-4)


@pytest.mark.parametrize('train_d', [False, True])
def test_backward_through_a_block_end_uses_the_unpool_kernel(ops, train_d):
  """A discriminator block end: conv2d(pool_only=True) backpropagates through tg_conv2d_bwd_data_unpool -- differentiated for
  its INPUT only (a generator step: the discriminator's parameters are frozen) the layer's gradient is never in memory;
  with the filter and bias gradients wanted too (a discriminator step) the same kernel writes it for them.  Every gradient
  equals the two-launch path's (TG_DGRAD_UNPOOL=0): the input gradient bit for bit, the parameter gradients to fp32
  summation order."""
  import twingan_amd.ops as O
  g = torch.Generator().manual_seed(41)
  n, hw, cin, cout = 2, 32, 32, 64
  x0 = torch.randn(n, hw, hw, cin, generator=g).bfloat16().to(dev())
  w0 = (torch.randn(3, 3, cin, cout, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev())
  b0 = (torch.randn(cout, generator=g) * 0.1).to(dev())
  gz = torch.randn(n, hw // 2, hw // 2, cout, generator=g).bfloat16().to(dev())
  res, used = {}, []
  real = O.conv_bwd_data_unpool_raw

  def counted(*a, **k):
    out = real(*a, **k)
    used.append(out is not None)
    return out
  saved = O.USE_DGRAD_UNPOOL
  O.conv_bwd_data_unpool_raw = counted
  try:
    for on in (True, False):
      O.USE_DGRAD_UNPOOL = on
      x = x0.clone().requires_grad_(True)
      w, b = w0.clone().requires_grad_(train_d), b0.clone().requires_grad_(train_d)
      out = O.conv2d(x, w, b, 3, 'SAME', lrelu=True, pool=True, pool_only=True)
      zp = out[1] if isinstance(out, tuple) else out
      zp.backward(gz)
      res[on] = (x.grad, w.grad, b.grad)
  finally:
    O.USE_DGRAD_UNPOOL = saved
    O.conv_bwd_data_unpool_raw = real
  assert used == [True]
  assert torch.equal(res[True][0], res[False][0])
  if train_d:
    assert rel_l2(host(res[True][1]), host(res[False][1])) < 1e-5 and rel_l2(host(res[True][2]), host(res[False][2])) < 1e-5


# ------------------------------------------------------------------------- sign bits instead of a pooled layer's output
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('n,hw,cin,cout', [(3, 32, 16, 32), (2, 16, 64, 64), (2, 64, 16, 24), (5, 16, 32, 72), (1, 16, 256, 256)])
def test_conv_pool_sign_bits(ops, dtype, n, hw, cin, cout):
  """tg_conv2d_fwd_pool_signs / tg_lrelu_pool_bwd_signs (the discriminator blocks' last conv when the pool is the only
  consumer of its output, nets/pggan.py:304-306): the pooled tensor equals tg_conv2d_fwd_pool's bit for bit, the sign
  bytes are (z > 0) of the z that launch stores (incl. a channel count that ends on a lone byte: 24, 72), and the
  LeakyReLU + unpool backward rebuilt from the bits equals the one rebuilt from z -- both against the float64 oracle
  too.  Zeros in z (bias-free rows that cancel) count as "not positive", as lrelu'(0) = alpha in util_misc.py:86."""
  import twingan_amd.ops as O
  from twingan_amd._lib import TG_EPI_BIAS, TG_EPI_LRELU
  rng = np.random.RandomState(31)
  rnd = bf16_round if dtype == torch.bfloat16 else f16_round
  x = rnd(rng.randn(n, hw, hw, cin))
  x[0, :4] = 0.0                                   # a patch whose pre-activation is exactly the bias
  w = rnd(rng.randn(3, 3, cin, cout) / np.sqrt(9 * cin))
  b = rng.randn(cout) * 0.1
  b[:3] = 0.0                                      # ... which is zero for three channels: z == 0 there
  xd, wd, bd = to_dev(x, dtype), to_dev(w), to_dev(b)
  spec = O.ConvSpec(3, 'SAME')
  epi = TG_EPI_BIAS | TG_EPI_LRELU
  assert O.conv_fwd_pool_signs_supported(xd, wd, spec, epi)
  z, zp = O.conv_fwd_pool_raw(xd, wd, bd, spec, epi)
  signs, zp2 = O.conv_fwd_pool_signs_raw(xd, wd, bd, spec, epi)
  assert torch.equal(zp, zp2)
  assert signs.shape == (n, hw, hw, cout // 8) and signs.dtype == torch.uint8
  bits = (z > 0).view(n, hw, hw, cout // 8, 8).to(torch.int32)
  want = (bits << torch.arange(8, device=z.device, dtype=torch.int32)).sum(dim=-1).to(torch.uint8)
  assert torch.equal(signs, want), int((signs != want).sum())
  assert int((z[0, :3, :, :3] == 0).sum()) > 0      # the zero case is really in the data
  ref = N.leaky_relu(N.conv2d(x, w, 'SAME') + b)
  assert rel_l2(host(z), ref) < (6e-3 if dtype == torch.bfloat16 else 8e-4)
  gzp = rnd(rng.randn(n, hw // 2, hw // 2, cout))
  gd = to_dev(gzp, dtype)
  g_ref, gb_ref = O.lrelu_pool_bwd(None, gd, z, 0.2, bd, True)
  g_sig, gb_sig = O.lrelu_pool_bwd_signs(gd, signs, 0.2, bd, True)
  assert torch.equal(g_sig, g_ref)
  up = np.repeat(np.repeat(gzp, 2, axis=1), 2, axis=2) * 0.25
  want_g = up * np.where(host(z) > 0, 1.0, 0.2)
  assert rel_l2(host(g_sig), want_g) < (4e-3 if dtype == torch.bfloat16 else 5e-4)
  assert rel_l2(host(gb_sig), host(g_sig).sum(axis=(0, 1, 2))) < 1e-5 and rel_l2(host(gb_sig), host(gb_ref)) < 1e-5
  # through autograd: conv2d(pool_only=True) -> (None, pooled); its gradients equal the z-keeping node's
  res = {}
  for only in (True, False):
    xa = xd.clone().requires_grad_(True)
    wa, ba = wd.clone().requires_grad_(True), bd.clone().requires_grad_(True)
    full, pooled = O.conv2d(xa, wa, ba, 3, 'SAME', lrelu=True, pool=True, pool_only=only)
    assert (full is None) == only
    pooled.backward(gd)
    res[only] = (pooled.detach(), xa.grad, wa.grad, ba.grad)
  assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
  assert rel_l2(host(res[True][2]), host(res[False][2])) < 1e-6 and rel_l2(host(res[True][3]), host(res[False][3])) < 1e-5
  # a create_graph pass keeps z
  with O.second_order():
    full, _ = O.conv2d(xd, wd, bd, 3, 'SAME', lrelu=True, pool=True, pool_only=True)
  assert full is not None


# ------------------------------------------------------------------------- ordered (fixed-order) sums
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
def test_ordered_sums_are_bit_reproducible_and_right(ops, dtype):
  """tg_channel_sum_ordered / tg_sum_ordered / tg_pointwise_conv_bwd_weight_ordered (what deterministic mode routes the
  bias gradients, loss sums and fromRGB / toRGB filter gradients through): per-workgroup partials in a workspace added in
  workgroup order -- the same bits on every call (the atomic forms differ run to run in their last ulp at these sizes),
  the float64 value to fp32 summation error, also with fewer workspace rows than workgroups wanted."""
  import twingan_amd.ops as O
  from twingan_amd import _lib
  lib = _lib.load()
  rng = np.random.RandomState(41)
  n, hw, c = 6, 64, 32
  g = rng.randn(n, hw, hw, c).astype(np.float32)
  gd = to_dev(g, dtype)
  gref = host(gd)
  st = torch.cuda.current_stream().cuda_stream
  dt = O._dt(gd)
  outs = []
  for rows in (512, 512, 7):
    ws = torch.empty(rows * c, dtype=torch.float32, device='cuda')
    out = torch.full((c,), 3.0, dtype=torch.float32, device='cuda')
    O.call('tg_channel_sum_ordered', gd.data_ptr(), out.data_ptr(), n * hw * hw, c, 1, ws.data_ptr(), ws.numel(), dt, st)
    outs.append(out.clone())
    assert rel_l2(host(out) - 3.0, gref.sum(axis=(0, 1, 2))) < 2e-6
  assert torch.equal(outs[0], outs[1])
  # loss sums
  a, b = gd, to_dev(rng.randn(n, hw, hw, c).astype(np.float32), dtype)
  vals = []
  for _ in range(2):
    ws = torch.empty(512, dtype=torch.float32, device='cuda')
    o1 = torch.empty(1, dtype=torch.float32, device='cuda')
    o2 = torch.empty(1, dtype=torch.float32, device='cuda')
    O.call('tg_sum_ordered', a.data_ptr(), None, o1.data_ptr(), a.numel(), 0.5, 0, ws.data_ptr(), ws.numel(), dt, st)
    O.call('tg_sum_ordered', a.data_ptr(), b.data_ptr(), o2.data_ptr(), a.numel(), 2.0, 0, ws.data_ptr(), ws.numel(), dt, st)
    vals.append((float(o1), float(o2)))
  assert vals[0] == vals[1]
  assert abs(vals[0][0] - 0.5 * host(a).sum()) < 1e-5 * np.abs(host(a)).sum()
  assert abs(vals[0][1] - 2.0 * np.abs(host(a) - host(b)).sum()) < 1e-5 * 2.0 * np.abs(host(a) - host(b)).sum()
  # fromRGB filter gradient (cin 3) and toRGB (cout 3)
  if dtype != torch.float32:
    x3 = to_dev(rng.rand(n, hw, hw, 3).astype(np.float32), dtype)
    for xa, xb in ((x3, gd), (gd, x3)):
      ca, cb = xa.shape[-1], xb.shape[-1]
      res = []
      for _ in range(2):
        ws = torch.empty(256 * ca * cb, dtype=torch.float32, device='cuda')
        gw = torch.zeros((ca, cb), dtype=torch.float32, device='cuda')
        O.call('tg_pointwise_conv_bwd_weight_ordered', xa.data_ptr(), xb.data_ptr(), gw.data_ptr(), n * hw * hw, ca, cb, 1,
               ws.data_ptr(), ws.numel(), dt, st)
        res.append(gw.clone())
      assert torch.equal(res[0], res[1])
      want = host(xa).reshape(-1, ca).T @ host(xb).reshape(-1, cb)
      assert rel_l2(host(res[0]), want) < 2e-6
  # the host routes through them exactly when the mode is on
  was = lib.tg_set_deterministic(1)
  try:
    assert O.deterministic()
    s1, s2 = O.channel_sum_raw(gd), O.channel_sum_raw(gd)
    assert torch.equal(s1, s2)
  finally:
    lib.tg_set_deterministic(was)
  assert O.deterministic() == bool(was)


def test_deferred_slab_reductions_equal_immediate_ones(ops):
  """tg_wgrad_defer / tg_wgrad_defer_flush: the split-K slab reductions of filter gradients that accumulate into a sink are
  queued and issued as ONE launch.  Each job keeps the slice-group shape and summation order of its stand-alone kernel, so
  a sink that starts at zero receives the same bits; two jobs into one sink add up; a queue of more than 120 jobs falls
  back to immediate launches for the rest; outside a defer window nothing changes."""
  spec = ops.ConvSpec(3, 'SAME')
  g = torch.Generator(device='cpu').manual_seed(5)
  cases = [(4, 16, 16, 32, 32), (8, 32, 32, 64, 64), (2, 64, 64, 16, 16), (16, 8, 8, 256, 256), (4, 16, 16, 512, 256)]
  data = []
  for n, h, w, cin, cout in cases:      # weight sizes on both sides of the 16 384 / 131 072 element thresholds
    x = torch.randn(n, h, w, cin, generator=g).to(dev()).to(torch.bfloat16)
    gy = torch.randn(n, h, w, cout, generator=g).to(dev()).to(torch.bfloat16)
    data.append((x, gy, torch.zeros(3, 3, cin, cout, device=dev()), torch.zeros(3, 3, cin, cout, device=dev())))
  for x, gy, now, _ in data:
    ops.conv_bwd_weight_raw(x, gy, spec, out=now)
  ops.defer_slab_reductions(True)
  try:
    for x, gy, _, later in data:
      ops.conv_bwd_weight_raw(x, gy, spec, out=later)
    torch.cuda.synchronize()
    queued = [float(later.abs().max()) for _, _, _, later in data]
    n_flushed = ops.flush_slab_reductions()
  finally:
    ops.defer_slab_reductions(False)
  torch.cuda.synchronize()
  assert n_flushed >= 1 and ops.flush_slab_reductions() == 0
  assert sum(1 for q in queued if q == 0.0) == n_flushed      # the queued sinks were untouched before the flush
  for (x, gy, now, later), q in zip(data, queued):
    assert torch.equal(now, later), (tuple(x.shape), float((now - later).abs().max()))
  # two jobs into one sink, and more jobs than the table holds
  x, gy, now, later = data[[i for i, q in enumerate(queued) if q == 0.0][0]]      # a layer whose reduction is a slab job
  ops.conv_bwd_weight_raw(x, gy, spec, out=now)
  many = [torch.zeros_like(now) for _ in range(130)]
  ref = torch.zeros_like(now)
  ops.conv_bwd_weight_raw(x, gy, spec, out=ref)
  ops.defer_slab_reductions(True)
  try:
    ops.conv_bwd_weight_raw(x, gy, spec, out=later)
    for m in many:
      ops.conv_bwd_weight_raw(x, gy, spec, out=m)
    n2 = ops.flush_slab_reductions()
  finally:
    ops.defer_slab_reductions(False)
  torch.cuda.synchronize()
  assert n2 == 120
  assert rel_l2(host(later), host(now)) < 1e-6
  for m in many:
    assert torch.equal(m, ref)


# ---------------------------------------------------------------------------------------------- grouped convs
GROUPED_CASES = [
    # n (both groups), hw, cin, cout, k, padding        -- what the two discriminators run from 32 x 32 down
    (4, 32, 16, 32, 3, 'SAME'),      # conv_tile
    (4, 16, 32, 32, 3, 'SAME'),      # conv_tile
    (4, 8, 32, 32, 3, 'SAME'),       # conv_img
    (6, 4, 24, 32, 3, 'SAME'),       # conv_small (the minibatch-stddev layer's padded channel count is no multiple of 16)
    (4, 4, 32, 32, 4, 'VALID'),      # the dense layer
]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('case', range(len(GROUPED_CASES)))
def test_grouped_conv_equals_one_call_per_weight_set(ops, dtype, case):
  """TgConvDesc.groups = 2 (a stacked [2, kh, kw, cin, cout] kernel: the layers of discriminator_s and discriminator_t,
  twingan.py:105-110, as ONE call over the batch [D_s rows; D_t rows]): every conv entry point the discriminators' layers
  reach gives each image range exactly what the single-set call with its own weights gives it -- bit for bit in the
  forward and backward-data forms (per-image arithmetic), to fp32 summation order in the filter / bias gradients."""
  import twingan_amd.ops as O
  n, hw, cin, cout, k, padding = GROUPED_CASES[case]
  g = torch.Generator().manual_seed(500 + case)
  spec = O.ConvSpec(k, padding)
  ho = hw if padding == 'SAME' else hw - k + 1
  h = n // 2
  x = torch.randn(n, hw, hw, cin, generator=g).to(dtype).to(dev())
  w2 = (torch.randn(2, k, k, cin, cout, generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(dev())
  b2 = (torch.randn(2, cout, generator=g) * 0.1).to(dev())
  gy = torch.randn(n, ho, ho, cout, generator=g).to(dtype).to(dev())
  epi = O.TG_EPI_BIAS | O.TG_EPI_LRELU

  def per_set(fn):
    return [fn(i, slice(i * h, (i + 1) * h)) for i in range(2)]

  def same(got, parts):
    want = torch.cat(parts)
    assert got.shape == want.shape and torch.equal(got, want), float((got.float() - want.float()).abs().max())

  # forward, bias + LeakyReLU epilogue
  y = O.conv_fwd_raw(x, w2, b2, spec, epi)
  if dtype != torch.float32:      # the MFMA kernels of these shapes pick the weight set per image: ONE launch
    from twingan_amd import _lib
    sym = _lib.load().tg_last_kernel().decode()
    assert sym.startswith(('conv_tile_kernel', 'conv_img_kernel', 'conv_small_kernel')[min(case, 3) - (1 if case > 0 else 0)]) and \
        sym.endswith(',sets>'), sym
  same(y, per_set(lambda i, r: O.conv_fwd_raw(x[r].contiguous(), w2[i], b2[i], spec, epi)))
  # backward-data, plain and with the producer's LeakyReLU mask
  same(O.conv_bwd_data_raw(gy, w2, tuple(x.shape), spec),
       per_set(lambda i, r: O.conv_bwd_data_raw(gy[r].contiguous(), w2[i], (h,) + tuple(x.shape[1:]), spec)))
  same(O.conv_bwd_data_masked_raw(gy, w2, x, spec),
       per_set(lambda i, r: O.conv_bwd_data_masked_raw(gy[r].contiguous(), w2[i], x[r].contiguous(), spec)))
  # forward with the mask epilogue (the gradient penalty's second backward)
  same(O.conv_fwd_masked_raw(x, w2, y, spec),
       per_set(lambda i, r: O.conv_fwd_masked_raw(x[r].contiguous(), w2[i], y[r].contiguous(), spec)))
  if padding == 'SAME' and dtype != torch.float32 and hw >= 16:
    # block ends: pooled output (+ sign bytes), and the backward-data that unpools
    z, zp = O.conv_fwd_pool_raw(x, w2, b2, spec, epi)
    ref = per_set(lambda i, r: O.conv_fwd_pool_raw(x[r].contiguous(), w2[i], b2[i], spec, epi))
    same(z, [p[0] for p in ref])
    same(zp, [p[1] for p in ref])
    assert O.conv_fwd_pool_signs_supported(x, w2, spec, epi)
    sg, zp2 = O.conv_fwd_pool_signs_raw(x, w2, b2, spec, epi)
    ref = per_set(lambda i, r: O.conv_fwd_pool_signs_raw(x[r].contiguous(), w2[i], b2[i], spec, epi))
    same(sg, [p[0] for p in ref])
    same(zp2, [p[1] for p in ref])
    gzp = torch.randn(n, ho // 2, ho // 2, cout, generator=g).to(dtype).to(dev())
    out = O.conv_bwd_data_unpool_raw(gzp, sg, w2, x, tuple(x.shape), spec, True)
    assert out is not None
    ref = per_set(lambda i, r: O.conv_bwd_data_unpool_raw(gzp[r].contiguous(), sg[r].contiguous(), w2[i], x[r].contiguous(),
                                                          (h,) + tuple(x.shape[1:]), spec, True))
    same(out[0], [p[0] for p in ref])
    same(out[1], [p[1] for p in ref])
  # filter (+ bias) gradients: into zeroed stacked sinks
  tol = 1e-6 if dtype == torch.float32 else 2e-5
  gw = O.conv_bwd_weight_raw(x, gy, spec, groups=2)
  ref = torch.stack(per_set(lambda i, r: O.conv_bwd_weight_raw(x[r].contiguous(), gy[r].contiguous(), spec)))
  assert gw.shape == ref.shape and rel_l2(host(gw), host(ref)) < tol
  if dtype != torch.float32:
    sink, bsink = torch.zeros_like(w2), torch.zeros_like(b2)
    O.conv_bwd_weight_raw(x, gy, spec, out=sink, gbias=bsink)
    assert rel_l2(host(sink), host(ref)) < tol
    assert rel_l2(host(bsink), host(gy.float().reshape(2, -1, cout).sum(1))) < 1e-3
    if O.conv_bwd_weight2_raw(x, gy, x[:n].contiguous(), gy[:n].contiguous(), spec, sink, bsink, 3):      # two segments, both grouped
      assert rel_l2(host(sink), 3.0 * host(ref)) < tol


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_latent_layer_as_a_gemm_equals_the_padded_valid_conv(ops, dtype, monkeypatch):
  """ops.latent_conv: the plain PGGAN generator's first layer (nets/pggan.py:135-153, a 4x4 VALID conv over the latent noise
  zero-padded to 7x7) as [B, C] @ [C, 16 C'] of the flipped kernel -- output, kernel gradient and noise gradient against the
  float64 oracle's conv over the padded tensor, and against the conv kernel it replaces (which the 16-bit MFMA dispatch does
  not take: under TG_STRICT_DISPATCH=1 that fallback is an error, not a 100x slower launch)."""
  import twingan_amd.ops as O
  from twingan_amd import _lib
  g = torch.Generator().manual_seed(77)
  b, c, co, k = 6, 32, 24, 4
  rnd = bf16_round if dtype == torch.bfloat16 else f16_round
  noise = torch.randn(b, 1, 1, c, generator=g)
  w = torch.randn(k, k, c, co, generator=g) * (2.0 / (k * k * c)) ** 0.5
  gy = torch.randn(b, k, k, co, generator=g)
  nd = noise.to(dtype).to(dev()).requires_grad_(True)
  wd = w.to(dev()).requires_grad_(True)
  y = O.latent_conv(nd, wd)
  assert y.shape == (b, k, k, co) and y.dtype == dtype
  y.backward(gy.to(dtype).to(dev()))
  # oracle: the VALID conv over the zero-padded noise, operands as stored
  xpad = np.zeros((b, 2 * k - 1, 2 * k - 1, c))
  xpad[:, k - 1, k - 1, :] = rnd(host(noise))[:, 0, 0, :]
  want = N.conv2d(xpad, host(w), 'VALID')
  assert rel_l2(host(y), want) < (1e-2 if dtype == torch.bfloat16 else 2e-3)
  gyr = rnd(host(gy))
  want_gw = N.conv2d_bwd_weight(xpad, gyr, (k, k), 'VALID')
  assert rel_l2(host(wd.grad), want_gw) < 1e-5      # fp32 products of the stored operands, fp32 sums
  want_gx = N.conv2d_bwd_data(gyr, host(w), (2 * k - 1, 2 * k - 1), 'VALID')[:, k - 1, k - 1, :]
  assert rel_l2(host(nd.grad).reshape(b, c), want_gx) < (1e-2 if dtype == torch.bfloat16 else 2e-3)
  # the conv it replaces: same numbers from the direct kernel -- with a warning, and an error in strict mode
  xp = torch.nn.functional.pad(noise.to(dtype).to(dev()), (0, 0, k - 1, k - 1, k - 1, k - 1)).contiguous()
  O._SLOW_SEEN.clear()
  c2, co2 = 64, 64      # above the 1 MFLOP threshold
  xp2 = torch.zeros(b, 7, 7, c2, dtype=dtype, device=dev())
  w2 = torch.zeros(4, 4, c2, co2, device=dev())
  with pytest.warns(UserWarning, match='conv_\\*_direct'):
    O.conv2d(xp2, w2, None, 4, 'VALID')
  monkeypatch.setenv('TG_STRICT_DISPATCH', '1')
  with pytest.raises(_lib.TgError, match='TG_STRICT_DISPATCH'):
    O.conv2d(xp2, w2, None, 4, 'VALID')
  monkeypatch.delenv('TG_STRICT_DISPATCH')
  ref = O.conv2d(xp, wd.detach(), None, 4, 'VALID')
  assert rel_l2(host(y), host(ref)) < (1e-2 if dtype == torch.bfloat16 else 2e-3)
