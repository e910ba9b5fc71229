"""Storage-rounding sensitivity of the TwinGAN graph (TEST INFRASTRUCTURE, like the rest of oracle/).

The 16-bit HIP paths store activations, weight packs and back-propagated tensors in bf16 / fp16 and accumulate in fp32.
``storage_rounding(dtype)`` inserts exactly those roundings into the float64 oracle (conv inputs and outputs, layer
outputs, pooled tensors, the conv weights; forward values and the gradients flowing back through the same points), so
that "how far may a correct 16-bit implementation be from the float64 gradients of THIS graph on THESE inputs" becomes a
number the parity tests can assert against, instead of a directional bound.  Reference semantics being perturbed:
nets/pggan.py / nets/pggan_utils.py layers as restated in oracle/torch_ref.py; no reference code is involved in the rounding.
"""
import contextlib

import torch

from . import torch_ref as R


@contextlib.contextmanager
def storage_rounding(dtype, forward=True, backward=True, weights=True):
  """Within the block the oracle's generator / encoder / discriminator convs round like the 16-bit storage path."""
  fdt, bdt, wdt = (dtype if forward else None), (dtype if backward else None), (dtype if weights else None)

  class Rnd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
      return x.to(fdt).to(x.dtype) if fdt else x

    @staticmethod
    def backward(ctx, g):
      return g.to(bdt).to(g.dtype) if bdt else g
  rnd = Rnd.apply

  def wr(w):      # straight-through: the pack is rounded, the gradient belongs to the fp32 master
    return w + (w.detach().to(wdt).to(w.dtype) - w.detach()) if wdt else w
  saved = dict(conv2d=R.conv2d, ge_conv=R.ge_conv, d_conv=R.d_conv, avg_pool2=R.avg_pool2, self_attention=R.self_attention)
  conv, inorm, pnorm, lrelu, pool = R.conv2d, R.instance_norm, R.pixel_norm, R.leaky_relu, R.avg_pool2

  def conv2d(x, w, padding):
    return rnd(conv(rnd(x), wr(w), padding))

  def ge_conv(P, scope, x, domain, cfg, k=3, padding='SAME', act=True, pixnorm=True, equalized=True, cond=None, spectral=True):
    assert cond is None and cfg.norm == 'instance_norm' and not cfg.equalized, \
        'the sensitivity probe covers the headline configuration (+ spectral norm / attention)'
    # a spectrally normalised kernel is computed in fp32 from the master and THEN packed to the storage format
    w = R.spectral_normed_weight(P, scope, cfg, False) if spectral else P[scope + '/weights']
    y = conv2d(x, w, padding)
    y = inorm(y, P[scope + '/InstanceNorm/gamma_' + domain], P[scope + '/InstanceNorm/beta_' + domain], cfg.in_eps)
    if act:
      y = lrelu(y, cfg.lrelu)
    if pixnorm and cfg.do_pixel_norm:
      y = pnorm(y, cfg.pn_eps)
    return rnd(y)

  def d_conv(P, scope, x, cfg, k=3, padding='SAME', **kw):
    assert not cfg.equalized
    y = conv(rnd(x), wr(R.spectral_normed_weight(P, scope, cfg)), padding) + P[scope + '/biases']
    return rnd(lrelu(y, cfg.lrelu))

  def self_attention(P, sc, layer, domain, cfg, is_discriminator, cond=None):
    """libs/self_attention.py:24-70 with the storage points of the HIP path: f, g (after tanh) and h are stored tensors, the
    softmax map is rounded where the flash kernels pack it for the second product, o and the gated sum are stored."""
    n, hh, ww, c = layer.shape
    outs = []
    for nm in ('sa_f', 'sa_g', 'sa_h'):
      scope = '%s/%s' % (sc, nm)
      if is_discriminator:
        y = rnd(conv2d(layer, P[scope + '/weights'], 'SAME') + P[scope + '/biases'])
      else:
        y = ge_conv(P, scope, layer, domain, cfg, k=1, act=False, pixnorm=False, equalized=False, cond=cond, spectral=False)
      outs.append(rnd(torch.tanh(y)) if nm != 'sa_h' else y)
    f, g_, h = outs
    npos = hh * ww
    s_ = torch.bmm(f.reshape(n, npos, -1), g_.reshape(n, npos, -1).transpose(1, 2))
    beta = rnd(torch.softmax(s_, dim=-1))
    o = rnd(torch.bmm(beta, h.reshape(n, npos, c)).reshape(layer.shape))
    return rnd(P[sc + '/sa_gamma'] * o + layer)
  R.conv2d, R.ge_conv, R.d_conv, R.self_attention = conv2d, ge_conv, d_conv, self_attention
  R.avg_pool2 = lambda x: rnd(pool(x))
  try:
    yield
  finally:
    for k, v in saved.items():
      setattr(R, k, v)


def gradient_sensitivity(P, names, loss_fn, dtype, reset=None, **which):
  """rel-L2 over the variables ``names`` between the float64 gradients of ``loss_fn(Q)`` (Q: a fresh copy of P that requires
  grad) and those of the same graph with ``dtype`` storage rounding inserted.  ``reset``: called before each of the two
  evaluations (spectral norm: put the pre-run u back, drop the run's normalised kernels).
  -> (rel_l2, rounded gradients, exact gradients)."""
  def grads():
    if reset is not None:
      reset()
    Q = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    return R.grads_of(loss_fn(Q), Q, names)
  exact = grads()
  with storage_rounding(dtype, **which):
    rounded = grads()
  num = sum(float(((rounded[k] - exact[k]) ** 2).sum()) for k in exact)
  den = sum(float((exact[k] ** 2).sum()) for k in exact)
  return (num / den) ** 0.5, rounded, exact


def generator_gradient_sensitivity(P, s, t, cfg, dtype, **which):
  """rel-L2 over the generator group between the float64 gradients of generator_loss and those of the same graph with
  ``dtype`` storage rounding inserted.  P: float64 parameters (fp32-representable); s, t: float64 images already
  rounded to ``dtype``.  Returns (rel_l2, rounded gradients dict, exact gradients dict)."""
  names = R.generator_var_names(P)

  def grads():
    Q = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    loss, _ = R.generator_loss(Q, s, t, cfg)
    return R.grads_of(loss, Q, names)
  exact = grads()
  with storage_rounding(dtype, **which):
    rounded = grads()
  num = sum(float(((rounded[k] - exact[k]) ** 2).sum()) for k in exact)
  den = sum(float((exact[k] ** 2).sum()) for k in exact)
  return (num / den) ** 0.5, rounded, exact
