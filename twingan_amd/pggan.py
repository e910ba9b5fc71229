"""PGGAN-style encoder / generator / discriminator on the gfx950 kernels.

Mirror of the reference network functions (nets/pggan.py) and layer helpers (nets/pggan_utils.py):
same function names, same end-point names, same variable names, same layer order
(conv -> norm -> LeakyReLU -> pixel-norm for G/E; conv + bias -> LeakyReLU for D).  What differs is
the execution: every layer is one or two fused HIP kernels (twingan_amd/ops.py) instead of 6-10
stock TF ops, and per-domain normaliser parameters are selected by ``domain`` ('s' | 't') instead of
a TF variable-scope postfix (conditional_layer_var_scope_postfix, nets/pggan_utils.py:102-113).

Network functions return ``(output, end_points)`` like the reference.  Tensors are NHWC.
"""
import math

from . import ops
from .params import NATIVE_NORM, NORM_SCOPE, get_num_channels, max_stage_of, mbstd_cpad, norm_var


# ------------------------------------------------------------------------------------------------
# layer helpers (nets/pggan_utils.py)
# ------------------------------------------------------------------------------------------------
def _pf(d):
  """Variable-name postfix of a domain (conditional_layer_var_scope_postfix, nets/pggan_utils.py:102-113): '_s' / '_t' in
  TwinGAN, nothing in the plain PGGAN trainer (image_generation.py)."""
  return '_' + d if d else ''


def _equalize(x, cfg, k, in_ch=None):
  """maybe_equalized_conv2d / maybe_equalized_fc (nets/pggan_utils.py:236-254): with --equalized_learning_rate the
  layer INPUT is scaled by sqrt(2 / (in_ch * k^2)); weights are then N(0,1)-initialised (params.declare_twingan)."""
  if not cfg.equalized_learning_rate:
    return x
  return ops.scale(x, math.sqrt(2.0 / ((in_ch or x.shape[-1]) * k * k)))


def _l2_normalize(x):
  """tf.nn.l2_normalize over all elements: x / sqrt(max(sum(x^2), 1e-12))."""
  return x / x.pow(2).sum().clamp_min(1e-12).sqrt()


def _sn(P, scope, cfg, is_discriminator):
  """The conv kernel of ``scope``, spectrally normalised when --spectral_norm applies to it
  (nets/pggan_utils.py:316-320; libs/sn.py:38-101): one power iteration from the persistent vector u,
      v = l2n(u W^T),  u' = l2n(v W),  sigma = v W u'^T,  W_bar = W / sigma      (W = kernel as [k*k*cin, cout])
  with the gradient flowing through sigma, v and u' (the reference does not stop it).  The reference re-runs the
  iteration and assigns u at EVERY use of the kernel inside one session.run, in unspecified order; here every use in
  a run sees the pre-run u (one valid schedule of those unordered assigns) and u is assigned once, by end_run().
  The iteration and its backward are the tg_spectral_norm_fwd / _bwd kernels on the fp32 master weight
  (ops.SpectralNormFn)."""
  w = P[scope + '/weights']
  if not (cfg.spectral_norm and (is_discriminator or cfg.spectral_norm_in_non_discriminator)):
    return w
  cache = P.__dict__.setdefault('sn_cache', {})
  if scope in cache:
    return cache[scope]
  w_bar = _sn_compute(P, scope)
  # not prepared by prepare_run (tests that call a network function directly): the packs made from this buffer in an
  # earlier run are stale -- rebuild the ones that exist (one launch); new ones are packed from the fresh values
  ops.PackCache.refresh([P.__dict__['sn_wbar'][scope]])
  return w_bar


def _sn_compute(P, scope):
  """One power iteration for ``scope`` into its persistent w_bar buffer; notes u' for end_run()."""
  w = P[scope + '/weights']
  bufs = P.__dict__.setdefault('sn_wbar', {})
  buf = bufs.get(scope)
  if buf is None or buf.numel() != w.numel() or buf.device != w.device:
    import torch
    buf = bufs[scope] = torch.empty(w.numel(), dtype=torch.float32, device=w.device)
    ops.PackCache.register(buf)
  w_bar, u1 = ops.spectral_norm(w, P.state[scope + '/u'], out=buf)
  P.__dict__.setdefault('sn_pending', {})[scope + '/u'] = u1
  w_bar = _sn_leaf(P, scope, w_bar)
  P.__dict__.setdefault('sn_cache', {})[scope] = w_bar
  return w_bar


def _sn_leaf(P, scope, w_bar):
  """The tensor the uses of a normalised kernel read.  Under the trainer (P.sn_sink_mode) or a segmented backward (ops.Cuts)
  it is a detached LEAF: one kernel serves several passes (the batched pass and the gradient-penalty pass of a discriminator
  step, in different backward segments when the step is cut), so its node must not sit inside any of them, and
  sn_segment_backward() sends the accumulated d loss / d w_bar through the power iteration's backward ONCE.
  Trainer mode also gives the leaf a GRADIENT SINK (a persistent fp32 buffer, zeroed by prepare_run): the filter-gradient
  kernels then add into it themselves -- paired launches, deferred slab reductions, the bias gradient riding along, as for
  an unnormalised kernel -- instead of one gradient tensor per use summed by the framework (config 4: ~80 framework adds
  and ~30 stand-alone slab reductions per step)."""
  sink_mode = bool(P.__dict__.get('sn_sink_mode'))
  if not ((ops.Cuts.active or sink_mode) and torch_is_grad_enabled() and w_bar.requires_grad):
    return w_bar
  leaf = w_bar.detach().requires_grad_(True)
  gbuf = None
  if sink_mode:
    gbuf = _sn_grad_buffer(P, scope, leaf)
    ops.GradSink.register(leaf, gbuf)
  P.__dict__.setdefault('sn_leaves', {})[scope] = (w_bar, leaf, gbuf)
  return leaf


def _sn_grad_buffer(P, scope, leaf):
  """The gradient sink of ``scope``'s w_bar: a slice of ONE flat fp32 buffer (so that prepare_run zeroes all of them in one
  launch), laid out on first use in the order the kernels are prepared."""
  import torch
  lay = P.__dict__.setdefault('sn_glayout', {})
  if scope not in lay:
    off = P.__dict__.get('sn_gsize', 0)
    lay[scope] = (off, leaf.numel())
    P.__dict__['sn_gsize'] = off + leaf.numel()
    P.__dict__['sn_gflat'] = None      # grown: re-allocated below
  flat = P.__dict__.get('sn_gflat')
  if flat is None or flat.numel() < P.__dict__['sn_gsize'] or flat.device != leaf.device:
    flat = P.__dict__['sn_gflat'] = torch.zeros(P.__dict__['sn_gsize'], dtype=torch.float32, device=leaf.device)
  off, n = lay[scope]
  return flat[off:off + n].view(leaf.shape)


def _sn_compute_multi(P, scopes):
  """_sn_compute for every scope of ``scopes`` with ONE batched power iteration (ops.spectral_norm_multi: three launches
  for all kernels instead of three per kernel); the per-kernel autograd nodes, the pending u' and the Cuts leaves as there."""
  import torch
  bufs = P.__dict__.setdefault('sn_wbar', {})
  items = []
  for scope in scopes:
    w = P[scope + '/weights']
    buf = bufs.get(scope)
    if buf is None or buf.numel() != w.numel() or buf.device != w.device:
      buf = bufs[scope] = torch.empty(w.numel(), dtype=torch.float32, device=w.device)
      ops.PackCache.register(buf)
    items.append((w, P.state[scope + '/u'], buf))
  outs, table = ops.spectral_norm_multi(items, P.__dict__.get('sn_table'))
  P.__dict__['sn_table'] = table
  P.__dict__['sn_table_scopes'] = list(scopes)
  for scope, (w_bar, u1) in zip(scopes, outs):
    P.__dict__.setdefault('sn_pending', {})[scope + '/u'] = u1
    P.__dict__.setdefault('sn_cache', {})[scope] = _sn_leaf(P, scope, w_bar)


def torch_is_grad_enabled():
  import torch
  return torch.is_grad_enabled()


def sn_segment_backward(P, done):
  """After a backward segment: every spectrally normalised kernel whose uses all lie in finished segments (``done(scope)``)
  gets its accumulated d loss / d w_bar sent through the normalisation's backward into the master kernel's gradient."""
  import torch
  leaves = P.__dict__.get('sn_leaves')
  if not leaves:
    return 0
  roots, grads = [], []
  for scope in [k for k in leaves if done(k)]:
    w_bar, leaf, gbuf = leaves.pop(scope)
    if gbuf is not None:      # trainer mode: the filter-gradient kernels added into the sink (flushed by the caller before this)
      # every consumer of the leaf adds into the sink itself; one that handed autograd a gradient tensor instead (an op
      # without sink support, a backward run with grad mode on) would be dropped here -- loudly, not silently
      assert leaf.grad is None, 'a consumer of the normalised kernel %s returned a gradient tensor next to its sink' % scope
      ops.GradSink.unregister(leaf)
      roots.append(w_bar)
      grads.append(gbuf)
    elif leaf.grad is not None:
      roots.append(w_bar)
      grads.append(leaf.grad)
  if roots:
    torch.autograd.backward(roots, grads)
  return len(roots)


def prepare_run(P, cfg):
  """Start of one session.run equivalent under --spectral_norm: the power iteration of EVERY normalised kernel (each is
  used by the run: a generator step also runs the discriminators forward, a discriminator step the encoder / generator)
  from the pre-run u, then ONE launch that rebuilds all their MFMA packs (PackCache.refresh) -- instead of one pack
  launch per kernel, direction and use (1 196 launches in five config-4 steps, profiles/r02_g_bench_c4_kernel_stats.csv)."""
  if not cfg.spectral_norm:
    return 0
  scopes = [k[:-2] for k in P.state if k.endswith('/u')]
  flat = P.__dict__.get('sn_gflat')
  if flat is not None and P.__dict__.get('sn_sink_mode'):
    ops.zero_(flat)      # the w_bar gradient sinks of this run (one launch for all of them)
  cache = P.__dict__.setdefault('sn_cache', {})
  todo = [scope for scope in scopes if scope not in cache]
  if len(todo) > 1:      # one tg_spectral_norm_fwd_multi for every kernel of the run (config 4: 60 launches -> 3, +2.5 %)
    _sn_compute_multi(P, todo)
    todo = []
  for scope in todo:
    _sn_compute(P, scope)
  bufs = P.__dict__.get('sn_wbar', {})
  if bufs:
    ops.PackCache.refresh(list(bufs.values()))
  return len(scopes)


def end_run(P):
  """End of one session.run equivalent: assign the power-iteration vectors (libs/sn.py:84-86) and drop the
  per-run normalised kernels."""
  pending = P.__dict__.get('sn_pending')
  if pending:
    table = P.__dict__.get('sn_table')
    done = set()
    if table is not None and P.__dict__.get('sn_table_scopes'):
      # the kernels whose power iteration ran through the job table: ONE launch assigns all their u (the table reads
      # P.state's u buffers in place, so their addresses are the table's)
      scopes = P.__dict__['sn_table_scopes']
      if all(s + '/u' in pending and P.state[s + '/u'].data_ptr() == key[1] for s, key in zip(scopes, table.key)):
        table.assign_u()
        done = {s + '/u' for s in scopes}
    for k, v in pending.items():
      if k not in done:
        P.state[k].copy_(v)
    pending.clear()
  cache = P.__dict__.get('sn_cache')
  if cache:
    cache.clear()
  leaves = P.__dict__.get('sn_leaves')
  if leaves:
    leaves.clear()


def self_attention_layer(P, sc, layer, domain, cfg, is_discriminator, cond=None):
  """libs/self_attention.py:24-70 (SAGAN): f, g = tanh(conv1x1 -> c/8), h = conv1x1 -> c under the scope's arg-scope
  (bias in D; the generator normaliser, no bias, in G / E -- nets/pggan_utils.py:86-98), s = f g^T over the h*w
  positions, beta = softmax(s), o = beta h, y = sa_gamma * o + layer.  The two batched matrix products are the MFMA
  batched GEMM of csrc/attention.hip, softmax / tanh / the sa_gamma scale its row and pointwise kernels; every op's
  backward is made of the same ops, so the layer is differentiable twice on them (the discriminator sits under the
  gradient penalty).  At supported shapes the three map ops run as the flash kernels of csrc/flash.hip
  (ops.flash_attention: forward, backward and the backward of the backward), which never write the [h*w, h*w] map."""
  n, hh, ww, c = layer.shape
  outs = []
  for nm in ('sa_f', 'sa_g', 'sa_h'):
    scope = '%s/%s' % (sc, nm)
    if is_discriminator:
      w, b = P[scope + '/weights'], P[scope + '/biases']
      y = ops.conv2d(layer, w, b, 1, 'SAME')      # libs.sn.convolution directly: no equalized-lr input scaling
    else:
      y = _ge_conv(P, scope, layer, domain, cfg, k=1, activation=False, pixel_norm=False, equalize=False, cond=cond,
                   spectral=False)
    outs.append(ops.tanh(y) if nm != 'sa_h' else y)
  f, g, h = outs
  npos = hh * ww
  f, g, h = f.reshape(n, npos, -1), g.reshape(n, npos, -1), h.reshape(n, npos, c)
  if ops.USE_FLASH_ATTENTION and ops.flash_attention_trainable(f, h):
    o = ops.flash_attention(f, g, h).reshape(layer.shape)      # the same three ops without the [npos, npos] map in HBM
  else:
    s = ops.bgemm(f, g, False, True)                           # tf.matmul(f, g, transpose_b=True)
    beta = ops.softmax_rows(s)
    o = ops.bgemm(beta, h, False, False).reshape(layer.shape)
  return ops.add(ops.scale_dev(o, P[sc + '/sa_gamma']), layer)


def maybe_add_self_attention(P, top, hw, name_channels, net, end_points, domain, cfg, is_discriminator=False, cond=None):
  """nets/pggan_utils.py:301-308."""
  if cfg.do_self_attention and hw == cfg.self_attention_hw:
    name = 'self_attention_%dx%dx%d' % (hw, hw, name_channels)
    net = self_attention_layer(P, '%s/%s' % (top, name), net, domain, cfg, is_discriminator, cond)
    end_points[name] = net
  return net


def _conv_any(x, w, bias, k):
  if k == 1 and (w.shape[2] <= 4 or w.shape[3] <= 4):
    return ops.pointwise_conv(x, w, bias)
  return ops.conv2d(x, w, bias, k, 'SAME')


def maybe_resblock(P, blk, input_layer, out_channels, conv2d_out, cfg, is_discriminator=False):
  """nets/pggan_utils.py:257-264,334-342: ``shortcut + conv2d_out`` with the shortcut = the block input, or a 1x1
  conv of it (scope 'shortcut', bias, no normaliser, no activation) when the channel count changes."""
  if not cfg.use_res_block:
    return conv2d_out
  if input_layer.shape[-1] == out_channels:
    return ops.add(input_layer, conv2d_out)
  sc = _conv_any(_equalize(input_layer, cfg, 1), _sn(P, blk + '/shortcut', cfg, is_discriminator),
                 P[blk + '/shortcut/biases'], 1)
  return ops.add(sc, conv2d_out)


def _cond_rows(P, scope, ns, cond, segments):
  """Per-image normaliser parameters from a style embedding (libs/instance_norm.py:93-120; libs/batch_norm.py:34-38):
  gamma = 1 + FC(cond; '<scope>/<norm>/gamma_<d>'), beta = FC(cond; '.../beta_<d>') with the FC of each image's domain.
  ``segments``: [(domain, lo, hi)] over the batch.  Returns ([n, C], [n, C]) fp32, differentiable."""
  import torch
  gs, bs = [], []
  for d, lo, hi in segments:
    c = cond[lo:hi].contiguous()
    gn, bn = norm_var(scope, ns, 'gamma', d), norm_var(scope, ns, 'beta', d)
    gs.append(ops.fully_connected(c, P[gn + '/weights'], P[gn + '/biases']) + 1.0)
    bs.append(ops.fully_connected(c, P[bn + '/weights'], P[bn + '/biases']))
  if len(gs) == 1:
    return gs[0], bs[0]
  return torch.cat(gs), torch.cat(bs)


def _ge_conv(P, scope, x, domain, cfg, k=3, padding='SAME', activation=True, pixel_norm=True, pool=False, equalize=True,
             upcat=None, cond=None, spectral=True, latent=None):
  """maybe_pixel_norm(maybe_equalized_conv2d(...)) for the generator / encoder arg-scope
  (nets/pggan_utils.py:86-98,236-245): conv without bias, per-domain instance norm, LeakyReLU(0.2),
  then pixel norm (nets/pggan.py:78-81).  ``domain`` is 's' | 't', or (d0, d1, split): the batch holds
  ``split`` images of domain d0 followed by images of domain d1 (two reference passes as one launch)."""
  # spectral=False: the attention convs go through libs.sn.convolution with do_spec_norm False (libs/self_attention.py:
  # 30-48): never normalised, whatever --spectral_norm_in_non_discriminator says
  w = _sn(P, scope, cfg, False) if spectral else P[scope + '/weights']
  if cfg.generator_norm_type == 'none':
    if upcat is not None:
      x = ops.upsample2x_concat(upcat[0], upcat[1], upcat[2], upcat[3])
    xe = _equalize(x, cfg, k) if equalize else x
    b = P[scope + '/biases']
    if k == 1 and (w.shape[2] <= 4 or w.shape[3] <= 4):
      y = ops.pointwise_conv(xe, w, b, lrelu=activation)
    else:
      y = ops.conv2d(xe, w, b, k, padding, lrelu=activation)
    if pixel_norm and cfg.do_pixel_norm:
      y = ops.pixel_norm(y)
    return (y, ops.avg_pool2(y)) if pool else y
  # instance norm without conditioning: the conv's epilogue sums its outputs per workgroup (ConvStats) and the
  # normaliser reads those sums instead of the tensor (tg_conv2d_fwd_stats -> tg_norm_act_fwd_conv_stats)
  want_stats = cfg.generator_norm_type == 'instance_norm' and cond is None
  cst = None
  if upcat is not None:      # x is None: the input is concat(up2(x0), skip), read from the two sources by the conv
    if want_stats:
      y, cst = ops.upcat_conv_stats(upcat[0], upcat[1], w, upcat[2], upcat[3])
    else:
      y = ops.upcat_conv(upcat[0], upcat[1], w, upcat[2], upcat[3])
  elif latent is not None and latent.dtype in ops.HALF_TYPES:
    # x is the [B, 1, 1, C] noise ``latent`` zero-padded for a k x k VALID conv (nets/pggan.py:135-153): the GEMM form
    y = ops.latent_conv(_equalize(latent, cfg, k, in_ch=latent.shape[-1]) if equalize else latent, w)
  elif k == 1 and (w.shape[2] <= 4 or w.shape[3] <= 4):
    y = ops.pointwise_conv(_equalize(x, cfg, k) if equalize else x, w)
  elif want_stats:
    y, cst = ops.conv2d_stats(_equalize(x, cfg, k) if equalize else x, w, k, padding)
  else:
    y = ops.conv2d(_equalize(x, cfg, k) if equalize else x, w, None, k, padding)
  nt = cfg.generator_norm_type
  if nt == 'none':
    # no normaliser (nets/pggan_utils.py:198-200): slim's conv2d then owns a bias; conv + bias + LeakyReLU is the
    # conv's epilogue, the pixel norm the fused kernel with constant statistics
    raise AssertionError('unreachable: generator_norm_type none is handled before the conv')
  if nt not in NORM_SCOPE:
    raise NotImplementedError('unsupported norm type: %s' % nt)      # nets/pggan_utils.py:201-202
  ns = NORM_SCOPE[nt]
  if ns == NATIVE_NORM:      # tf.contrib's own layers (nets/pggan_utils.py:175-197)
    assert cond is None, 'Tensorflow implementation does not support `conditional_layer`.'      # pggan_utils.py:177,191
  renorm = nt in ('batch_renorm', 'batch_renorm_native')
  if isinstance(domain, tuple):
    d0, d1, split = domain[:3]
    passes = domain[3] if len(domain) > 3 else 2
  else:
    d0, d1, split, passes = domain, None, None, 1
  pn = pixel_norm and cfg.do_pixel_norm
  if not cfg.is_training and nt not in ('instance_norm', 'layer_norm_native'):      # those two keep no statistics
    return _batch_norm_inference(P, scope, y, d0, d1, split, cond, activation, pn, pool, ns)
  if cond is not None:      # conditional norm: one parameter row per image (twingan.py:245-267)
    n_ = y.shape[0]
    segs = [(d0, 0, n_)] if d1 is None else [(d0, 0, split), (d1, split, n_)]
    if nt == 'instance_norm':
      g_rows, b_rows = _cond_rows(P, scope, ns, cond, segs)
      return ops.norm_act(y, g_rows, b_rows, lrelu=activation, pixel_norm=pn, pool=pool, stats=ops.instance_stats(y, 1e-6))
    if nt == 'batch_renorm':      # libs/batch_norm.py:209-259 with a conditional layer: r / d per pass, gamma / beta per image
      cn = cond.float()
      cn = cn / cn.pow(2).sum(dim=1, keepdim=True).clamp_min(1e-12).sqrt()
      rows = _cond_rows(P, scope, ns, cn, segs)
      n, h, w, c = y.shape
      return _batch_renorm(P, scope, y.view(passes, (n // passes) * h, w, c),
                           (d0, d1, None if split is None else split * passes // n, passes), activation, pn, pool,
                           cond_rows=rows, image_shape=(n, h, w, c))
    return _cond_batch_norm(P, scope, y, cond, segs, passes, activation, pn, pool, cfg)
  g0, b0 = P[norm_var(scope, ns, 'gamma', d0)], P[norm_var(scope, ns, 'beta', d0)]
  g1 = P[norm_var(scope, ns, 'gamma', d1)] if d1 else None
  b1 = P[norm_var(scope, ns, 'beta', d1)] if d1 else None
  if nt == 'instance_norm':
    return ops.norm_act(y, g0, b0, lrelu=activation, pixel_norm=pn, gamma2=g1, beta2=b1, split=split, pool=pool,
                        conv_stats=cst)
  if nt == 'layer_norm_native':
    # tf.contrib.layers.layer_norm(center, scale, scope=<postfix>) (nets/pggan_utils.py:189-197): statistics of one IMAGE
    # over (H, W, C), gamma / beta per channel, variance_epsilon 1e-12
    return ops.layer_norm_act(y, g0, b0, lrelu=activation, pixel_norm=pn, gamma2=g1, beta2=b1, split=split, pool=pool)
  # batch norm (libs/batch_norm.py:42-326, training mode): moments over (N,H,W) of ONE reference pass.  Each of
  # the `passes` batched along N is a statistic group: the [passes, B*H, W, C] view turns the per-image
  # kernels into per-pass ones (pixel norm and pooling are per pixel / per 2x2 block, which the view preserves).
  n, h, w, c = y.shape
  assert n % passes == 0 and (split is None or (split * passes) % n == 0)
  yv = y.view(passes, (n // passes) * h, w, c)
  st = P.state if hasattr(P, 'state') else None
  if renorm:
    out = _batch_renorm(P, scope, yv, (d0, d1, None if split is None else split * passes // n, passes), activation, pn,
                        pool, ns=ns)
    if pool:
      return out[0].view(n, h, w, c), out[1].view(n, h // 2, w // 2, c)
    return out.view(n, h, w, c)
  ema = None
  if st is not None:
    pairs = [(st['%s/BatchNorm/moving_mean%s' % (scope, _pf(d))], st['%s/BatchNorm/moving_variance%s' % (scope, _pf(d))])
             for d in ((d0, d1) if d1 else (d0,))]
    ema = (0.999, pairs)
  out = ops.norm_act(yv, g0, b0, lrelu=activation, pixel_norm=pn, in_eps=1e-3, gamma2=g1, beta2=b1,
                     split=None if split is None else split * passes // n, pool=pool, ema=ema)
  if pool:
    return out[0].view(n, h, w, c), out[1].view(n, h // 2, w // 2, c)
  return out.view(n, h, w, c)


def _cond_batch_norm(P, scope, y, cond, segs, passes, activation, pn, pool, cfg):
  """conditional_batch_norm with a conditional layer (libs/batch_norm.py:82-85,152-159,403-470): the embedding is
  l2-normalised per image, gamma = 1 + FC, beta = FC give one row per image, applied to activations normalised with
  the statistics of the whole reference pass.  An option row, not on the headline path: the batch statistics and
  their backward are the HIP normaliser (per-pass view, unit gamma, zero beta, no activation); the per-image affine,
  LeakyReLU and pixel norm that follow are the same fused kernel in its per-image-row mode with constant statistics
  (ops.affine_act)."""
  import torch
  n, h, w, c = y.shape
  cn = cond.float()
  cn = cn / cn.pow(2).sum(dim=1, keepdim=True).clamp_min(1e-12).sqrt()
  g_rows, b_rows = _cond_rows(P, scope, 'BatchNorm', cn, segs)
  one = torch.ones(c, dtype=torch.float32, device=y.device)
  zero = torch.zeros(c, dtype=torch.float32, device=y.device)
  ema = None
  st = P.state if hasattr(P, 'state') else None
  if st is not None:
    doms = [d for d, _, _ in segs]
    ema = (0.999, [(st['%s/BatchNorm/moving_mean%s' % (scope, _pf(d))], st['%s/BatchNorm/moving_variance%s' % (scope, _pf(d))])
                   for d in doms])
  split_v = None if len(segs) == 1 else segs[0][2] * passes // n
  yhat = ops.norm_act(y.view(passes, (n // passes) * h, w, c), one, zero, lrelu=False, pixel_norm=False, in_eps=BN_EPS,
                      gamma2=one if split_v is not None else None, beta2=zero if split_v is not None else None,
                      split=split_v, ema=ema).view(n, h, w, c)
  # the per-image affine, LeakyReLU, pixel norm (and pool) after it: the fused kernel with constant statistics
  return ops.affine_act(yhat, g_rows, b_rows, lrelu=activation, pixel_norm=pn, pool=pool)


def _batch_norm_inference(P, scope, y, d0, d1, split, cond, activation, pn, pool, ns='BatchNorm'):
  """conditional_batch_norm(is_training=False), with or without renorm (libs/batch_norm.py:403-470): normalise with the
  domain's MOVING mean / variance -- one (mean, rstd, gamma, beta) row per image through the fused kernel's
  per-image-row mode.  ``cond``: the l2-normalised embedding gives gamma = 1 + FC, beta = FC per image."""
  import torch
  n, h, w, c = y.shape
  segs = [(d0, 0, n)] if d1 is None else [(d0, 0, split), (d1, split, n)]
  st = P.state
  nm = lambda v, d: norm_var(scope, ns, v, d)
  mean = torch.cat([st[nm('moving_mean', d)].expand(hi - lo, c) for d, lo, hi in segs])
  rstd = torch.cat([torch.rsqrt(st[nm('moving_variance', d)] + BN_EPS).expand(hi - lo, c) for d, lo, hi in segs])
  if cond is not None:
    cn = cond.float()
    cn = cn / cn.pow(2).sum(dim=1, keepdim=True).clamp_min(1e-12).sqrt()
    g_rows, b_rows = _cond_rows(P, scope, ns, cn, segs)
  else:
    g_rows = torch.cat([P[nm('gamma', d)].expand(hi - lo, c) for d, lo, hi in segs])
    b_rows = torch.cat([P[nm('beta', d)].expand(hi - lo, c) for d, lo, hi in segs])
  return ops.norm_act(y, g_rows.contiguous(), b_rows.contiguous(), lrelu=activation, pixel_norm=pn, in_eps=BN_EPS, pool=pool,
                      stats=(mean.contiguous().reshape(-1), rstd.contiguous().reshape(-1)))


BN_EPS = 1e-3            # libs/batch_norm.py:48
RENORM_MOMENTUM = 0.99   # libs/batch_norm.py:62; the moving averages use the same decay (nets/pggan_utils.py:163)


def _batch_renorm(P, scope, yv, domain, activation, pn, pool, cond_rows=None, image_shape=None, ns='BatchNorm'):
  """conditional_batch_norm(renorm=True) in training mode (libs/batch_norm.py:209-246,329-470), for yv =
  [passes, B*H, W, C] where pass i belongs to domain d0 (i < split) or d1.  Per pass, in call order:
    stddev = sqrt(var + eps);  r = clip(stddev / mixed_stddev),  d = clip((mean - mixed_mean) / mixed_stddev)
    with mixed_x = renorm_x + (1 - renorm_x_weight) * batch_x  (pre-update values, stop-gradient),
    out = x_hat * (r * gamma) + (d * gamma + beta);  then renorm_mean / renorm_stddev (+ weights) and the moving
    mean / variance (of the unbiased renorm values) are updated with momentum 0.99.
  The reference applies the passes' update ops in unspecified order within one session.run; here (and in the
  oracle) passes update the state sequentially.  The per-channel state arithmetic is a handful of tiny tensor ops;
  the normalisation itself is the fused kernel with one parameter row per pass.  ``cond_rows`` = (gamma [n, c], beta
  [n, c]) of a conditional layer: the passes are normalised with unit rows and the per-image rows r * gamma,
  d * gamma + beta are applied by ops.affine_act (image_shape = the [n, h, w, c] shape behind the per-pass view)."""
  import torch
  d0, d1, split, passes = domain
  st = P.state
  mean, rstd = ops.instance_stats(yv, BN_EPS)
  c = yv.shape[3]
  mean_p, rstd_p = mean.view(passes, c), rstd.view(passes, c)
  rmax, rmin, dmax = st['renorm/rmax'], st['renorm/rmin'], st['renorm/dmax']
  g_rows, b_rows = [], []
  m = RENORM_MOMENTUM
  for i in range(passes):
    d = d0 if (split is None or i < split) else d1
    nm = lambda v: norm_var(scope, ns, v, d)
    if cond_rows is None:
      gamma, beta = P[nm('gamma')], P[nm('beta')]
    else:      # this pass' images
      per = image_shape[0] // passes
      gamma, beta = cond_rows[0][i * per:(i + 1) * per], cond_rows[1][i * per:(i + 1) * per]
    with torch.no_grad():
      bmean, stddev = mean_p[i], 1.0 / rstd_p[i]
      rm, rmw = st[nm('renorm_mean')], st[nm('renorm_mean_weight')]
      rs, rsw = st[nm('renorm_stddev')], st[nm('renorm_stddev_weight')]
      mixed_mean = rm + (1.0 - rmw) * bmean
      mixed_std = rs + (1.0 - rsw) * stddev
      r = torch.minimum(torch.maximum(stddev / mixed_std, rmin), rmax)
      dd = torch.minimum(torch.maximum((bmean - mixed_mean) / mixed_std, -dmax), dmax)
      rm.mul_(m).add_(bmean, alpha=1.0 - m)
      rmw.mul_(m).add_(1.0 - m)
      rs.mul_(m).add_(stddev, alpha=1.0 - m)
      rsw.mul_(m).add_(1.0 - m)
      new_mean, new_std = rm / rmw, rs / rsw
      st[nm('moving_mean')].mul_(m).add_(new_mean, alpha=1.0 - m)
      st[nm('moving_variance')].mul_(m).add_(new_std * new_std - BN_EPS, alpha=1.0 - m)
    g_rows.append(r * gamma)
    b_rows.append(dd * gamma + beta)
  if cond_rows is not None:
    one = torch.ones(passes, c, dtype=torch.float32, device=yv.device)
    yhat = ops.norm_act(yv, one, torch.zeros_like(one), lrelu=False, pixel_norm=False, in_eps=BN_EPS, stats=(mean, rstd))
    return ops.affine_act(yhat.view(image_shape), torch.cat(g_rows), torch.cat(b_rows), lrelu=activation, pixel_norm=pn,
                          pool=pool)
  return ops.norm_act(yv, torch.stack(g_rows), torch.stack(b_rows), lrelu=activation, pixel_norm=pn, in_eps=BN_EPS,
                      pool=pool, stats=(mean, rstd))


def maybe_gdrop(P, x, cfg, in_ch=None):
  """nets/pggan.py:351-355: ``ops.gdrop(layer, mode='prop', strength=gdrop_strength)`` when do_dgrop and is_training and
  the strength is set -- the `gdrop_strength` variable under --use_gdrop (twingan.py:861-865), else the constant of the
  signature (nets/pggan.py:341).  No trainer of the reference passes do_dgrop=True, so by default this is the identity
  there and here; -> (tensor, whether the layer ran)."""
  if not (cfg.do_dgrop and cfg.is_training):
    return x, False
  strength = P.state['gdrop_strength'] if cfg.use_gdrop else cfg.gdrop_strength
  if not isinstance(strength, float) or strength:
    return ops.gdrop(x, strength, c_logical=in_ch), True
  return x, False


def _d_conv(P, scope, x, cfg, k=3, padding='SAME', pool=False, in_ch=None, sole_consumer=False, pool_only=False, gdrop=False):
  """Discriminator arg-scope (nets/pggan_utils.py:116-127): conv + bias, no norm, LeakyReLU(0.2),
  fused into the conv epilogue.  ``gdrop``: the conv's input goes through maybe_gdrop first (the two convs of a block,
  nets/pggan.py:221-231, and the two after the minibatch stddev, :328-331)."""
  if gdrop:
    x, ran = maybe_gdrop(P, x, cfg, in_ch)
    sole_consumer = sole_consumer and not ran      # the producer's LeakyReLU output is then not this conv's input
  if scope.startswith(PAIR_TOP + '/'):      # both discriminators' layer: the stacked [2, ...] kernel and bias (ParamStore.pairs)
    w, b = P.pairs[scope + '/weights'], P.pairs[scope + '/biases']
  else:
    w = _sn(P, scope, cfg, True)
    b = P[scope + '/biases']
  x = _equalize(x, cfg, k, in_ch)      # in_ch: logical channel count when x is channel-padded (minibatch stddev)
  if k == 1 and (w.shape[2] <= 4 or w.shape[3] <= 4):
    return ops.pointwise_conv(x, w, b, lrelu=True)
  # sole_consumer: nothing else reads x, so (when x is the previous conv's LeakyReLU output and no input scaling sits
  # in between) that layer's LeakyReLU backward is folded into this conv's backward-data
  fuse = sole_consumer and not cfg.equalized_learning_rate and not cfg.use_res_block
  return ops.conv2d(x, w, b, k, padding, lrelu=True, pool=pool, fuse_input_lrelu=fuse, pool_only=pool_only)


def resize_twice_as_big(x):
  """nets/pggan_utils.py:349-350."""
  return ops.upsample2x_concat(x, None)


def maybe_concat_unet_layer(layer_hw, unet_end_points, max_ch, max_concat_hw=None):
  """nets/pggan_utils.py:281-298: pick the encoder end-point to concatenate at resolution hw (none above
  --pggan_unet_max_concat_hw)."""
  if unet_end_points is None or (max_concat_hw and layer_hw > max_concat_hw):
    return None
  num_channels = get_num_channels(max_stage_of(layer_hw) - 1, max_ch)
  name = 'encoder_block_interpolated_%dx%dx%d' % (layer_hw, layer_hw, num_channels)
  if name not in unet_end_points:
    name = 'encoder_block_%dx%dx%d' % (layer_hw, layer_hw, num_channels)
  if name not in unet_end_points:
    raise ValueError('%s not in unet_end_points' % name)
  return unet_end_points[name]


# ------------------------------------------------------------------------------------------------
# encoder (nets/pggan.py:382-479)
# ------------------------------------------------------------------------------------------------
def encoder_before_classification(P, source, domain, cfg, top='encoder_content', cuts=None):
  """``cuts`` = (low segment, high segment) of a segmented backward (ops.Cuts; no-ops unless the trainer opened one):
  the UNet skip end-points become leaves, and so does the tensor entering the first block at cfg.overlap_cut_hw --
  blocks above that resolution (whose backward runs last) resume in the high segment."""
  hw = source.shape[1]
  max_stage = max_stage_of(hw)
  assert max_stage >= 0
  end_points = {'source': source}
  shrinked = None
  if cfg.is_growing:
    pooled = ops.avg_pool2(source)
    name = 'from_rgb_%dx%d' % (hw // 2, hw // 2)
    shrinked = _ge_conv(P, '%s/%s/Conv' % (top, name), pooled, domain, cfg, k=1)
    shrinked = maybe_resblock(P, '%s/%s' % (top, name), pooled, shrinked.shape[-1], shrinked, cfg)
    end_points[name] = shrinked
  name = 'from_rgb_%dx%d' % (hw, hw)
  net = _ge_conv(P, '%s/%s/Conv' % (top, name), source, domain, cfg, k=1)
  net = maybe_resblock(P, '%s/%s' % (top, name), source, net.shape[-1], net, cfg)
  end_points[name] = net
  for stage in range(max_stage, 0, -1):
    num_channels = get_num_channels(stage - 1, cfg.max_ch)
    current_hw = hw // (2 ** (max_stage - stage))
    if cuts is not None and current_hw == cfg.overlap_cut_hw and current_hw < hw:
      net = ops.Cuts.cut(net, cuts[1])
    net = maybe_add_self_attention(P, top, current_hw, num_channels, net, end_points, domain, cfg)
    name = 'encoder_block_%dx%dx%d' % (current_hw, current_hw, num_channels)
    block_in = net
    net = _ge_conv(P, '%s/%s/Conv' % (top, name), net, domain, cfg)
    if cfg.use_res_block:
      net = _ge_conv(P, '%s/%s/Conv_1' % (top, name), net, domain, cfg)
      end_points[name] = maybe_resblock(P, '%s/%s' % (top, name), block_in, num_channels, net, cfg)
      net = ops.avg_pool2(end_points[name])
    else:
      # last layer of the block + tf.nn.avg_pool (nets/pggan.py:466-468) as one op: (skip end-point, pooled)
      end_points[name], net = _ge_conv(P, '%s/%s/Conv_1' % (top, name), net, domain, cfg, pool=True)
    if cuts is not None:      # the skip end-point is read by the generator only
      end_points[name] = ops.Cuts.cut(end_points[name], cuts[1] if current_hw > cfg.overlap_cut_hw else cuts[0])
    current_hw //= 2
    end_points['downsample_to_%dx%dx%d' % (current_hw, current_hw, num_channels)] = net
    if stage == max_stage and cfg.is_growing:
      net = ops.lerp(net, shrinked, cfg.alpha_grow)
      # the generator reads this end-point as its UNet skip at hw / 2: under a segmented backward it is a leaf of the HIGH
      # segment whatever its own resolution -- its producers are the full-resolution block and the shrink path
      end_points['encoder_block_interpolated_%dx%dx%d' % (current_hw, current_hw, num_channels)] = \
          net if cuts is None else ops.Cuts.cut(net, cuts[1])
  end_points['before_classification'] = net
  return net, end_points


def encoder_classification(P, net, domain, cfg, top):
  """nets/pggan.py:482-507: conv3x3 SAME -> conv4x4 VALID (generator arg-scope: norm + LeakyReLU, no pixel norm) -> FC.
  The 4x4 VALID output is 1x1, so its instance norm sees one pixel (variance 0): the layer outputs lrelu(beta), as in
  the reference."""
  blk = '%s/before_fc_1x1x%d' % (top, cfg.max_ch)
  end_points = {}
  net = _ge_conv(P, blk + '/Conv', net, domain, cfg, pixel_norm=False)
  net = _ge_conv(P, blk + '/Conv_1', net, domain, cfg, k=4, padding='VALID', pixel_norm=False)
  end_points['before_fc_1x1x%d' % cfg.max_ch] = net
  feat = net.reshape(net.shape[0], -1)
  pred = ops.fully_connected(_equalize(feat, cfg, 1), P[top + '/prediction/fully_connected/weights'],
                             P[top + '/prediction/fully_connected/biases'])
  end_points['prediction'] = pred
  return pred, end_points


def encoder(P, source, domain, cfg, top='encoder_style'):
  """nets/pggan.py:510-541: the full encoder -> [B, output_dim] (the style encoder of twingan.py:201-223)."""
  net, end_points = encoder_before_classification(P, source, domain, cfg, top)
  pred, ep = encoder_classification(P, net, domain, cfg, top)
  end_points.update(ep)
  return pred, end_points


# ------------------------------------------------------------------------------------------------
# generator (nets/pggan.py:69-211), TwinGAN mode: source is the encoder's [B,4,4,C] content tensor
# ------------------------------------------------------------------------------------------------
def rgb_kernel_size(cfg, hw):
  """--use_larger_filter_at_rgb_layer (nets/pggan.py:172-175,194-197): to-RGB kernel min(7, hw / 2) instead of 1x1.
  The reference computes it from the CURRENT block's hw for both the grown and the previous-resolution layer."""
  return min(7, hw // 2) if cfg.use_larger_filter_at_rgb_layer else 1


def get_noise_shape(batch_size=None, max_num_channels=256):
  """nets/pggan.py:86-90: the latent noise of the plain PGGAN generator, [B, 1, 1, get_num_channels(1)]."""
  return (batch_size, 1, 1, get_num_channels(1, max_num_channels))


def generator(P, source, domain, cfg, unet_end_points=None, top='generator', unet_groups=None, cond=None):
  """``source``: the encoder's [B,4,4,C] content tensor (TwinGAN), or latent noise [B,1,1,C] / [B,C] (plain PGGAN,
  nets/pggan.py:135-153: zero-padded to 7x7 so that the block's first conv, 4x4 VALID, yields the 4x4 map)."""
  import torch
  max_stage = max_stage_of(cfg.hw)
  if source.dim() == 2:
    source = source.reshape(source.shape[0], 1, 1, source.shape[1])
  noise_mode = source.shape[1] == 1 and source.shape[2] == 1
  noise = source if noise_mode else None
  if noise_mode:
    source = torch.nn.functional.pad(source, (0, 0, 3, 3, 3, 3)).contiguous()
  else:
    assert source.shape[1] == 4 and source.shape[2] == 4, 'the generator takes a 4x4 content tensor or 1x1 noise'
  end_points = {'source': source}
  net = source
  net_before_growth = None
  hw = 4
  for stage in range(0, max_stage + 1):
    hw = 2 ** (stage + 2)
    output_channels = get_num_channels(stage, cfg.max_ch)
    name = 'block_%dx%dx%d' % (hw, hw, output_channels)
    if hw == 4:
      if noise_mode:
        net = _ge_conv(P, '%s/%s/Conv' % (top, name), net, domain, cfg, k=4, padding='VALID', cond=cond, latent=noise)
      else:
        net = _ge_conv(P, '%s/%s/Conv' % (top, name), net, domain, cfg, cond=cond)
      net = _ge_conv(P, '%s/%s/Conv_1' % (top, name), net, domain, cfg, cond=cond)
    else:
      if stage == max_stage and cfg.is_growing:
        rgb = 'generator_to_rgb_%dx%d' % (hw // 2, hw // 2)
        net_before_growth = _ge_conv(P, '%s/%s/Conv' % (top, rgb), net, domain, cfg, k=rgb_kernel_size(cfg, hw),
                                     activation=False, pixel_norm=False, cond=cond)
        net_before_growth = resize_twice_as_big(net_before_growth)
        end_points[rgb] = net_before_growth
      # generator_three_layer_block: upsample -> concat(UNet) -> conv -> conv  (pggan.py:69-83)
      # unet_groups = (gsz, perm): batched passes read the skip tensors of the encoder batch by group permutation
      skip = maybe_concat_unet_layer(hw, unet_end_points, cfg.max_ch, cfg.unet_max_concat_hw)
      gsz, perm = unet_groups if (skip is not None and unet_groups is not None) else (0, ())
      w0 = P['%s/%s/Conv/weights' % (top, name)]
      if not (cfg.use_res_block or cfg.equalized_learning_rate) and ops.upcat_conv_supported(net, skip, w0):
        # resize_twice_as_big + maybe_concat_unet_layer + the block's first conv as one op: no concat tensor
        block_in = None
        net = _ge_conv(P, '%s/%s/Conv' % (top, name), None, domain, cfg, upcat=(net, skip, gsz, perm), cond=cond)
      else:
        net = ops.upsample2x_concat(net, skip, gsz, perm)
        block_in = net
        net = _ge_conv(P, '%s/%s/Conv' % (top, name), net, domain, cfg, cond=cond)
      net = _ge_conv(P, '%s/%s/Conv_1' % (top, name), net, domain, cfg, cond=cond)
      net = maybe_resblock(P, '%s/%s' % (top, name), block_in, output_channels, net, cfg)
    end_points[name] = net
    net = maybe_add_self_attention(P, top, hw, output_channels, net, end_points, domain, cfg, cond=cond)      # pggan.py:188-190
  rgb = 'generator_to_rgb_%dx%d' % (hw, hw)
  # to_rgb: activation None, normaliser still applied, no pixel norm (pggan.py:192-200)
  to_rgb = _ge_conv(P, '%s/%s/Conv' % (top, rgb), net, domain, cfg, k=rgb_kernel_size(cfg, hw), activation=False,
                    pixel_norm=False, cond=cond)
  if cfg.is_growing:
    output = ops.lerp(to_rgb, net_before_growth, cfg.alpha_grow)
    end_points['alpha_grow'] = cfg.alpha_grow
  else:
    output = to_rgb
  end_points['output'] = output
  return output, end_points


# ------------------------------------------------------------------------------------------------
# discriminator (nets/pggan.py:217-376)
# ------------------------------------------------------------------------------------------------
PAIR_TOP = 'discriminator_*'      # scope of the stacked twin variables of discriminator_s / discriminator_t (ParamStore.pairs)
PAIR_HW = int(__import__('os').environ.get('TG_PAIR_HW', '32'))      # the two discriminators run as ONE grouped launch per layer from this resolution down


def discriminator_before_fc(P, source, cfg, top, groups=1, cut_seg=None, block_end_points=True, until_hw=None, from_hw=None,
                            full_hw=None):
  """``cut_seg``: segment of a segmented backward (ops.Cuts) in which the blocks above cfg.overlap_cut_hw resume.
  ``block_end_points=False``: the caller wants the prediction only -- the full-resolution output of a block's last conv
  (end_points['encoder_block_*'], which the reference overwrites with its pooled version as `net`, nets/pggan.py:304-306)
  is then not materialised where the conv can hand the pool and the LeakyReLU sign bits over directly.
  ``until_hw``: only the HEAD of the network -- from_rgb and the blocks above that resolution; returns the tensor that
  enters the block at ``until_hw``.  ``from_hw`` (+ ``full_hw``, the network's input resolution): only the TAIL -- ``source``
  is such a tensor, the blocks from ``from_hw`` down, the minibatch stddev and the two last convs follow (discriminator_pair
  runs the heads per domain and the tail of both discriminators as one batch with ``top`` = PAIR_TOP)."""
  hw = full_hw or source.shape[1]
  max_stage = max_stage_of(hw)
  assert max_stage >= 0
  max_ch = cfg.max_ch_dis or cfg.max_ch      # get_discriminator_max_num_channels (nets/pggan_utils.py:375-380)
  end_points = {}
  shrinked = None
  net = source
  if from_hw is None:
    if cfg.is_growing:
      pooled = ops.avg_pool2(source)
      name = 'from_rgb_%dx%d' % (hw // 2, hw // 2)
      shrinked = _d_conv(P, '%s/%s/Conv' % (top, name), pooled, cfg, k=1)
      shrinked = maybe_resblock(P, '%s/%s' % (top, name), pooled, shrinked.shape[-1], shrinked, cfg, True)
      end_points[name] = shrinked
    name = 'from_rgb_%dx%d' % (hw, hw)
    net = _d_conv(P, '%s/%s/Conv' % (top, name), source, cfg, k=1)
    net = maybe_resblock(P, '%s/%s' % (top, name), source, net.shape[-1], net, cfg, True)
    end_points[name] = net
  for stage in range(max_stage, 0, -1):
    num_channels = get_num_channels(stage - 1, max_ch)
    current_hw = hw // (2 ** (max_stage - stage))
    if from_hw is not None and current_hw > from_hw:
      continue
    if until_hw is not None and current_hw <= until_hw:
      return net, end_points
    if cut_seg is not None and current_hw == cfg.overlap_cut_hw and current_hw < hw:
      net = ops.Cuts.cut(net, cut_seg)
    net = maybe_add_self_attention(P, top, current_hw, num_channels, net, end_points, None, cfg, True)   # pggan.py:294-296
    name = 'encoder_block_%dx%dx%d' % (current_hw, current_hw, num_channels)
    block_in = net
    net = _d_conv(P, '%s/%s/Conv' % (top, name), net, cfg, sole_consumer=True, gdrop=True)
    if cfg.use_res_block:
      net = _d_conv(P, '%s/%s/Conv_1' % (top, name), net, cfg, gdrop=True)
      end_points[name] = maybe_resblock(P, '%s/%s' % (top, name), block_in, num_channels, net, cfg, True)
      net = ops.avg_pool2(end_points[name])
    else:
      full, net = _d_conv(P, '%s/%s/Conv_1' % (top, name), net, cfg, pool=True, sole_consumer=True,      # conv + avg_pool (pggan.py:304-306)
                          pool_only=not block_end_points, gdrop=True)
      if full is not None:
        end_points[name] = full
    current_hw //= 2
    end_points['downsample_to_%dx%dx%d' % (current_hw, current_hw, num_channels)] = net
    if stage == max_stage and cfg.is_growing:
      net = ops.lerp(net, shrinked, cfg.alpha_grow)
      end_points['encoder_block_interpolated_%dx%dx%d' % (current_hw, current_hw, num_channels)] = net
  if until_hw is not None:
    return net, end_points
  blk = 'before_fc_1x1x%d' % max_ch
  net = ops.minibatch_state_concat(net, mbstd_cpad(net.shape[3]), groups)      # pggan_utils.py:353-366
  net = _d_conv(P, '%s/%s/Conv' % (top, blk), net, cfg, k=3, padding='SAME', in_ch=max_ch + 1, gdrop=True)
  net = _d_conv(P, '%s/%s/Conv_1' % (top, blk), net, cfg, k=4, padding='VALID', sole_consumer=True, gdrop=True)
  end_points[blk] = net
  end_points['before_fc'] = net
  return net, end_points


def discriminator(P, source, cfg, top, groups=1, cut_seg=None, block_end_points=True):
  """``groups`` > 1: ``source`` is that many discriminator calls batched along N (each keeps its own
  minibatch-stddev statistic, as separate reference calls would)."""
  net, end_points = discriminator_before_fc(P, source, cfg, top, groups, cut_seg, block_end_points)
  feat = net.reshape(net.shape[0], -1)                                 # tf.squeeze(net, (1, 2))
  pred = ops.fully_connected(_equalize(feat, cfg, 1), P[top + '/prediction/fully_connected/weights'],
                             P[top + '/prediction/fully_connected/biases'])
  end_points['prediction'] = pred
  return pred, end_points


# The two discriminators as ONE batch through grouped convs from PAIR_HW down.  OFF by default -- measured on MI355X
# (profiles/r06_b_ab_discriminator_pair.txt, config 3, ms per step): two independent streams, no pair 15.67; pair from 32 / 16 /
# 8 / 4 down 16.59 / 16.47 / 16.43 / 16.47; pair without streams 16.60; neither 17.29.  Grouping does shorten the serial
# chain (17.29 -> 16.6), but two streams that never meet hide more (-> 15.67), and the pair needs a join per pass, which a
# replayed hipGraph pays for dearly.  TG_D_PAIR=1 (or the tests) turns it on; the grouped kernels are exact either way.
USE_DISCRIMINATOR_PAIR = __import__('os').environ.get('TG_D_PAIR', '0') == '1'


def discriminator_pair_supported(P, cfg, hw):
  """Can discriminator_s and discriminator_t run their layers at <= PAIR_HW as grouped launches?  The plain tower only:
  every option that reads more than (kernel, bias) per layer keeps the per-domain path."""
  return bool(USE_DISCRIMINATOR_PAIR and getattr(P, 'pairs', None) and hw > PAIR_HW and not cfg.spectral_norm
              and not cfg.equalized_learning_rate and not cfg.use_res_block and not (cfg.do_dgrop and cfg.is_training)
              and not (cfg.do_self_attention and cfg.self_attention_hw <= PAIR_HW)
              and not (cfg.is_growing and hw // 2 <= PAIR_HW))


def discriminator_pair(P, source_s, source_t, cfg, groups=1, cut_seg=None, streams=None, meanwhile=None):
  """discriminator(source_s; 'discriminator_s') and discriminator(source_t; 'discriminator_t') -> (pred_s, pred_t).

  The reference builds the two discriminators as two towers of identical layers over different variables
  (twingan.py:105-110; image_generation.py:348-439 once per domain).  Above PAIR_HW each runs on its own stream (kernels
  that fill the chip); from PAIR_HW down -- launches of a few workgroups each, where the step is bound by the length of
  the dependent launch chain -- both run as ONE batch [D_s rows; D_t rows] through grouped convs (TgConvDesc.groups = 2:
  the kernel picks the weight set from the image index), i.e. half the launches.  Every image sees exactly the arithmetic
  of its own tower: the minibatch-stddev groups stay per call (2 x ``groups``), the convs are per image.
  ``meanwhile``: work for the MAIN stream that is independent of the discriminators (the generator step's re-encoding pass),
  enqueued after the heads were forked onto the domain streams and before the main stream joins them for the tail.  (The
  tail runs on the main stream: a wait edge between the two domain streams made hipStreamEndCapture fault, ROCm 7.2.)"""
  hw = source_s.shape[1]
  if not discriminator_pair_supported(P, cfg, hw):
    preds = []
    for i, (d, src) in enumerate((('s', source_s), ('t', source_t))):
      with (streams.domain(i) if streams is not None else _null()):
        preds.append(discriminator(P, src, cfg, 'discriminator_' + d, groups, cut_seg, False)[0])
    if meanwhile is not None:
      meanwhile()
    return preds[0], preds[1]
  heads = []
  for i, (d, src) in enumerate((('s', source_s), ('t', source_t))):
    with (streams.domain(i) if streams is not None else _null()):
      heads.append(discriminator_before_fc(P, src, cfg, 'discriminator_' + d, groups, cut_seg, False, until_hw=PAIR_HW)[0])
  if meanwhile is not None:
    meanwhile()
  if streams is not None:
    streams.join()
    if heads[0].is_cuda:
      import torch
      for h in heads:
        h.record_stream(torch.cuda.current_stream(h.device))      # allocated on a domain stream, read on the main one
  return discriminator_pair_tail(P, heads[0], heads[1], cfg, hw, groups, cut_seg)


def discriminator_pair_tail(P, net_s, net_t, cfg, hw, groups=1, cut_seg=None):
  n = net_s.shape[0]
  net = ops.cat_rows([net_s, net_t])
  net, _ = discriminator_before_fc(P, net, cfg, PAIR_TOP, 2 * groups, cut_seg, False, from_hw=PAIR_HW, full_hw=hw)
  feat = net.reshape(net.shape[0], -1)                                 # tf.squeeze(net, (1, 2))
  fs, ft = ops.rows(feat, [(0, n), (n, 2 * n)])
  return tuple(ops.fully_connected(_equalize(f, cfg, 1), P['discriminator_%s/prediction/fully_connected/weights' % d],
                                   P['discriminator_%s/prediction/fully_connected/biases' % d]) for d, f in (('s', fs), ('t', ft)))


def _null():
  import contextlib
  return contextlib.nullcontext()
