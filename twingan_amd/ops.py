"""Autograd operators over the HIP kernels of libtwingan_hip.so.

This is the MI355X counterpart of the reference's op facade (libs/ops.py:28-40 and the stock TF
ops used by nets/pggan_utils.py): every function here enqueues hand-written gfx950 kernels through
the C ABI (include/twingan_hip.h) on torch's current HIP stream.  torch supplies device memory,
streams and the autograd tape only.  Every backward is itself built from these operators, so
second-order gradients (WGAN-GP, image_generation.py:414-439) flow through the same kernels.

All activations are NHWC contiguous tensors (fp32 or bf16); conv weights are fp32 HWIO masters.
"""
import ctypes
import os
import weakref

import torch

from . import _lib
from ._lib import (NF_LRELU, NF_NOSTATS, NF_PIXNORM, TG_ALGO_DIRECT, TG_ALGO_MFMA, TG_BF16, TG_EPI_BIAS, TG_EPI_LRELU, TG_F16, TG_F32,
                   TgConvDesc, call)

LRELU_ALPHA = 0.2    # util_misc.py:68


_DTYPES = {torch.float32: TG_F32, torch.bfloat16: TG_BF16, torch.float16: TG_F16}
HALF_TYPES = (torch.bfloat16, torch.float16)      # the 16-bit storage formats of the MFMA path


def _dt(t):
  try:
    return _DTYPES[t.dtype]
  except KeyError:
    raise _lib.TgError('unsupported tensor dtype %s' % t.dtype)


def _stream():
  """torch's current HIP stream of the current device; _chk() verifies that the tensors live on that device (a Trainer
  on cuda:N enters ``torch.cuda.device(N)`` around every step)."""
  return torch.cuda.current_stream().cuda_stream


def _p(t):
  return None if t is None else t.data_ptr()


def _chk(*ts):
  cur = None
  for t in ts:
    if t is not None:
      if not t.is_cuda:
        raise _lib.TgError('twingan_amd ops need CUDA/HIP tensors (no CPU fallback)')
      if cur is None:
        cur = torch.cuda.current_device()
      if t.device.index != cur:
        raise _lib.TgError('tensor on cuda:%d but the current device is cuda:%d: wrap the call in torch.cuda.device()'
                           % (t.device.index, cur))
      if not t.is_contiguous():
        raise _lib.TgError('twingan_amd ops need contiguous NHWC tensors')


# ------------------------------------------------------------------------------------------------
# weight-pack cache for the MFMA kernels
# ------------------------------------------------------------------------------------------------
# statistics of a normalised conv's output from the conv's own epilogue (USE_CONV_STATS = False (tests): separate statistics pass, for A/Bs)
USE_CONV_STATS = True
# avg_pool2 of a discriminator block's last conv written by that conv (USE_CONV_POOL = False (tests): separate pool launch)
USE_CONV_POOL = True
# ... and, where the pool is the only consumer of the full-resolution output, only the SIGN bits of that output are
# kept for the LeakyReLU backward (tg_conv2d_fwd_pool_signs / tg_lrelu_pool_bwd_signs; USE_POOL_SIGNS = False (tests): the tensor itself)
USE_POOL_SIGNS = True
# self-attention's score / softmax / value products as the flash kernels in first-order passes (USE_FLASH_ATTENTION = False (tests): the
# batched-GEMM + row-softmax composition everywhere)
USE_FLASH_ATTENTION = True
# the gradient-penalty pass through an attention layer on the flash kernels too (forward, differentiable first-order
# backward, second-order backward); USE_FLASH_BWD_BWD = False (tests): that pass keeps the batched-GEMM / softmax composition
USE_FLASH_BWD_BWD = True
# the input gradient of a generator block's first conv (upsample + UNet concat read in place) written straight into the two
# sources' gradients by the backward-data kernel (tg_conv2d_upcat_bwd_data; USE_UPCAT_BWD_FUSED = False (tests): backward-data into a
# concat-layout tensor + tg_upsample2x_concat_bwd, for A/Bs)
USE_UPCAT_BWD_FUSED = True

class PackCache:
  """bf16 K-contiguous packs (tg_conv2d_pack_weights) of registered master weights.

  Lazy mode (tests, eager steps): a pack is rebuilt in place (same device address) at its next use
  when ``version`` has moved since it was made.  Explicit mode (the trainer, and the only mode that
  is safe under hipGraph capture): the optimiser calls ``refresh(weights)`` right after every apply,
  which re-packs every existing pack of those weights and stamps it current, so a captured step
  contains its own pack launches and never depends on cache state."""
  version = 0
  _registered = {}     # data_ptr -> weight tensor
  _packs = {}          # (data_ptr, mode, elems) -> [version, buffer, descriptor copy]

  @classmethod
  def register(cls, w):
    """(Re)registers a master weight.  Packs made for a previous tensor at the same address (an earlier ParamStore whose
    memory the allocator handed out again) describe another shape and must not be refreshed from this one."""
    ptr = w.data_ptr()
    if ptr in cls._registered and cls._registered[ptr] is not w:
      for k in [k for k in cls._packs if k[0] == ptr]:
        del cls._packs[k]
      for k in [k for k in cls._tables if ptr in k[0]]:
        del cls._tables[k]
    cls._registered[ptr] = w

  @classmethod
  def clear(cls):
    cls._registered.clear()
    cls._packs.clear()
    cls._tables.clear()

  @classmethod
  def unregister(cls, w):
    """Forgets a master weight and its packs / job tables (ParamStore.close())."""
    ptr = w.data_ptr()
    if cls._registered.get(ptr) is w:
      del cls._registered[ptr]
      for k in [k for k in cls._packs if k[0] == ptr]:
        del cls._packs[k]
      for k in [k for k in cls._tables if ptr in k[0]]:
        del cls._tables[k]

  @classmethod
  def _pack(cls, w, desc, mode, buf):
    call('tg_conv2d_pack_weights', ctypes.byref(desc), _p(w), mode, _p(buf), _stream())

  @classmethod
  def get(cls, w, desc, mode):
    lib = _lib.load()
    n = lib.tg_conv2d_pack_elems(ctypes.byref(desc), mode)
    # a pack is made FOR a descriptor: its element order is the one the kernel that descriptor dispatches reads
    key = (w.data_ptr(), mode, n, desc.dtype, lib.tg_conv2d_pack_layout(ctypes.byref(desc), mode))
    cached = w.data_ptr() in cls._registered
    ent = cls._packs.get(key) if cached else None
    if ent is not None and ent[0] == cls.version:
      return ent[1]
    buf = ent[1] if ent is not None else torch.empty(
        n, dtype=torch.float16 if desc.dtype == TG_F16 else torch.bfloat16, device=w.device)
    cls._pack(w, desc, mode, buf)
    if cached:
      dcopy = TgConvDesc()
      ctypes.pointer(dcopy)[0] = desc
      cls._packs[key] = [cls.version, buf, dcopy]
    return buf

  _tables = {}         # (weight ptrs, pack keys) -> (device job table, njobs, total blocks)

  @classmethod
  def refresh(cls, weights):
    """Re-packs (in place) every existing pack of ``weights`` in ONE launch (tg_conv2d_pack_weights_multi) and
    marks them current.  The job table is built (and copied to the device) the first time a given set of packs is
    refreshed -- in the trainer that happens during the eager warm-up runs, before any graph capture."""
    ptrs = {w.data_ptr() for w in weights}
    keys = tuple(k for k in cls._packs if k[0] in ptrs)
    if not keys:
      return 0
    tkey = (tuple(sorted(ptrs)), keys)
    tab = cls._tables.get(tkey)
    if tab is None:
      lib = _lib.load()
      njobs = sum(max(int(cls._packs[k][2].groups), 1) for k in keys)
      host = ctypes.create_string_buffer(lib.tg_pack_table_bytes(njobs))
      blocks = ctypes.c_int32(0)
      j = nbytes = 0
      for k in keys:
        ent = cls._packs[k]
        g = max(int(ent[2].groups), 1)
        d1 = TgConvDesc()
        ctypes.pointer(d1)[0] = ent[2]
        d1.groups = 1
        # weight set i of a stacked kernel (params.ParamStore.pairs): its master starts i sets after the first, its pack i
        # packs after the first; one job per set
        wset, per = 4 * d1.kh * d1.kw * d1.cin * d1.cout, k[2] // g
        for i in range(g):
          call('tg_pack_table_fill', ctypes.byref(d1), k[0] + i * wset, k[1], ent[1].data_ptr() + i * per * ent[1].element_size(), j,
               ctypes.addressof(host), ctypes.byref(blocks))
          j += 1
        nbytes += g * wset + _nb(ent[1])
      dev = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(cls._packs[keys[0]][1].device)
      tab = cls._tables[tkey] = (dev, njobs, blocks.value, nbytes)
    call('tg_conv2d_pack_weights_multi', _p(tab[0]), tab[1], tab[2], _stream(), work=('pack_multi:%d' % tab[1], 0, tab[3]))
    for k in keys:
      cls._packs[k][0] = cls.version
    return len(keys)


def _keep_for_aux(ws, accumulate):
  if _DEFERRED is not None and accumulate and ws is not None:
    _DEFERRED.append(ws)      # its slab reduction is queued in the library: alive until flush_slab_reductions()


# Deferred slab reductions of a backward pass (Trainer): between defer_slab_reductions(True) and flush_slab_reductions() the
# library queues the split-K reduction of every filter gradient that accumulates into a gradient sink and the flush issues
# them as one launch (tg_wgrad_defer / tg_wgrad_defer_flush); the slabs those launches read are kept alive here
_DEFERRED = None


def defer_slab_reductions(on):
  global _DEFERRED
  _lib.load().tg_wgrad_defer(1 if on else 0)
  _DEFERRED = [] if on else None


def flush_slab_reductions():
  """Issues the queued reductions on the current stream (which must be ordered after every backward launch of the pass:
  the trainer joins its domain streams first).  -> the number of reductions in the launch."""
  if _DEFERRED is None:
    return 0
  n = _lib.load().tg_wgrad_defer_flush(_stream())
  if n < 0:
    raise _lib.TgError('tg_wgrad_defer_flush failed (%d): %s' % (n, _lib.load().tg_last_error().decode()))
  cur = torch.cuda.current_stream()
  for ws in _DEFERRED:
    ws.record_stream(cur)      # may have been allocated on a domain stream
  del _DEFERRED[:]
  return n


class GradSink:
  """Fused gradient accumulation: a parameter registered here has its gradient ADDED straight into the
  registered buffer (its slice of the optimiser group's flat fp32 gradient, params.ParamStore) by the
  backward kernels themselves, and the autograd Function returns None for it -- no temporary gradient
  tensor, no zero-fill of it, no AccumulateGrad add kernel.  Only used outside create_graph mode (nothing
  differentiates a parameter gradient)."""
  _sinks = {}      # (data_ptr, numel) -> (weakref of the parameter, gradient buffer).  numel is part of the key: the stacked
                   # [2, ...] view of two adjacent discriminator variables (params.ParamStore.pairs) starts where the first does

  @staticmethod
  def _key(p):
    return (p.data_ptr(), p.numel())

  @classmethod
  def register(cls, p, grad):
    cls._sinks[cls._key(p)] = (weakref.ref(p), grad)
    cls._held.pop(cls._key(p), None)

  @classmethod
  def clear(cls):
    cls._sinks.clear()
    cls._held = {}

  @classmethod
  def unregister(cls, p):
    ent = cls._sinks.get(cls._key(p))
    if ent is not None and ent[0]() is p:
      del cls._sinks[cls._key(p)]
      cls._held.pop(cls._key(p), None)

  @classmethod
  def get(cls, p):
    if p is None or torch.is_grad_enabled():
      return None
    ent = cls._sinks.get(cls._key(p))
    if ent is None:
      return None
    if ent[0]() is None:       # the registered parameter is gone and the allocator handed its address out again
      del cls._sinks[cls._key(p)]
      return None
    return ent[1]              # p is the parameter or a reshaped view of it (its memory cannot be anything else while it lives)

  # Pairing of filter gradients: with ``pair`` on (the trainer), the first filter-gradient request of a weight in a
  # backward pass is held back; when a second one for the same weight arrives (the batched real/fake/interpolate pass
  # and the gradient-penalty double-backward term of a discriminator conv; the two encoder passes of a generator step)
  # both run as ONE tg_conv2d_bwd_weight2 launch.  flush() issues the ones that stayed alone; it runs where gradients
  # are consumed (Trainer before the all-reduce / Adam, ParamStore.grad_dict()).
  pair = False
  _held = {}      # (weight data_ptr, numel) -> (x, gy, spec, sink, bias_sink)

  @classmethod
  def submit(cls, w, x, gy, spec, sink, bias_sink=None):
    """``bias_sink``: the layer's bias gradient buffer when it is to be produced from this read of gy."""
    if not cls.pair:
      conv_bwd_weight_raw(x, gy, spec, out=sink, gbias=bias_sink)
      return
    key = cls._key(w)
    first = cls._held.pop(key, None)
    if first is None:
      cls._held[key] = (x, gy, spec, sink, bias_sink)
      return
    gb = first[4] if first[4] is not None else bias_sink
    segs = (1 if first[4] is not None else 0) | (2 if bias_sink is not None else 0)
    if not conv_bwd_weight2_raw(first[0], first[1], x, gy, spec, sink, gb, segs):
      conv_bwd_weight_raw(first[0], first[1], first[2], out=sink, gbias=first[4])
      conv_bwd_weight_raw(x, gy, spec, out=sink, gbias=bias_sink)

  @classmethod
  def flush(cls, only=None):
    """Issues the held filter gradients.  ``only``: predicate on the weight's data_ptr -- the trainer flushes, at the
    end of a backward segment, the weights whose gradient that segment completes and keeps the rest held for the
    pair that a later segment brings."""
    if only is None:
      held, cls._held = cls._held, {}
    else:
      held = {k: v for k, v in cls._held.items() if only(k[0])}
      for k in held:
        del cls._held[k]
    cur = torch.cuda.current_stream() if held else None
    for x, gy, spec, sink, bias_sink in held.values():
      conv_bwd_weight_raw(x, gy, spec, out=sink, gbias=bias_sink)
      x.record_stream(cur)       # may have been produced on a domain stream
      gy.record_stream(cur)
    return len(held)


class Cuts:
  """Segmented backward for the overlapped clone all-reduce (deployment/model_deploy.py:473-503 sums the clones'
  gradients after the whole backward; here the sum of the gradients a segment completes travels over xGMI while the
  next segment runs).  While ``active``, ``cut(t, seg)`` replaces an activation by a detached leaf: the backward of
  the loss stops there (segment 0), and segment ``seg`` later resumes from the producer ``t`` with the gradient the
  leaf received.  Inactive (single clone, growing stages): cut() is the identity and the graph is the usual one."""
  active = False
  pairs = {}       # seg -> [(producer tensor, leaf)]

  @classmethod
  def begin(cls):
    cls.active, cls.pairs = True, {}

  @classmethod
  def end(cls):
    cls.active, cls.pairs = False, {}

  @classmethod
  def cut(cls, t, seg):
    if not cls.active or not torch.is_grad_enabled() or not t.requires_grad:
      return t
    leaf = t.detach().requires_grad_(True)
    cls.pairs.setdefault(seg, []).append((t, leaf))
    return leaf

  @classmethod
  def roots(cls, seg):
    """([producers], [their gradients]) of segment ``seg``; a leaf nothing consumed has no gradient and is dropped."""
    prs = [(t, leaf.grad) for t, leaf in cls.pairs.get(seg, ()) if leaf.grad is not None]
    return [t for t, _ in prs], [g for _, g in prs]


class _State:
  skip_param_grads = False


class no_param_grads:
  """Context: backward passes run inside it do not compute parameter gradients (weights, biases, norm
  affine).  Used around the WGAN-GP inner ``tf.gradients(pred, interp)`` (image_generation.py:429), which
  only needs d pred / d interp; the parameters receive their gradient through the double backward."""

  def __enter__(self):
    self.prev = _State.skip_param_grads
    _State.skip_param_grads = True

  def __exit__(self, *a):
    _State.skip_param_grads = self.prev


class ConvSpec:
  """Static description of one stride-1 conv (TF SAME / VALID padding rules)."""
  __slots__ = ('kh', 'kw', 'pad_t', 'pad_l', 'valid', 'epilogue', 'alpha')

  def __init__(self, k, padding='SAME', epilogue=0, alpha=LRELU_ALPHA):
    self.kh = self.kw = k
    self.valid = padding == 'VALID'
    self.pad_t = self.pad_l = 0 if self.valid else (k - 1) // 2      # TF SAME: low side gets floor((k-1)/2)
    self.epilogue = epilogue
    self.alpha = alpha

  def out_hw(self, h, w):
    return (h - self.kh + 1, w - self.kw + 1) if self.valid else (h, w)


def _mfma_ok(dtype, cin, cout, spec, hin, win):
  if dtype not in HALF_TYPES or cin % 8 or cout % 8:
    return False
  if spec.kh == 1 or spec.kh == 3:
    return True
  return spec.valid and hin == spec.kh and win == spec.kw     # k x k VALID on k x k input -> dense


_SLOW_SEEN = set()


def _slow_dispatch(n, ho, wo, cin, cout, spec):
  """A 16-bit conv the MFMA kernels do not take runs on the one-thread-per-output kernels (conv_direct.hip: the exact
  fp32 path's kernels) -- two orders of magnitude slower.  Never silently: above 1 MFLOP a warning once per shape, an error
  under TG_STRICT_DISPATCH=1 (bench.py sets it: a bench line must not time such a fallback under an MFMA label)."""
  flops = 2 * n * ho * wo * cout * spec.kh * spec.kw * cin
  if flops < 1e6:
    return
  key = (ho, wo, cin, cout, spec.kh, spec.valid)
  msg = ('16-bit conv k%d %s c%d>%d at %dx%d (n %d, %.1f MFLOP) is not taken by the MFMA kernels (cin %% 8, cout %% 8, k in {1, 3} '
         'or the dense k x k VALID case) and runs on conv_*_direct' % (spec.kh, 'VALID' if spec.valid else 'SAME', cin, cout, ho, wo,
                                                                        n, flops / 1e6))
  if os.environ.get('TG_STRICT_DISPATCH') == '1':
    raise _lib.TgError(msg + ' (TG_STRICT_DISPATCH=1)')
  if key not in _SLOW_SEEN:
    _SLOW_SEEN.add(key)
    import warnings
    warnings.warn(msg)


def _esize(t):
  return 2 if t.dtype in HALF_TYPES else 4


def _shape_tag(t):
  return ':c%d:hw%d:n%d' % (t.shape[-1], t.shape[-2], t.shape[0]) if t.dim() == 4 else ':numel%d' % t.numel()


def _nb(*ts):
  """Bytes of the given tensors (None skipped): the compulsory traffic of a streaming kernel that touches each once."""
  return sum(t.numel() * t.element_size() for t in ts if t is not None)


def _conv_work(d, tag, es):
  """Algorithmic work of one conv launch: MACs*2 and compulsory bytes (input + output + weights once)."""
  flops = 2 * d.n * d.hout * d.wout * d.cout * d.kh * d.kw * d.cin
  by = es * d.n * (d.hin * d.win * d.cin + d.hout * d.wout * d.cout) + es * d.kh * d.kw * d.cin * d.cout
  algo = 'mfma' if d.algo == TG_ALGO_MFMA else 'direct'
  return ('%s:%s:k%d:c%d>%d:hw%d:n%d' % (tag, algo, d.kh, d.cin, d.cout, d.hout, d.n), flops, by)


def _wg(w):
  """Weight sets of a conv kernel tensor: a 5-d [G, kh, kw, cin, cout] tensor holds G sets (TgConvDesc.groups: the batch is
  G equal image ranges, range g convolved with set g -- the two discriminators' layers as one call), else 1."""
  return int(w.shape[0]) if w.dim() == 5 else 1


def _desc(x_shape, cout, spec, dtype, epilogue, groups=1):
  n, h, w, cin = x_shape
  ho, wo = spec.out_hw(h, w)
  d = TgConvDesc()
  d.n, d.hin, d.win, d.cin = n, h, w, cin
  d.hout, d.wout, d.cout = ho, wo, cout
  d.kh, d.kw, d.pad_t, d.pad_l = spec.kh, spec.kw, spec.pad_t, spec.pad_l
  d.dtype = _DTYPES[dtype]
  d.algo = TG_ALGO_MFMA if _mfma_ok(dtype, cin, cout, spec, h, w) else TG_ALGO_DIRECT
  if d.algo == TG_ALGO_DIRECT and dtype in HALF_TYPES:
    _slow_dispatch(n, ho, wo, cin, cout, spec)
  d.epilogue = epilogue
  d.lrelu_alpha = spec.alpha
  d.groups = groups
  assert groups == 1 or n % groups == 0, (n, groups)
  return d


# ------------------------------------------------------------------------------------------------
# raw (non-autograd) kernel wrappers
# ------------------------------------------------------------------------------------------------
def conv_fwd_raw(x, w, bias, spec, epilogue):
  _chk(x, w, bias)
  d = _desc(x.shape, w.shape[-1], spec, x.dtype, epilogue, _wg(w))
  y = torch.empty((d.n, d.hout, d.wout, d.cout), dtype=x.dtype, device=x.device)
  wk = PackCache.get(w, d, 0) if d.algo == TG_ALGO_MFMA else w
  call('tg_conv2d_fwd', ctypes.byref(d), _p(x), _p(wk), _p(bias), _p(y), _stream(),
       work=lambda: _conv_work(d, 'fwd', _esize(x)))
  return y


def conv_fwd_pool_raw(x, w, bias, spec, epilogue):
  """(z, avg_pool2(z)): one launch where the tile kernels take the shape (tg_conv2d_fwd_pool), else conv + pool."""
  _chk(x, w, bias)
  d = _desc(x.shape, w.shape[-1], spec, x.dtype, epilogue, _wg(w))
  if USE_CONV_POOL and d.algo == TG_ALGO_MFMA and d.hout % 2 == 0 and d.wout % 2 == 0 and \
      _lib.load().tg_conv2d_fwd_pool_supported(ctypes.byref(d)):
    z = torch.empty((d.n, d.hout, d.wout, d.cout), dtype=x.dtype, device=x.device)
    zp = torch.empty((d.n, d.hout // 2, d.wout // 2, d.cout), dtype=x.dtype, device=x.device)

    def work():
      tag, fl, by = _conv_work(d, 'fwd', _esize(x))
      return tag, fl, by + zp.numel() * _esize(x)
    wk = PackCache.get(w, d, 0)      # a local keeps an uncached pack alive across the launch
    call('tg_conv2d_fwd_pool', ctypes.byref(d), _p(x), _p(wk), _p(bias), _p(z), _p(zp), _stream(),
         work=work)
    return z, zp
  z = conv_fwd_raw(x, w, bias, spec, epilogue)
  n, h, ww, c = z.shape
  zp = torch.empty((n, h // 2, ww // 2, c), dtype=z.dtype, device=z.device)
  call('tg_pool2x2_fwd', _p(z), _p(zp), n, h, ww, c, 0.25, _dt(z), _stream(),
       work=('pool_fwd' + _shape_tag(z), 0, int(1.25 * z.numel()) * _esize(z)))
  return z, zp


def conv_fwd_pool_signs_supported(x, w, spec, epilogue):
  """Can (sign bits of z, avg_pool2(z)) come out of one launch for this layer (tg_conv2d_fwd_pool_signs)?"""
  if not (USE_CONV_POOL and USE_POOL_SIGNS) or x.dtype not in HALF_TYPES or w.shape[-1] % 8 or not (epilogue & TG_EPI_LRELU):
    return False
  d = _desc(x.shape, w.shape[-1], spec, x.dtype, epilogue, _wg(w))
  return d.algo == TG_ALGO_MFMA and d.hout % 2 == 0 and d.wout % 2 == 0 and \
      bool(_lib.load().tg_conv2d_fwd_pool_supported(ctypes.byref(d)))


def conv_fwd_pool_signs_raw(x, w, bias, spec, epilogue):
  """(signs, avg_pool2(z)) of z = epilogue(conv(x, w) + bias): z itself is never written; signs is uint8
  [n, h, w, cout / 8], bit j of byte q = (z[.., 8q+j] > 0)."""
  _chk(x, w, bias)
  d = _desc(x.shape, w.shape[-1], spec, x.dtype, epilogue, _wg(w))
  signs = torch.empty((d.n, d.hout, d.wout, d.cout // 8), dtype=torch.uint8, device=x.device)
  zp = torch.empty((d.n, d.hout // 2, d.wout // 2, d.cout), dtype=x.dtype, device=x.device)

  def work():      # input + pooled output + one bit per full-resolution output
    tag, fl, _ = _conv_work(d, 'fwd', _esize(x))
    return tag, fl, _nb(x, zp, signs) + _esize(x) * d.kh * d.kw * d.cin * d.cout
  wk = PackCache.get(w, d, 0)      # a local keeps an uncached pack alive across the launch
  call('tg_conv2d_fwd_pool_signs', ctypes.byref(d), _p(x), _p(wk), _p(bias), _p(signs), _p(zp), _stream(),
       work=work)
  return signs, zp


def lrelu_pool_bwd_signs(gzp, signs, alpha, bias, want_bias):
  """g = 0.25 * upsample2(gzp) * (sign ? 1 : alpha) [+ the bias gradient]; see lrelu_pool_bwd."""
  _chk(gzp, signs)
  n, h, w, c8 = signs.shape
  c = c8 * 8
  g = torch.empty((n, h, w, c), dtype=gzp.dtype, device=gzp.device)
  sink = GradSink.get(bias) if want_bias else None
  gb = None
  if want_bias:
    gb = sink if sink is not None else torch.empty(tuple(bias.shape), dtype=torch.float32, device=gzp.device)
  fused_bias = want_bias and not deterministic() and bias.dim() == 1      # stacked biases (grouped conv): per-group sums below
  call('tg_lrelu_pool_bwd_signs', _p(gzp), _p(signs), _p(g), _p(gb if fused_bias else None), n, h, w, c, alpha,
       1 if sink is not None else 0, _dt(gzp), _stream(), work=('lrelu_pool_bwd' + _shape_tag(g), 0, _nb(gzp, signs, g)))
  if want_bias and not fused_bias:
    _channel_sum_into(g, gb, sink is not None)
  return g, (None if sink is not None else gb)


class ConvStats:
  """Per-workgroup statistics partials a conv wrote from its epilogue (tg_conv2d_fwd_stats): fp32
  [n][chunks][2][cout], consumed by norm_act instead of a statistics pass over the conv output."""
  __slots__ = ('part', 'chunks')

  def __init__(self, part, chunks):
    self.part, self.chunks = part, chunks


def conv_fwd_stats_raw(x, w, spec):
  """(y, ConvStats) of a bias-free conv, or (y, None) when the kernel this shape dispatches has no statistics epilogue."""
  _chk(x, w)
  d = _desc(x.shape, w.shape[-1], spec, x.dtype, 0, _wg(w))
  chunks = _lib.load().tg_conv2d_fwd_stats_chunks(ctypes.byref(d)) if (USE_CONV_STATS and d.algo == TG_ALGO_MFMA) else 0
  if chunks <= 0:
    return conv_fwd_raw(x, w, None, spec, 0), None
  y = torch.empty((d.n, d.hout, d.wout, d.cout), dtype=x.dtype, device=x.device)
  part = torch.empty(d.n * chunks * 2 * d.cout, dtype=torch.float32, device=x.device)
  wk = PackCache.get(w, d, 0)      # a local keeps an uncached pack alive across the launch
  call('tg_conv2d_fwd_stats', ctypes.byref(d), _p(x), _p(wk), _p(y), _p(part), chunks, _stream(),
       work=lambda: _conv_work(d, 'fwd', _esize(x)))
  return y, ConvStats(part, chunks)


def conv_bwd_data_raw(gy, w, x_shape, spec):
  _chk(gy, w)
  d = _desc(x_shape, w.shape[-1], spec, gy.dtype, 0, _wg(w))
  gx = torch.empty(tuple(x_shape), dtype=gy.dtype, device=gy.device)
  wk = PackCache.get(w, d, 1) if d.algo == TG_ALGO_MFMA else w
  call('tg_conv2d_bwd_data', ctypes.byref(d), _p(gy), _p(wk), _p(gx), _stream(),
       work=lambda: _conv_work(d, 'dgrad', _esize(gy)))
  return gx


def conv_bwd_data_masked_raw(gy, w, x_act, spec):
  """gx = conv^T(gy, w) * (x_act > 0 ? 1 : alpha): backward-data with the LeakyReLU backward of the layer that produced
  this conv's input ``x_act`` folded into the epilogue (tg_conv2d_bwd_data_masked)."""
  _chk(gy, w, x_act)
  d = _desc(x_act.shape, w.shape[-1], spec, gy.dtype, 0, _wg(w))
  gx = torch.empty_like(x_act)
  wk = PackCache.get(w, d, 1) if d.algo == TG_ALGO_MFMA else w
  def work():      # the mask is one more read of a tensor of the input's size
    tag, fl, by = _conv_work(d, 'dgrad', _esize(gy))
    return tag.replace('dgrad:', 'dgrad_masked:'), fl, by + _nb(x_act)
  call('tg_conv2d_bwd_data_masked', ctypes.byref(d), _p(gy), _p(wk), _p(x_act), _p(gx), _stream(), work=work)
  return gx


# backward-data of a discriminator block's last conv straight from the pooled gradient + the layer's sign bytes
# (tg_conv2d_bwd_data_unpool) wherever the layer's filter gradient is not needed (USE_DGRAD_UNPOOL = False (tests): the two-launch path)
USE_DGRAD_UNPOOL = True
# ... and where the filter / bias gradient does need that tensor (a discriminator step), the same kernel writes it
# (USE_DGRAD_UNPOOL_KEEP = False (tests): tg_lrelu_pool_bwd_signs + the plain backward-data there)
USE_DGRAD_UNPOOL_KEEP = True
# ... and for block ends that kept their activation output instead of sign bytes (the gradient-penalty pass' nodes in the
# second differentiation): the signs read from that tensor (USE_DGRAD_UNPOOL_ACT = False (tests): tg_lrelu_pool_bwd + backward-data)
USE_DGRAD_UNPOOL_ACT = True
# ... and in the gradient penalty's first (create_graph) backward pass: LReluPoolBwdFn + MaskedDgradFn as the one
# differentiable UnpoolMaskedDgradFn (USE_DGRAD_UNPOOL_GP = False (tests): the two nodes)
USE_DGRAD_UNPOOL_GP = True


def conv_bwd_data_unpool_raw(gzp, signs, w, x_act, x_shape, spec, keep=False):
  """gx = conv^T(unpool_lrelu(gzp, signs), w) [* mask(x_act)] without reading the full-resolution gradient from memory, or
  None when the kernels do not take the layer.  ``gzp`` [n, h/2, w/2, cout]: gradient of the pooled output; ``signs``
  [n, h, w, cout/8] uint8 (conv_fwd_pool_signs_raw); ``x_act``: the conv's forward input when the producer's LeakyReLU
  backward is folded in (as conv_bwd_data_masked_raw), else None.  ``keep``: -> (gx, g) with g = unpool_lrelu(gzp, signs)
  [n, h, w, cout] written by the same kernel (for the layer's filter / bias gradient)."""
  _chk(gzp, signs, w, x_act)
  d = _desc(x_shape, w.shape[-1], spec, gzp.dtype, 0, _wg(w))
  if d.algo != TG_ALGO_MFMA or not _lib.load().tg_conv2d_bwd_data_unpool_supported(ctypes.byref(d)):
    return None
  gx = torch.empty(tuple(x_shape), dtype=gzp.dtype, device=gzp.device)
  g = torch.empty((d.n, d.hout, d.wout, d.cout), dtype=gzp.dtype, device=gzp.device) if keep else None
  wk = PackCache.get(w, d, 1)

  def work():      # reads: a quarter of the gradient tensor + its sign bytes (+ the mask); the tensor itself is at most written
    tag, fl, by = _conv_work(d, 'dgrad', _esize(gzp))
    full = d.n * d.hout * d.wout * d.cout * _esize(gzp)
    return (tag.replace('dgrad:', 'dgrad_unpool_keep:' if keep else 'dgrad_unpool:'), fl,
            by - (0 if keep else full) + _nb(gzp) + signs.numel() + (_nb(x_act) if x_act is not None else 0))
  # ``signs`` uint8: the sign bytes; a 16-bit tensor: the layer's activation output itself (a pass that kept it)
  entry = 'tg_conv2d_bwd_data_unpool' if signs.dtype == torch.uint8 else 'tg_conv2d_bwd_data_unpool_act'
  call(entry, ctypes.byref(d), _p(gzp), _p(signs), _p(wk), _p(x_act), _p(gx), _p(g), _stream(), work=work)
  return (gx, g) if keep else gx


def conv_fwd_masked_raw(x, w, mask_src, spec):
  """y = conv(x, w) * (mask_src > 0 ? 1 : alpha): a forward conv with the LeakyReLU derivative of ``mask_src`` (the shape
  of y) in its epilogue (tg_conv2d_fwd_masked) -- the second backward pass of the gradient penalty."""
  _chk(x, w, mask_src)
  d = _desc(x.shape, w.shape[-1], spec, x.dtype, 0, _wg(w))
  y = torch.empty((d.n, d.hout, d.wout, d.cout), dtype=x.dtype, device=x.device)
  assert tuple(mask_src.shape) == tuple(y.shape), (tuple(mask_src.shape), tuple(y.shape))
  wk = PackCache.get(w, d, 0) if d.algo == TG_ALGO_MFMA else w
  def work():      # the mask is one more read of a tensor of the output's size
    tag, fl, by = _conv_work(d, 'fwd', _esize(x))
    return tag.replace('fwd:', 'fwd_masked:'), fl, by + _nb(mask_src)
  call('tg_conv2d_fwd_masked', ctypes.byref(d), _p(x), _p(wk), _p(mask_src), _p(y), _stream(), work=work)
  return y


def conv_bwd_weight_raw(x, gy, spec, out=None, gbias=None, groups=1):
  """gw = x^T * gy; with ``out`` the result is ADDED into that fp32 HWIO buffer (gradient sink).  ``gbias``: fp32 [cout]
  buffer that also receives += sum over pixels of gy (the layer's bias gradient, from the same read of gy).  ``groups`` (or
  a 5-d ``out``): G weight sets, image range g of the batch feeds gw[g] (and gbias[g])."""
  _chk(x, gy)
  if out is not None and out.dim() == 5:
    groups = int(out.shape[0])
  d = _desc(x.shape, gy.shape[3], spec, x.dtype, 0, groups)
  shape = (spec.kh, spec.kw, x.shape[3], gy.shape[3])
  if groups > 1:
    shape = (groups,) + shape
  if out is None:
    gw = torch.empty(shape, dtype=torch.float32, device=x.device)
  else:
    assert tuple(out.shape) == shape and out.is_contiguous(), (tuple(out.shape), shape)
    gw = out
  nbytes = _lib.load().tg_conv2d_bwd_weight_workspace(ctypes.byref(d))
  ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device) if nbytes else None
  if gbias is not None:
    call('tg_conv2d_bwd_weight_bias', ctypes.byref(d), _p(x), _p(gy), _p(gw), _p(gbias), 0 if out is None else 1, _p(ws),
         nbytes, _stream(), work=lambda: _conv_work(d, 'wgrad', _esize(x)))
  else:
    call('tg_conv2d_bwd_weight', ctypes.byref(d), _p(x), _p(gy), _p(gw), 0 if out is None else 1, _p(ws), nbytes, _stream(),
         work=lambda: _conv_work(d, 'wgrad', _esize(x)))
  _keep_for_aux(ws, out is not None)
  return gw


def conv_bwd_weight2_raw(xa, gya, xb, gyb, spec, out, gbias=None, bias_segs=3):
  """out += wgrad(xa, gya) + wgrad(xb, gyb) in one launch; False when the layer is not eligible.  ``gbias`` += the
  pixel sums of gya (bias_segs bit 0) and / or gyb (bit 1)."""
  _chk(xa, gya, xb, gyb)
  if xa.shape[1:] != xb.shape[1:] or gya.shape[1:] != gyb.shape[1:] or xa.dtype != xb.dtype:
    return False
  groups = int(out.shape[0]) if out.dim() == 5 else 1
  if xb.shape[0] % groups:
    return False
  d = _desc(xa.shape, gya.shape[3], spec, xa.dtype, 0, groups)
  nb = xb.shape[0]
  nbytes = _lib.load().tg_conv2d_bwd_weight2_workspace(ctypes.byref(d), nb)
  if not nbytes:
    return False
  ws = torch.empty(nbytes, dtype=torch.uint8, device=xa.device)
  es = _esize(xa)

  def work():
    tag, fl, by = _conv_work(d, 'wgrad', es)
    k = (d.n + nb) / d.n
    return (tag.replace(':n%d' % d.n, ':n%d+%d' % (d.n, nb)), int(fl * k), int(by * k))
  if gbias is not None:
    call('tg_conv2d_bwd_weight2_bias', ctypes.byref(d), nb, _p(xa), _p(gya), _p(xb), _p(gyb), _p(out), _p(gbias), bias_segs,
         1, _p(ws), nbytes, _stream(), work=work)
  else:
    call('tg_conv2d_bwd_weight2', ctypes.byref(d), nb, _p(xa), _p(gya), _p(xb), _p(gyb), _p(out), 1, _p(ws), nbytes,
         _stream(), work=work)
  _keep_for_aux(ws, True)
  return True


def _bias_rides_with_filter_gradient(need_b, need_w, bias, w):
  """Can the filter-gradient kernel sum g over pixels as well (the layer's bias gradient from the same read of g,
  tg_conv2d_bwd_weight*_bias)?  When the bias has a gradient sink and the weight has one too (the paired launches of
  GradSink) -- or has none in a first-order pass: a spectrally normalised kernel, whose ``w`` is the per-run w_bar and
  whose gradient is a plain tensor (config 4: 38 tg_channel_sum launches of 15 us per step otherwise)."""
  return (need_b and need_w and GradSink.get(bias) is not None and not deterministic()
          and (GradSink.get(w) is not None or not torch.is_grad_enabled()))


def _weight_grad(x, g, spec, w, bias_sink=None):
  """Parameter gradient of a conv: into the sink when ``w`` has one (returns None), else a differentiable node.
  ``bias_sink``: also add the bias gradient (pixel sums of g) into that buffer from the same kernel."""
  sink = GradSink.get(w)
  if sink is not None:
    GradSink.submit(w, x, g, spec, sink, bias_sink)
    return None
  if bias_sink is not None:      # first-order pass, no sink for w: the gradient tensor, the bias gradient riding along
    assert not torch.is_grad_enabled()
    return conv_bwd_weight_raw(x, g, spec, gbias=bias_sink, groups=_wg(w))
  return ConvBwdWeightFn.apply(x, g, spec, _wg(w))


def deterministic():
  """Is the library's deterministic mode on (TG_DETERMINISTIC=1 / tg_set_deterministic)?  The host then routes the sums
  that end in fp32 atomics -- bias gradients, fromRGB / toRGB filter gradients, loss sums -- through their ORDERED forms
  (per-workgroup partials in a workspace, added in workgroup order: tg_*_ordered) instead of the fused / atomic ones."""
  return bool(_lib.load().tg_get_deterministic())


_ORDERED_ROWS = 512      # workgroups whose partial rows the workspace can hold


def _channel_sum_into(g, out, accumulate):
  """out[c] (+)= sum over pixels of g: ordered two-stage sum in deterministic mode, tg_channel_sum otherwise.  A 2-d
  ``out`` [G, c] (the stacked biases of a grouped conv): image range i of g feeds row i."""
  c = g.shape[-1]
  if out.dim() == 2:
    n1 = g.shape[0] // out.shape[0]
    for i in range(out.shape[0]):
      _channel_sum_into(g[i * n1:(i + 1) * n1], out[i], accumulate)
    return
  if deterministic():
    ws = torch.empty(_ORDERED_ROWS * c, dtype=torch.float32, device=g.device)
    call('tg_channel_sum_ordered', _p(g), _p(out), g.numel() // c, c, 1 if accumulate else 0, _p(ws), ws.numel(), _dt(g),
         _stream(), work=('channel_sum' + _shape_tag(g), 0, _nb(g)))
  else:
    call('tg_channel_sum', _p(g), _p(out), g.numel() // c, c, 1 if accumulate else 0, _dt(g), _stream(),
         work=('channel_sum' + _shape_tag(g), 0, _nb(g)))


def _bias_grad(g, bias):
  sink = GradSink.get(bias)
  if sink is not None:
    _channel_sum_into(g, sink, True)
    return None
  return ChannelSumFn.apply(g, int(bias.shape[0]) if bias.dim() == 2 else 1)


def lrelu_pool_bwd(gz, gzp, z, alpha, bias, want_bias):
  """g = (gz + 0.25 * upsample2(gzp)) * lrelu'(z), optionally with the bias gradient, in one pass.
  Returns (g, gb) with gb None when it was added into the bias' gradient sink (or not wanted)."""
  _chk(gz, gzp, z)
  g = torch.empty_like(z)
  n, h, w, c = z.shape
  sink = GradSink.get(bias) if want_bias else None
  gb = None
  if want_bias:
    gb = sink if sink is not None else torch.empty(tuple(bias.shape), dtype=torch.float32, device=z.device)
  # deterministic mode: the bias sum is its own ordered two-stage pass; stacked biases (grouped conv): per-group sums
  fused_bias = want_bias and not deterministic() and bias.dim() == 1
  call('tg_lrelu_pool_bwd', _p(gz), _p(gzp), _p(z), _p(g), _p(gb if fused_bias else None), n, h, w, c, alpha,
       1 if sink is not None else 0, _dt(z), _stream(),
       work=('lrelu_pool_bwd' + _shape_tag(z), 0, int((2 + (gz is not None) + 0.25 * (gzp is not None)) * z.numel()) * _esize(z)))
  if want_bias and not fused_bias:
    _channel_sum_into(g, gb, sink is not None)
  return g, (None if sink is not None else gb)


def lrelu_bwd_raw(g, z, alpha):
  _chk(g, z)
  out = torch.empty_like(g)
  call('tg_lrelu_bwd', _p(g), _p(z), _p(out), g.numel(), alpha, _dt(g), _stream(),
       work=('lrelu_bwd' + _shape_tag(z), 0, 3 * z.numel() * _esize(z)))
  return out


def channel_sum_raw(g, groups=1):
  _chk(g)
  c = g.shape[-1]
  out = torch.empty((groups, c) if groups > 1 else (c,), dtype=torch.float32, device=g.device)
  _channel_sum_into(g, out, False)
  return out


def sample_lerp(x, y, alpha):
  """x + alpha[b] * (y - x)  -- WGAN-GP interpolates (image_generation.py:424).  No autograd."""
  _chk(x, y, alpha)
  out = torch.empty_like(x)
  b = x.shape[0]
  call('tg_sample_lerp', _p(x), _p(y), _p(alpha), _p(out), b, x.numel() // b, _dt(x), _stream(),
       work=('sample_lerp' + _shape_tag(x), 0, _nb(x, y, out)))
  return out


def zero_(t):
  """t[...] = 0 for a contiguous device tensor, through the library (tg_fill_scaled) rather than the framework's fill."""
  assert t.is_contiguous()
  call('tg_fill_scaled', _p(t), None, 0.0, t.numel(), _dt(t), _stream(), work=('fill:numel%d' % t.numel(), 0, _nb(t)))
  return t


def fill(shape, value, dtype, device, scalar=None):
  out = torch.empty(shape, dtype=dtype, device=device)
  call('tg_fill_scaled', _p(out), _p(scalar), float(value), out.numel(), _dt(out), _stream(),
       work=('fill:numel%d' % out.numel(), 0, _nb(out)))
  return out


def cast_raw(x, dtype):
  _chk(x)
  out = torch.empty(x.shape, dtype=dtype, device=x.device)
  call('tg_cast', _p(x), _p(out), x.numel(), _dt(x), _dt(out), _stream(), work=('cast:numel%d' % x.numel(), 0, _nb(x, out)))
  return out




# ------------------------------------------------------------------------------------------------
# convolution  (layers.conv2d, nets/pggan_utils.py:316-320)
# ------------------------------------------------------------------------------------------------
def first_order_only(t):
  """Marks ``t`` -- an input a pass is differentiated with respect to under create_graph (the gradient penalty's interpolates,
  image_generation.py:414-439) -- as wanting no gradient from the FINAL backward: d penalty / d interpolates is not a
  parameter gradient, yet the second backward would compute it (the first layer's backward-data at full resolution, the
  input scaling's backward, a copy into .grad).  The nodes that read ``t`` skip their input gradient when they run without
  a graph being recorded.  Returns ``t``."""
  t._tg_first_order_only = True
  return t


def _input_grad_wanted(ctx, x):
  return ctx.needs_input_grad[0] and not (getattr(x, '_tg_first_order_only', False) and not torch.is_grad_enabled())


def _conv_backward(ctx, gz, gzp=None):
  """Shared backward of Conv2dFn / Conv2dPoolFn.  ``gzp``: gradient of the 2x2-average-pooled output."""
  x, w, z, bias = ctx.saved_tensors
  spec = ctx.spec
  params = not _State.skip_param_grads
  need_x = _input_grad_wanted(ctx, x)
  need_w = ctx.needs_input_grad[1] and params
  need_b = bool(ctx.epilogue & TG_EPI_BIAS) and ctx.needs_input_grad[2] and params
  gb = None
  gz = gz.contiguous() if gz is not None else None
  gzp = gzp.contiguous() if gzp is not None else None
  fused = (ctx.epilogue & TG_EPI_LRELU) and not torch.is_grad_enabled()
  # the (single) consumer of z is a conv whose backward-data applies this layer's LeakyReLU mask itself -- in
  # first-order passes with the raw masked kernel, in create_graph passes with the differentiable MaskedDgradFn
  premasked = bool(ctx.epilogue & TG_EPI_LRELU) and gzp is None and getattr(ctx, 'tg_premasked', False)
  bias_sink = None
  pooled_lrelu = None
  gx_done = None      # the input gradient when the branch below already ran the backward-data (the unpooling kernel)
  if gzp is not None and not fused:
    if (gz is None and (ctx.epilogue & TG_EPI_LRELU) and USE_DGRAD_UNPOOL and USE_DGRAD_UNPOOL_GP and getattr(ctx, 'mask_input', False)
        and need_x and not need_w and not need_b and z.dtype in HALF_TYPES
        and _unpool_act_supported(tuple(x.shape), w, spec, z.dtype)):
      # create_graph pass over a pooled LeakyReLU layer whose input gradient is all that is wanted (the gradient penalty's
      # inner gradient): unpool + mask + masked backward-data as ONE differentiable node
      gx = UnpoolMaskedDgradFn.apply(gzp, w, x, z, spec)
      if gx.grad_fn is not None:
        gx.grad_fn.tg_masks_with = (x.data_ptr(), tuple(x.shape))      # what this node masks an incoming cotangent with
      return gx, None, None, None, None, None
    if gz is None and (ctx.epilogue & TG_EPI_LRELU):
      # create_graph pass over a pooled LeakyReLU layer: unpool + mask in one differentiable node
      pooled_lrelu = LReluPoolBwdFn.apply(gzp, z, spec.alpha)
      if pooled_lrelu.grad_fn is not None:
        pooled_lrelu.grad_fn.tg_masks_with = (z.data_ptr(), tuple(z.shape))      # see the mask_input branch below
    else:
      # differentiable composition or no activation: materialise the upsampled pooled gradient
      up = Pool2BwdFn.apply(gzp, 0.25, (z.shape[1], z.shape[2]) if z is not None else ctx.out_hw)
      gz = up if gz is None else gz + up
    gzp = None
  if premasked:
    g = gz
    if _bias_rides_with_filter_gradient(need_b, need_w, bias, w):
      bias_sink = GradSink.get(bias)      # the filter-gradient kernel sums g over pixels as well
      need_b = False
  elif pooled_lrelu is not None:
    g = pooled_lrelu
  elif getattr(ctx, 'tg_signs', False):      # z holds the sign bits of the layer's output (Conv2dPoolSignsFn)
    if USE_DGRAD_UNPOOL and need_x and not torch.is_grad_enabled():
      # this layer's gradient is formed from the pooled gradient and the sign bytes inside the backward-data kernel.  A
      # generator step (the discriminator's parameters are not trained): nothing else reads it and it is never in memory;
      # a discriminator step: the same kernel writes it for the filter / bias gradient
      keep = need_w or need_b
      if not keep or USE_DGRAD_UNPOOL_KEEP:
        out = conv_bwd_data_unpool_raw(gzp, z, w, x if getattr(ctx, 'mask_input', False) else None, tuple(x.shape), spec, keep)
        if out is not None and not keep:
          return out, None, None, None, None, None
        if out is not None:
          gx_done, g = out
    if gx_done is None:
      g, gb = lrelu_pool_bwd_signs(gzp, z, spec.alpha, bias if need_b else None, need_b)
      need_b = False
    elif _bias_rides_with_filter_gradient(need_b, need_w, bias, w):
      bias_sink = GradSink.get(bias)      # the filter-gradient kernel sums g over pixels as well
      need_b = False
  elif ctx.epilogue & TG_EPI_LRELU:
    if (fused and gz is None and gzp is not None and USE_DGRAD_UNPOOL and USE_DGRAD_UNPOOL_ACT and need_x
        and z.dtype in HALF_TYPES):
      # a block end whose activation output was kept (the gradient-penalty pass), differentiated once more: as above, the
      # signs taken from z itself
      keep = need_w or need_b
      out = conv_bwd_data_unpool_raw(gzp, z, w, x if getattr(ctx, 'mask_input', False) else None, tuple(x.shape), spec, keep)
      if out is not None and not keep:
        return out, None, None, None, None, None
      if out is not None:
        gx_done, g = out
        if _bias_rides_with_filter_gradient(need_b, need_w, bias, w):
          bias_sink = GradSink.get(bias)
          need_b = False
    if gx_done is not None:
      pass
    elif fused and (need_b or gzp is not None):
      g, gb = lrelu_pool_bwd(gz, gzp, z, spec.alpha, bias if need_b else None, need_b)
      need_b = False
    else:
      g = LReluBwdFn.apply(gz, z, spec.alpha)
  else:
    g = gz
  gx = gx_done
  if gx is not None:
    pass
  elif need_x:
    if getattr(ctx, 'mask_input', False):      # x = the producer's LeakyReLU output
      if torch.is_grad_enabled():
        # the node that produced g masks the cotangent it gets back from us with THIS layer's LeakyReLU output z: when
        # we are its only consumer (the gradient penalty's inner gradient: no parameter gradients hang off g) our own
        # backward applies that mask in its conv's epilogue and tells the producer so
        node = g.grad_fn
        premask = (USE_GP_PREMASK and _State.skip_param_grads and z is not None and (ctx.epilogue & TG_EPI_LRELU)
                   and node is not None and getattr(node, 'tg_masks_with', None) == (z.data_ptr(), tuple(z.shape)))
        gx = MaskedDgradFn.apply(g, w, x, spec, z if premask else None)
        if premask:
          node.tg_v_premasked = True
        if gx.grad_fn is not None:
          gx.grad_fn.tg_masks_with = (x.data_ptr(), tuple(x.shape))      # what this node masks an incoming cotangent with
      else:
        gx = conv_bwd_data_masked_raw(g, w, x, spec)
    else:
      # the first conv of a discriminator block (its input is a pooled tensor: nothing to mask on the way in).  As in the
      # mask_input branch: when the node that produced g masks the cotangent it gets back from us with THIS layer's
      # LeakyReLU output and we are its only consumer, our backward applies that mask in its conv's epilogue
      node = g.grad_fn
      premask = (torch.is_grad_enabled() and USE_GP_PREMASK and _State.skip_param_grads and z is not None
                 and bool(ctx.epilogue & TG_EPI_LRELU) and node is not None
                 and getattr(node, 'tg_masks_with', None) == (z.data_ptr(), tuple(z.shape))
                 and not getattr(node, 'tg_v_premasked', False))
      gx = ConvBwdDataFn.apply(g, w, tuple(x.shape), spec, z if premask else None)
      if premask:
        node.tg_v_premasked = True
  gw = _weight_grad(x, g, spec, w, bias_sink) if need_w else None
  if need_b:
    gb = _bias_grad(g, bias)
  return gx, gw, gb, None, None, None


class Conv2dFn(torch.autograd.Function):
  """z = epilogue(conv(x, w) [+ bias]) with epilogue in {none, bias, bias+lrelu, lrelu}."""

  @staticmethod
  def forward(ctx, x, w, bias, spec, epilogue, mask_input=False):
    z = conv_fwd_raw(x, w, bias, spec, epilogue)
    ctx.spec, ctx.epilogue, ctx.mask_input = spec, epilogue, mask_input
    ctx.out_hw = (z.shape[1], z.shape[2])
    # No gradient for z means no work here.  The gradient penalty's pass reads the activations after the minibatch
    # stddev only through LeakyReLU masks: the second backward reaches those convs with an undefined gradient, which the
    # default would turn into a tensor of zeros -- backward-data, filter-gradient and mbstd launches that compute zeros.
    ctx.set_materialize_grads(False)
    ctx.save_for_backward(x, w, z if (epilogue & TG_EPI_LRELU) else None, bias)
    return z

  @staticmethod
  def backward(ctx, gz):
    if gz is None:
      return None, None, None, None, None, None
    return _conv_backward(ctx, gz)


class Conv2dStatsFn(torch.autograd.Function):
  """y = conv(x, w) of a normalised layer; the statistics partials the kernel wrote ride along on ``holder`` (a list the
  caller passes; not a tensor output, so the autograd graph is the plain conv's)."""

  @staticmethod
  def forward(ctx, x, w, spec, holder):
    z, st = conv_fwd_stats_raw(x, w, spec)
    holder.append(st)
    ctx.spec, ctx.epilogue, ctx.mask_input = spec, 0, False
    ctx.out_hw = (z.shape[1], z.shape[2])
    ctx.set_materialize_grads(False)
    ctx.save_for_backward(x, w, None, None)
    return z

  @staticmethod
  def backward(ctx, gz):
    if gz is None:
      return None, None, None, None
    return _conv_backward(ctx, gz)[:2] + (None, None)


class Conv2dPoolFn(torch.autograd.Function):
  """(z, avg_pool2(z)) with z = epilogue(conv(x, w) [+ bias]) -- the last conv of a discriminator block and
  the tf.nn.avg_pool after it (nets/pggan.py:304-306).  Outside create_graph mode the backward folds the
  pool's gradient into the LeakyReLU / bias-gradient kernel instead of upsampling it through HBM."""

  @staticmethod
  def forward(ctx, x, w, bias, spec, epilogue, mask_input=False):
    z, zp = conv_fwd_pool_raw(x, w, bias, spec, epilogue)
    ctx.mask_input = mask_input
    n, h, ww, c = z.shape
    ctx.spec, ctx.epilogue = spec, epilogue
    ctx.out_hw = (h, ww)
    ctx.set_materialize_grads(False)
    ctx.save_for_backward(x, w, z if (epilogue & TG_EPI_LRELU) else None, bias)
    return z, zp

  @staticmethod
  def backward(ctx, gz, gzp):
    if gz is None and gzp is None:
      return None, None, None, None, None, None
    return _conv_backward(ctx, gz, gzp)


class Conv2dPoolSignsFn(torch.autograd.Function):
  """avg_pool2(z), z = lrelu(conv(x, w) + bias), when the pool is the only consumer of z (the discriminator blocks,
  nets/pggan.py:304-306) and the pass is differentiated once: the forward keeps only the sign bits of z
  (tg_conv2d_fwd_pool_signs: 1/16 of z's bytes written instead of all of them), the backward rebuilds the LeakyReLU
  derivative from them (tg_lrelu_pool_bwd_signs).  Not for create_graph passes (ops.second_order): those keep z."""

  @staticmethod
  def forward(ctx, x, w, bias, spec, epilogue, mask_input=False):
    signs, zp = conv_fwd_pool_signs_raw(x, w, bias, spec, epilogue)
    ctx.mask_input, ctx.spec, ctx.epilogue, ctx.tg_signs = mask_input, spec, epilogue, True
    ctx.out_hw = (signs.shape[1], signs.shape[2])
    ctx.save_for_backward(x, w, signs, bias)
    return zp

  @staticmethod
  def backward(ctx, gzp):
    if torch.is_grad_enabled():
      raise _lib.TgError('Conv2dPoolSignsFn is first order only: run the forward inside ops.second_order() to keep z')
    return _conv_backward(ctx, None, gzp)


class ConvBwdDataFn(torch.autograd.Function):
  """gx = conv^T(gy, w)  (Conv2DBackpropInput); differentiable in gy and w.

  ``out_act`` (optional): the LeakyReLU output of this conv's layer, with which the node that produced gy masks the
  cotangent this node's backward hands it (see MaskedDgradFn): given, the backward applies that mask in the epilogue of
  its conv (tg_conv2d_fwd_masked) and the producer, flagged ``tg_v_premasked`` by the caller, skips its launch."""

  @staticmethod
  def forward(ctx, gy, w, x_shape, spec, out_act=None):
    ctx.spec, ctx.x_shape = spec, x_shape
    ctx.save_for_backward(gy, w, out_act)
    return conv_bwd_data_raw(gy, w, x_shape, spec)

  @staticmethod
  def backward(ctx, v):
    gy, w, out_act = ctx.saved_tensors
    v = v.contiguous()
    ggy = None
    if ctx.needs_input_grad[0]:
      if out_act is not None and not torch.is_grad_enabled():
        ggy = conv_fwd_masked_raw(v, w, out_act, ctx.spec)
      else:
        ggy = Conv2dFn.apply(v, w, None, ctx.spec, 0, False)
        if out_act is not None:      # a third-order pass: keep the premasking contract, differentiably
          ggy = LReluBwdFn.apply(ggy, out_act, ctx.spec.alpha)
    gw = _weight_grad(v, gy, ctx.spec, w) if (ctx.needs_input_grad[1] and not _State.skip_param_grads) else None
    return ggy, gw, None, None, None


# the LeakyReLU mask a node of the gradient penalty's second backward pass applies to its incoming cotangent, moved into
# the conv that PRODUCES that cotangent (tg_conv2d_fwd_masked; USE_GP_PREMASK = False (tests): every node masks for itself, for A/Bs)
USE_GP_PREMASK = True


class MaskedDgradFn(torch.autograd.Function):
  """gx = conv^T(gy, w) * mask(x_act), mask = (x_act > 0 ? 1 : alpha): backward-data with the LeakyReLU backward of the
  layer that produced this conv's input folded into its epilogue, for create_graph passes (the gradient-penalty
  first backward).  Linear in gy and in w, so its own backward is  v' = v * mask(x_act);  d/dgy = conv(v', w),
  d/dw = x'^T-style filter gradient of (v', gy);  the mask is piecewise constant: no gradient to x_act.

  ``out_act`` (optional): the LeakyReLU output of THIS conv's layer.  The node that produced gy -- the masked
  backward-data of the next layer, or the unpool + mask of a block end -- starts its own backward by masking the
  cotangent it receives, which is this node's d/dgy, with exactly that tensor: with out_act given this node's backward
  applies that mask in the epilogue of its conv (tg_conv2d_fwd_masked) and the producer, flagged ``tg_v_premasked`` by
  the caller, skips its LeakyReluGrad launch."""

  @staticmethod
  def forward(ctx, gy, w, x_act, spec, out_act=None):
    ctx.spec = spec
    ctx.save_for_backward(gy, w, x_act, out_act)
    return conv_bwd_data_masked_raw(gy, w, x_act, spec)

  @staticmethod
  def backward(ctx, v):
    gy, w, x_act, out_act = ctx.saved_tensors
    v = v.contiguous()
    vm = v if getattr(ctx, 'tg_v_premasked', False) else LReluBwdFn.apply(v, x_act, ctx.spec.alpha)
    ggy = None
    if ctx.needs_input_grad[0]:
      if out_act is not None and not torch.is_grad_enabled():
        ggy = conv_fwd_masked_raw(vm, w, out_act, ctx.spec)
      else:
        ggy = Conv2dFn.apply(vm, w, None, ctx.spec, 0, False)
        if out_act is not None:      # a third-order pass: keep the premasking contract, differentiably
          ggy = LReluBwdFn.apply(ggy, out_act, ctx.spec.alpha)
    gw = _weight_grad(vm, gy, ctx.spec, w) if (ctx.needs_input_grad[1] and not _State.skip_param_grads) else None
    return ggy, gw, None, None, None


class UnpoolMaskedDgradFn(torch.autograd.Function):
  """gx = conv^T(g, w) * mask(x_act) with g = 0.25 * upsample2(gzp) * mask(z) -- LReluPoolBwdFn followed by MaskedDgradFn as
  ONE launch (tg_conv2d_bwd_data_unpool_act: g is formed while the backward-data kernel stages its tiles, and written once
  for the second differentiation), for the gradient penalty's first backward pass through a discriminator block end
  (image_generation.py:414-439 over nets/pggan.py:304-306).  Linear in gzp and w; its backward is the two nodes' backwards
  in sequence: v' = v * mask(x_act); d/dgzp = avg_pool2(conv(v', w) * mask(z)) (the mask in the conv's epilogue);
  d/dw = filter gradient of (v', g)."""

  @staticmethod
  def forward(ctx, gzp, w, x_act, z, spec):
    gx, g = conv_bwd_data_unpool_raw(gzp.contiguous(), z, w, x_act, tuple(x_act.shape), spec, True)
    ctx.spec = spec
    ctx.save_for_backward(g, w, x_act, z)
    return gx

  @staticmethod
  def backward(ctx, v):
    g, w, x_act, z = ctx.saved_tensors
    v = v.contiguous()
    vm = v if getattr(ctx, 'tg_v_premasked', False) else LReluBwdFn.apply(v, x_act, ctx.spec.alpha)
    ggzp = None
    if ctx.needs_input_grad[0]:
      if not torch.is_grad_enabled():
        t = conv_fwd_masked_raw(vm, w, z, ctx.spec)
      else:      # a third-order pass: differentiably
        t = LReluBwdFn.apply(Conv2dFn.apply(vm, w, None, ctx.spec, 0, False), z, ctx.spec.alpha)
      ggzp = Pool2Fn.apply(t, 0.25)
    gw = _weight_grad(vm, g, ctx.spec, w) if (ctx.needs_input_grad[1] and not _State.skip_param_grads) else None
    return ggzp, gw, None, None, None


def _unpool_act_supported(x_shape, w, spec, dtype):
  d = _desc(x_shape, w.shape[-1], spec, dtype, 0, _wg(w))
  return d.algo == TG_ALGO_MFMA and bool(_lib.load().tg_conv2d_bwd_data_unpool_supported(ctypes.byref(d)))


class ConvBwdWeightFn(torch.autograd.Function):
  """gw = x^T * gy  (Conv2DBackpropFilter), fp32 HWIO.  Third-order terms are not needed by any
  TwinGAN loss, so this node is a leaf of the double-backward graph."""

  @staticmethod
  def forward(ctx, x, gy, spec, groups=1):
    return conv_bwd_weight_raw(x, gy, spec, groups=groups)

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, v):
    raise NotImplementedError('third-order gradient through conv backward-weight')


class LReluBwdFn(torch.autograd.Function):
  """g * (z > 0 ? 1 : alpha); linear in g (its own double backward), zero gradient wrt z."""

  @staticmethod
  def forward(ctx, g, z, alpha):
    ctx.alpha = alpha
    ctx.save_for_backward(z)
    return lrelu_bwd_raw(g, z, alpha)

  @staticmethod
  def backward(ctx, v):
    z, = ctx.saved_tensors
    return LReluBwdFn.apply(v.contiguous(), z, ctx.alpha), None, None


class LReluPoolBwdFn(torch.autograd.Function):
  """g = 0.25 * upsample2(gzp) * (z > 0 ? 1 : alpha): the backward of avg_pool2(lrelu(.)) in one pass, differentiable
  in gzp for create_graph passes (linear: d/dgzp = avg_pool2(v * mask(z)); the mask has no gradient)."""

  @staticmethod
  def forward(ctx, gzp, z, alpha):
    ctx.alpha = alpha
    ctx.save_for_backward(z)
    return lrelu_pool_bwd(None, gzp.contiguous(), z, alpha, None, False)[0]

  @staticmethod
  def backward(ctx, v):
    z, = ctx.saved_tensors
    v = v.contiguous()
    if not getattr(ctx, 'tg_v_premasked', False):      # else: the conv that produced v masked it (MaskedDgradFn out_act)
      v = LReluBwdFn.apply(v, z, ctx.alpha)
    return Pool2Fn.apply(v, 0.25), None, None


class ChannelSumFn(torch.autograd.Function):
  """BiasAddGrad: sum over pixels -> fp32 [C]."""

  @staticmethod
  def forward(ctx, g, groups=1):
    return channel_sum_raw(g, groups)

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, v):
    raise NotImplementedError('gradient through a bias gradient')


def _claim_input_lrelu(x, alpha):
  """``x`` is the LeakyReLU output of a conv node and THIS conv is its only consumer: take over its LeakyReLU
  backward (our backward-data applies the mask in its epilogue; the producer then skips its own mask pass)."""
  node = x.grad_fn
  if node is None or not getattr(node, 'tg_lrelu_out', False) or getattr(node, 'tg_lrelu_alpha', None) != alpha:
    return False
  node.tg_premasked = True
  return True


def conv2d_stats(x, w, k=3, padding='SAME'):
  """(y, ConvStats | None): bias-free conv whose output goes to a normaliser (norm_act(..., conv_stats=...))."""
  holder = []
  y = Conv2dStatsFn.apply(x, w, ConvSpec(k, padding, 0, LRELU_ALPHA), holder)
  return y, holder[0]


def conv2d(x, w, bias=None, k=3, padding='SAME', lrelu=False, alpha=LRELU_ALPHA, pool=False, fuse_input_lrelu=False,
           pool_only=False):
  """Stride-1 conv, optional fused bias and LeakyReLU (discriminator layers).  ``pool``: also return the
  2x2 average-pooled output -> (z, z_pooled).  ``fuse_input_lrelu``: the caller guarantees that ``x`` is consumed by
  this conv only; when x is a conv node's LeakyReLU output, that layer's LeakyReLU backward moves into this conv's
  backward-data epilogue (raw kernel in first-order passes, the differentiable MaskedDgradFn in create_graph passes).
  ``pool_only``: the caller uses nothing but the pooled output -> (None, z_pooled) where the layer can run without
  writing z (Conv2dPoolSignsFn: first-order passes of LeakyReLU layers on the MFMA path), else (z, z_pooled) as usual."""
  spec = ConvSpec(k, padding, 0, alpha)
  epi = (TG_EPI_BIAS if bias is not None else 0) | (TG_EPI_LRELU if lrelu else 0)
  mask_input = bool(fuse_input_lrelu) and _claim_input_lrelu(x, alpha)
  if pool and pool_only and not in_second_order() and conv_fwd_pool_signs_supported(x, w, spec, epi):
    return None, Conv2dPoolSignsFn.apply(x, w, bias, spec, epi, mask_input)
  if pool:
    return Conv2dPoolFn.apply(x, w, bias, spec, epi, mask_input)
  z = Conv2dFn.apply(x, w, bias, spec, epi, mask_input)
  if lrelu and z.grad_fn is not None:
    z.grad_fn.tg_lrelu_out, z.grad_fn.tg_lrelu_alpha = True, alpha
  return z


# ------------------------------------------------------------------------------------------------
# 1x1 convs with a 3-channel side: fromRGB / toRGB (nets/pggan.py:176-178,198-200,233-240,395-399)
# ------------------------------------------------------------------------------------------------
def _pw_fwd_raw(x, w, bias, wt, epilogue, alpha):
  _chk(x, w, bias)
  cin = x.shape[-1]
  cout = w.shape[0] if wt else w.shape[1]
  assert (w.shape[1] if wt else w.shape[0]) == cin, (w.shape, cin, wt)
  y = torch.empty(x.shape[:-1] + (cout,), dtype=x.dtype, device=x.device)
  call('tg_pointwise_conv_fwd', _p(x), _p(w), _p(bias), _p(y), x.numel() // cin, cin, cout, int(wt), epilogue, alpha,
       _dt(x), _stream(), work=('pw_fwd:c%d>%d:px%d' % (cin, cout, x.numel() // cin), 2 * x.numel() * cout, _nb(x, y)))
  return y


def _pw_fwd_masked_raw(x, w, wt, mask_src, alpha):
  """rnd(x @ W) * (mask_src > 0 ? 1 : alpha), x with <= 4 channels (tg_pointwise_conv_fwd_masked)."""
  _chk(x, w, mask_src)
  cin = x.shape[-1]
  cout = w.shape[0] if wt else w.shape[1]
  assert (w.shape[1] if wt else w.shape[0]) == cin and tuple(mask_src.shape) == tuple(x.shape[:-1]) + (cout,), (w.shape, cin, wt)
  y = torch.empty(x.shape[:-1] + (cout,), dtype=x.dtype, device=x.device)
  call('tg_pointwise_conv_fwd_masked', _p(x), _p(w), _p(mask_src), _p(y), x.numel() // cin, cin, cout, int(wt), alpha, _dt(x),
       _stream(), work=('pw_fwd_masked:c%d>%d:px%d' % (cin, cout, x.numel() // cin), 2 * x.numel() * cout, _nb(x, y, mask_src)))
  return y


class PointwiseConvFn(torch.autograd.Function):
  """z = epilogue(x @ W [+ b]) with W = w (wt=False) or w^T (wt=True); w is fp32 [a, b]."""

  @staticmethod
  def forward(ctx, x, w, bias, wt, epilogue, alpha, out_act=None):
    """``out_act`` (optional, x's shape; see MaskedDgradFn): the LeakyReLU output with which the node that produced x masks
    the cotangent this node's backward hands it -- given, the backward applies that mask itself
    (tg_pointwise_conv_fwd_masked) and the producer, flagged ``tg_v_premasked`` by the caller, skips its launch."""
    z = _pw_fwd_raw(x, w, bias, wt, epilogue, alpha)
    ctx.wt, ctx.epilogue, ctx.alpha = wt, epilogue, alpha
    ctx.save_for_backward(x, w, z if (epilogue & TG_EPI_LRELU) else None, bias, out_act)
    return z

  @staticmethod
  def backward(ctx, gz):
    x, w, z, bias, out_act = ctx.saved_tensors
    gz = gz.contiguous()
    params = not _State.skip_param_grads
    need_b = bool(ctx.epilogue & TG_EPI_BIAS) and ctx.needs_input_grad[2] and params
    gb = None
    if (ctx.epilogue & TG_EPI_LRELU) and getattr(ctx, 'tg_premasked', False):
      g = gz      # the consumer conv's backward-data already applied this layer's LeakyReLU mask
    elif ctx.epilogue & TG_EPI_LRELU:
      if need_b and not torch.is_grad_enabled():
        g, gb = lrelu_pool_bwd(gz, None, z.reshape(-1, 1, 1, z.shape[-1]) if z.dim() != 4 else z, ctx.alpha, bias, True)
        g = g.reshape(gz.shape)
        need_b = False
      else:
        g = LReluBwdFn.apply(gz, z, ctx.alpha)
    else:
      g = gz
    gx = None
    if _input_grad_wanted(ctx, x):
      if out_act is not None and not torch.is_grad_enabled() and g.shape[-1] <= 4:
        gx = _pw_fwd_masked_raw(g, w, not ctx.wt, out_act, ctx.alpha)
      else:
        # the node that produced g (the first block's masked backward-data in the gradient penalty's first backward) masks
        # the cotangent it gets back from us with THIS layer's LeakyReLU output: when we are its only consumer our own
        # backward applies that mask in its epilogue and tells the producer so (as _conv_backward does for the convs)
        node = g.grad_fn
        premask = (torch.is_grad_enabled() and USE_GP_PREMASK and _State.skip_param_grads and z is not None
                   and out_act is None and node is not None
                   and getattr(node, 'tg_masks_with', None) == (z.data_ptr(), tuple(z.shape))
                   and not getattr(node, 'tg_v_premasked', False))
        gx = PointwiseConvFn.apply(g, w, None, not ctx.wt, 0, ctx.alpha, z if premask else None)
        if premask:
          node.tg_v_premasked = True
        if out_act is not None:      # a pass this node cannot fuse (third order, or the wide side in): mask differentiably
          gx = LReluBwdFn.apply(gx, out_act, ctx.alpha)
    gw = None
    if ctx.needs_input_grad[1] and params:
      # y = x @ W: dW = x^T g.  With wt the stored tensor is W^T, so dw = g^T x.
      a, b = (g, x) if ctx.wt else (x, g)
      sink = GradSink.get(w)
      bsink = GradSink.get(bias) if need_b else None
      if sink is not None and bsink is not None and not ctx.wt and a.shape[-1] <= 4 and not deterministic():
        # fromRGB: the bias gradient (pixel sums of g) from the filter gradient's own read of g
        ca, cb = a.shape[-1], b.shape[-1]
        call('tg_pointwise_conv_bwd_weight_bias', _p(a), _p(b), _p(sink), _p(bsink), a.numel() // ca, ca, cb, 1, _dt(a), _stream(),
             work=('pw_wgrad:c%d>%d:px%d' % (ca, cb, a.numel() // ca), 2 * a.numel() * cb, _nb(a, b)))
        need_b = False
      elif sink is not None:
        _pw_wgrad_into(a, b, sink, True)
      else:
        gw = PointwiseWgradFn.apply(a, b)
    if need_b:
      gb = _bias_grad(g, bias)
    return gx, gw, gb, None, None, None, None


def _pw_wgrad_into(a, b, out, accumulate):
  """out[ca, cb] (+)= a^T b over pixels (one of ca, cb <= 4); the ordered two-stage form in deterministic mode."""
  ca, cb = a.shape[-1], b.shape[-1]
  work = ('pw_wgrad:c%d>%d:px%d' % (ca, cb, a.numel() // ca), 2 * a.numel() * cb, _nb(a, b))
  if deterministic():
    ws = torch.empty(256 * ca * cb, dtype=torch.float32, device=a.device)
    call('tg_pointwise_conv_bwd_weight_ordered', _p(a), _p(b), _p(out), a.numel() // ca, ca, cb, 1 if accumulate else 0,
         _p(ws), ws.numel(), _dt(a), _stream(), work=work)
  else:
    call('tg_pointwise_conv_bwd_weight', _p(a), _p(b), _p(out), a.numel() // ca, ca, cb, 1 if accumulate else 0, _dt(a),
         _stream(), work=work)


class PointwiseWgradFn(torch.autograd.Function):
  """a^T b over pixels -> fp32 [ca, cb] (one of ca, cb <= 4)."""

  @staticmethod
  def forward(ctx, a, b):
    _chk(a, b)
    ca, cb = a.shape[-1], b.shape[-1]
    out = torch.empty((ca, cb), dtype=torch.float32, device=a.device)
    _pw_wgrad_into(a, b, out, False)
    return out

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, v):
    raise NotImplementedError('third-order gradient through pointwise weight gradient')


def pointwise_conv(x, w_hwio, bias=None, lrelu=False, alpha=LRELU_ALPHA):
  """1x1 conv with w [1,1,cin,cout] where cin <= 4 or cout <= 4."""
  w2 = w_hwio.view(w_hwio.shape[2], w_hwio.shape[3])
  epi = (TG_EPI_BIAS if bias is not None else 0) | (TG_EPI_LRELU if lrelu else 0)
  z = PointwiseConvFn.apply(x, w2, bias, False, epi, alpha)
  if lrelu and z.grad_fn is not None:
    z.grad_fn.tg_lrelu_out, z.grad_fn.tg_lrelu_alpha = True, alpha
  return z


# ------------------------------------------------------------------------------------------------
# instance norm + LeakyReLU + pixel norm (generator / encoder layers)
# ------------------------------------------------------------------------------------------------
def _ema_update(mean, rstd, n, c, split, eps, ema):
  """BatchNorm moving statistics (libs/batch_norm.py:283-300): one assign_moving_average per batched pass
  (= per statistic group), into the pass' domain variables.  ema = (decay, [(moving_mean, moving_var) per domain])."""
  decay, pairs = ema
  with torch.no_grad():
    m = mean.view(n, c)
    v = rstd.view(n, c).pow(-2) - eps
    for gi in range(n):
      mm, mv = pairs[0 if gi < split else 1]
      mm.sub_((mm - m[gi]) * (1.0 - decay))
      mv.sub_((mv - v[gi]) * (1.0 - decay))


def instance_stats(y, eps):
  """(mean, rstd), fp32 [n*c], of y[n,h,w,c] over (h,w): biased variance, rstd = rsqrt(var + eps).  No autograd."""
  _chk(y)
  n, h, w, c = y.shape
  mean = torch.empty(n * c, dtype=torch.float32, device=y.device)
  rstd = torch.empty(n * c, dtype=torch.float32, device=y.device)
  call('tg_instance_norm_stats', _p(y), _p(mean), _p(rstd), n, h, w, c, eps, _dt(y), _stream(),
       work=('in_stats' + _shape_tag(y), 0, y.numel() * _esize(y)))
  return mean, rstd


def _norm_act_forward(ctx, y, gamma, beta, gamma2, beta2, split, flags, in_eps, pn_eps, alpha, ema=None, stats=None,
                      zp=None, conv_stats=None):
  _chk(y, gamma, beta, gamma2, beta2)
  n, h, w, c = y.shape
  split = n if gamma2 is None else int(split)
  per_image = gamma.dim() == 2          # [n, c] parameter rows (batch renorm): statistics come from the caller
  z = torch.empty_like(y)
  s = torch.empty(n * h * w, dtype=torch.float32, device=y.device) if (flags & NF_PIXNORM) else None
  if per_image:
    assert stats is not None and tuple(gamma.shape) == (n, c) and tuple(beta.shape) == (n, c) and gamma2 is None
    mean, rstd = stats
    call('tg_norm_act_fwd', _p(y), _p(mean), _p(rstd), _p(gamma), _p(beta), 0, 0, split, 1, _p(z), _p(s), n, h, w, c,
         flags, alpha, pn_eps, _dt(y), _stream(), work=('norm_act_fwd' + _shape_tag(y), 0, 2 * y.numel() * _esize(y)))
  else:
    mean = torch.empty(n * c, dtype=torch.float32, device=y.device)
    rstd = torch.empty(n * c, dtype=torch.float32, device=y.device)
    fwd_work = ('norm_act_fwd' + _shape_tag(y), 0, int((2 + (0.25 if zp is not None else 0)) * y.numel()) * _esize(y))
    if conv_stats is not None:      # the producing conv already summed its outputs per workgroup
      assert conv_stats.part.numel() == n * conv_stats.chunks * 2 * c
      call('tg_norm_act_fwd_conv_stats', _p(y), _p(conv_stats.part), conv_stats.chunks, _p(mean), _p(rstd), _p(gamma),
           _p(beta), _p(gamma2), _p(beta2), split, _p(z), _p(zp), _p(s), n, h, w, c, flags, alpha, in_eps, pn_eps, _dt(y),
           _stream(), work=fwd_work)
    else:
      chunks = _lib.load().tg_norm_chunks(n, h, w)
      part = torch.empty(n * chunks * 2 * c, dtype=torch.float32, device=y.device)
      call('tg_instance_norm_partials', _p(y), _p(part), n, h, w, c, _dt(y), _stream(),
           work=('in_stats' + _shape_tag(y), 0, y.numel() * _esize(y)))
      call('tg_norm_act_fwd_partials', _p(y), _p(part), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(gamma2), _p(beta2), split,
           _p(z), _p(zp), _p(s), n, h, w, c, flags, alpha, in_eps, pn_eps, _dt(y), _stream(), work=fwd_work)
    if ema is not None:
      _ema_update(mean, rstd, n, c, split, in_eps, ema)
  ctx.flags, ctx.alpha, ctx.split, ctx.per_image = flags, alpha, split, per_image
  ctx.save_for_backward(y, mean, rstd, gamma, beta, gamma2, beta2, s)
  return z


def _norm_act_backward(ctx, gz, gzp=None):
  y, mean, rstd, gamma, beta, gamma2, beta2, s = ctx.saved_tensors
  gz = gz.contiguous() if gz is not None else None
  gzp = gzp.contiguous() if gzp is not None else None
  n, h, w, c = y.shape
  gy = torch.empty_like(y)
  sums = torch.empty(n * _lib.load().tg_norm_chunks(n, h, w) * 2 * c, dtype=torch.float32, device=y.device)
  two = gamma2 is not None
  params = [gamma, beta] + ([gamma2, beta2] if two else [])
  per_image = ctx.per_image
  sinks = [None if per_image else GradSink.get(q) for q in params]
  sunk = all(t is not None for t in sinks)
  if _State.skip_param_grads:
    outs = [None] * 4
  elif per_image:
    outs = [torch.empty((n, c), dtype=torch.float32, device=y.device) for _ in params] + [None, None]
  elif sunk:
    outs = sinks + [None] * (4 - len(sinks))
  else:
    outs = [torch.empty(c, dtype=torch.float32, device=y.device) for _ in params] + [None] * (4 - len(params))
  passes = 3 + (0.25 if gzp is not None else 0) - (0 if gz is not None else 1)
  call('tg_norm_act_bwd', _p(gz), _p(gzp), _p(y), _p(s), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(gamma2), _p(beta2),
       ctx.split, _p(gy), 1 if per_image else 0, _p(outs[0]), _p(outs[1]), _p(outs[2]), _p(outs[3]), _p(sums), n, h, w, c,
       ctx.flags, ctx.alpha,
       1 if (sunk and not _State.skip_param_grads) else 0, _dt(y), _stream(),
       work=('norm_act_bwd' + _shape_tag(y), 0, int(passes * y.numel()) * _esize(y)))
  if sunk or _State.skip_param_grads:
    outs = [None] * 4
  return gy, outs[0], outs[1], outs[2], outs[3], None, None, None, None, None, None


class NormActFn(torch.autograd.Function):
  """z = pixel_norm(lrelu(instance_norm(y; gamma, beta))) -- libs/instance_norm.py:131-135,
  util_misc.py:86, nets/pggan_utils.py:330-331.  First-order only (E/G never sit under the
  gradient penalty).  With (gamma2, beta2, split) images [split, n) use the second domain's parameters."""

  @staticmethod
  def forward(ctx, y, gamma, beta, gamma2, beta2, split, flags, in_eps, pn_eps, alpha, ema, stats, conv_stats=None):
    return _norm_act_forward(ctx, y, gamma, beta, gamma2, beta2, split, flags, in_eps, pn_eps, alpha, ema, stats,
                             conv_stats=conv_stats)

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, gz):
    return _norm_act_backward(ctx, gz) + (None, None)


class NormActPoolFn(torch.autograd.Function):
  """(z, avg_pool2(z)) for the last layer of an encoder block (nets/pggan.py:466-468): z is the UNet skip
  end-point, the pooled tensor feeds the next block.  The backward takes both gradients and folds the pool
  (and the sum of the two) into the normalisation backward kernel."""

  @staticmethod
  def forward(ctx, y, gamma, beta, gamma2, beta2, split, flags, in_eps, pn_eps, alpha, ema, stats, conv_stats=None):
    n, h, w, c = y.shape
    zp = torch.empty((n, h // 2, w // 2, c), dtype=y.dtype, device=y.device)
    fused = gamma.dim() == 1      # the partial-sums forward writes the pooled tensor itself
    z = _norm_act_forward(ctx, y, gamma, beta, gamma2, beta2, split, flags, in_eps, pn_eps, alpha, ema, stats,
                          zp if fused else None, conv_stats=conv_stats if fused else None)
    if not fused:
      call('tg_pool2x2_fwd', _p(z), _p(zp), n, h, w, c, 0.25, _dt(z), _stream(),
           work=('pool_fwd' + _shape_tag(z), 0, int(1.25 * z.numel()) * _esize(z)))
    ctx.set_materialize_grads(False)
    return z, zp

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, gz, gzp):
    if gz is None and gzp is None:
      return (None,) * 13
    return _norm_act_backward(ctx, gz, gzp) + (None, None)


_CONST = {}


def pixel_norm(y, pn_eps=1e-6):
  """x / sqrt(mean_C(x^2) + eps) alone (nets/pggan_utils.py:330-331; --generator_norm_type none): the fused
  normalisation kernel with constant unit statistics and parameters; NF_NOSTATS makes its backward treat them so."""
  n, c = y.shape[0], y.shape[3]
  key = (n, c, y.device)
  if key not in _CONST:
    one = torch.ones(n * c, dtype=torch.float32, device=y.device)
    _CONST[key] = (one, torch.zeros_like(one))
  one, zero = _CONST[key]
  return NormActFn.apply(y, one.view(n, c), zero.view(n, c), None, None, None, NF_PIXNORM | NF_NOSTATS, 0.0, pn_eps,
                         LRELU_ALPHA, None, (zero, one))


def affine_act(y_hat, gamma_rows, beta_rows, lrelu=True, pixel_norm=True, pool=False, pn_eps=1e-6, alpha=LRELU_ALPHA):
  """pixel_norm(lrelu(y_hat * gamma_rows[n] + beta_rows[n])) with one parameter row per image ([n, c], ordinary
  differentiable tensors): the tail of the CONDITIONAL batch (re)norm layers, whose statistics belong to the whole pass
  and are applied before (libs/batch_norm.py:403-470) -- the fused kernel with constant unit statistics (NF_NOSTATS)."""
  n, c = y_hat.shape[0], y_hat.shape[3]
  key = (n, c, y_hat.device)
  if key not in _CONST:
    one = torch.ones(n * c, dtype=torch.float32, device=y_hat.device)
    _CONST[key] = (one, torch.zeros_like(one))
  one, zero = _CONST[key]
  flags = (NF_LRELU if lrelu else 0) | (NF_PIXNORM if pixel_norm else 0) | NF_NOSTATS
  fn = NormActPoolFn if pool else NormActFn
  return fn.apply(y_hat, gamma_rows.contiguous(), beta_rows.contiguous(), None, None, None, flags, 0.0, pn_eps, alpha, None,
                  (zero, one))


_MOMENT_EPS = 1e-12      # only keeps rsqrt finite for a constant channel; removed again from the variance


def _unit_rows(n, c, device):
  key = (n, c, device)
  if key not in _CONST:
    one = torch.ones(n * c, dtype=torch.float32, device=device)
    _CONST[key] = (one, torch.zeros_like(one))
  return _CONST[key]


class ChannelMomentsFn(torch.autograd.Function):
  """(mean, var), fp32 [n, c], of y[n, h, w, c] over (h, w) (population variance) as a differentiable node: the statistics
  kernel forward; backward d/dy = gm / hw + gv * 2 (y - mean) / hw = y * a[n, c] + b[n, c], ONE launch of the fused
  normalisation kernel in its per-image-row mode with constant unit statistics.  First-order only (E / G)."""

  @staticmethod
  def forward(ctx, y):
    n, h, w, c = y.shape
    mean, rstd = instance_stats(y, _MOMENT_EPS)
    mean = mean.view(n, c)
    var = (rstd.view(n, c).pow(-2) - _MOMENT_EPS).clamp_min_(0.0)
    ctx.save_for_backward(y, mean)
    return mean, var

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, gm, gv):
    y, mean = ctx.saved_tensors
    n, h, w, c = y.shape
    inv = 1.0 / (h * w)
    a = torch.zeros((n, c), dtype=torch.float32, device=y.device) if gv is None else gv.float() * (2.0 * inv)
    b = -a * mean
    if gm is not None:
      b = b + gm.float() * inv
    one, zero = _unit_rows(n, c, y.device)
    gy = torch.empty_like(y)
    call('tg_norm_act_fwd', _p(y), _p(zero), _p(one), _p(a.contiguous()), _p(b.contiguous()), 0, 0, n, 1, _p(gy), 0, n, h, w, c,
         NF_NOSTATS, LRELU_ALPHA, 1e-6, _dt(y), _stream(), work=('norm_act_fwd' + _shape_tag(y), 0, 2 * y.numel() * _esize(y)))
    return gy


def layer_norm_act(y, gamma, beta, lrelu=True, pixel_norm=True, ln_eps=1e-12, pn_eps=1e-6, alpha=LRELU_ALPHA, gamma2=None,
                   beta2=None, split=None, pool=False):
  """pixel_norm(lrelu(layer_norm(y; gamma, beta))): tf.contrib.layers.layer_norm as 'layer_norm_native' calls it
  (nets/pggan_utils.py:189-197; TF 1.8: moments of each image over (H, W, C), gamma / beta per channel, epsilon 1e-12).
  An option row, built from the kernels of the headline path: per-(image, channel) moments (the statistics kernel), the
  image's mean / variance from them by the law of total variance ([n, c] row arithmetic, no cancellation), and the fused
  per-image-row kernel with scale = gamma * rstd_n, shift = beta - mean_n * scale (affine + LeakyReLU + pixel norm + pool
  in one pass, its backward returning the row gradients) -- two tensor passes more than a dedicated kernel would need
  (the statistics pass and its backward).  Images [split, n) use (gamma2, beta2)."""
  n, c = y.shape[0], y.shape[3]
  m_c, v_c = ChannelMomentsFn.apply(y)
  mu = m_c.mean(dim=1, keepdim=True)
  var = (v_c + (m_c - mu) ** 2).mean(dim=1, keepdim=True)
  rstd = torch.rsqrt(var + ln_eps)
  if gamma2 is None:
    g_rows, b_rows = gamma.float().expand(n, c), beta.float().expand(n, c)
  else:
    sp = int(split)
    g_rows = torch.cat([gamma.float().expand(sp, c), gamma2.float().expand(n - sp, c)])
    b_rows = torch.cat([beta.float().expand(sp, c), beta2.float().expand(n - sp, c)])
  scale_rows = g_rows * rstd
  return affine_act(y, scale_rows, b_rows - mu * scale_rows, lrelu=lrelu, pixel_norm=pixel_norm, pool=pool, pn_eps=pn_eps,
                    alpha=alpha)


def norm_act(y, gamma, beta, lrelu=True, pixel_norm=True, in_eps=1e-6, pn_eps=1e-6, alpha=LRELU_ALPHA, gamma2=None,
             beta2=None, split=None, pool=False, ema=None, stats=None, conv_stats=None):
  """Statistics are per leading index of ``y`` (instance norm: one image; batch norm: the caller passes the view
  [passes, B*H, W, C] so that each batched pass is one statistic group).  ``pool``: also return the 2x2
  average-pooled output -> (z, z_pooled).  ``ema``: (decay, [(moving_mean, moving_var) per domain]) to update.
  Batch renorm: ``gamma`` / ``beta`` are [n, c] (one effective r*gamma, d*gamma+beta row per statistic group,
  ordinary differentiable tensors) and ``stats`` = instance_stats(y, eps) computed by the caller."""
  flags = (NF_LRELU if lrelu else 0) | (NF_PIXNORM if pixel_norm else 0)
  fn = NormActPoolFn if pool else NormActFn
  if conv_stats is not None and (gamma.dim() != 1 or stats is not None):
    conv_stats = None      # per-image parameter rows take their statistics from the caller
  return fn.apply(y, gamma, beta, gamma2, beta2, split, flags, in_eps, pn_eps, alpha, ema, stats, conv_stats)


# ------------------------------------------------------------------------------------------------
# resampling / concat / fade-in
# ------------------------------------------------------------------------------------------------
def _pack_perm(perm):
  v = 0
  for k, g in enumerate(perm):
    v |= (int(g) & 0xff) << (8 * k)
  return v


class UpsampleConcatFn(torch.autograd.Function):
  """concat(nearest_up2(x0), x1) on C (nets/pggan_utils.py:349-350, 281-298).  With (gsz, perm) output group k
  (gsz images) reads skip group perm[k] of x1 -- several generator passes batched along N share the encoder's
  skip tensors without copies."""

  @staticmethod
  def forward(ctx, x0, x1, gsz, perm):
    _chk(x0, x1)
    n, h, w, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[3]
    if gsz:
      assert x1 is not None and n % gsz == 0 and len(perm) == n // gsz and x1.shape[0] == (max(perm) + 1) * gsz
    out = torch.empty((n, 2 * h, 2 * w, c0 + c1), dtype=x0.dtype, device=x0.device)
    pk = _pack_perm(perm) if gsz else 0
    call('tg_upsample2x_concat_fwd', _p(x0), _p(x1), _p(out), n, h, w, c0, c1, gsz, pk, _dt(x0), _stream(),
         work=('upcat_fwd' + _shape_tag(out), 0, (x0.numel() + (x1.numel() if x1 is not None else 0) + out.numel()) * _esize(out)))
    ctx.dims = (n, h, w, c0, c1, gsz, pk, 0 if x1 is None else x1.shape[0])
    return out

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, go):
    n, h, w, c0, c1, gsz, pk, n1 = ctx.dims
    go = go.contiguous()
    g0 = torch.empty((n, h, w, c0), dtype=go.dtype, device=go.device) if ctx.needs_input_grad[0] else None
    g1 = torch.empty((n1, 2 * h, 2 * w, c1), dtype=go.dtype, device=go.device) \
        if (c1 and ctx.needs_input_grad[1]) else None
    call('tg_upsample2x_concat_bwd', _p(go), _p(g0), _p(g1), n, h, w, c0, c1, gsz, pk, _dt(go), _stream(),
         work=('upcat_bwd' + _shape_tag(go), 0, (go.numel() + g0.numel() + (g1.numel() if g1 is not None else 0)) * _esize(go)))
    return g0, g1, None, None


def upcat_conv_supported(x0, x1, w):
  """Can conv3x3(concat(up2(x0), x1), w) run from the two sources without materialising the concat?"""
  if x1 is None or x0.dtype not in HALF_TYPES or x1.dtype != x0.dtype or w.shape[0] != 3:
    return False
  return bool(_lib.load().tg_conv2d_upcat_supported(2 * x0.shape[1], 2 * x0.shape[2], x0.shape[3], x1.shape[3], w.shape[3]))


class UpcatConvFn(torch.autograd.Function):
  """y = conv3x3_same(concat(nearest_up2(x0), x1), w): the first conv of generator_three_layer_block
  (nets/pggan.py:69-78) reading its input from the two source tensors (tg_conv2d_upcat_fwd / _bwd_weight); the
  concatenated tensor is never written.  The input gradient is the ordinary backward-data over a temporary in concat
  layout, split by tg_upsample2x_concat_bwd.  First-order only (the generator never sits under the gradient penalty)."""

  @staticmethod
  def forward(ctx, x0, x1, w, gsz, perm, holder=None):
    _chk(x0, x1, w)
    n, h, ww, c0 = x0.shape
    c1, cout = x1.shape[3], w.shape[3]
    if gsz:
      assert n % gsz == 0 and len(perm) == n // gsz and x1.shape[0] == (max(perm) + 1) * gsz
    H, W = 2 * h, 2 * ww
    spec = ConvSpec(3, 'SAME')
    d = _desc((n, H, W, c0 + c1), cout, spec, x0.dtype, 0)
    y = torch.empty((n, H, W, cout), dtype=x0.dtype, device=x0.device)
    pk = _pack_perm(perm) if gsz else 0
    work = lambda: ('fwd:upcat:k3:c%d+%d>%d:hw%d:n%d' % (c0, c1, cout, H, n), 2 * n * H * W * cout * 9 * (c0 + c1),    # noqa: E731
                    2 * (x0.numel() + x1.numel() + y.numel()) + 2 * w.numel())
    chunks = _lib.load().tg_conv2d_upcat_fwd_stats_chunks(n, H, W, c0, c1, cout) if (holder is not None and USE_CONV_STATS) else 0
    if chunks > 0:
      part = torch.empty(n * chunks * 2 * cout, dtype=torch.float32, device=x0.device)
      wk = PackCache.get(w, d, 0)      # a local keeps an uncached pack alive across the launch
      call('tg_conv2d_upcat_fwd_stats', _p(x0), _p(x1), _p(wk), _p(y), _p(part), chunks, n, H, W, c0, c1,
           cout, gsz, pk, _dt(x0), _stream(), work=work)
      holder.append(ConvStats(part, chunks))
    else:
      wk = PackCache.get(w, d, 0)      # a local keeps an uncached pack alive across the launch
      call('tg_conv2d_upcat_fwd', _p(x0), _p(x1), _p(wk), _p(y), n, H, W, c0, c1, cout, gsz, pk,
           _dt(x0), _stream(), work=work)
      if holder is not None:
        holder.append(None)
    ctx.dims = (n, H, W, c0, c1, cout, gsz, pk, x1.shape[0])
    ctx.spec = spec
    ctx.save_for_backward(x0, x1, w)
    return y

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, gy):
    x0, x1, w = ctx.saved_tensors
    n, H, W, c0, c1, cout, gsz, pk, n1 = ctx.dims
    gy = gy.contiguous()
    g0 = g1 = gw = None
    if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
      g0 = torch.empty_like(x0) if ctx.needs_input_grad[0] else None
      g1 = torch.empty_like(x1) if ctx.needs_input_grad[1] else None
      if USE_UPCAT_BWD_FUSED and H >= 16:
        # backward-data with the upsample / concat adjoint in its epilogue: no concat-layout gradient tensor
        d = _desc((n, H, W, c0 + c1), cout, ctx.spec, gy.dtype, 0)
        wk = PackCache.get(w, d, 1)      # a local keeps an uncached pack alive across the launch
        call('tg_conv2d_upcat_bwd_data', _p(gy), _p(wk), _p(g0), _p(g1), n, H, W, c0, c1, cout, gsz, pk,
             _dt(gy), _stream(),
             work=lambda: ('dgrad:upcat:k3:c%d>%d+%d:hw%d:n%d' % (cout, c0, c1, H, n), 2 * n * H * W * cout * 9 * (c0 + c1),
                           2 * (gy.numel() + x0.numel() + x1.numel()) + 2 * w.numel()))
      else:
        gcat = conv_bwd_data_raw(gy, w, (n, H, W, c0 + c1), ctx.spec)
        call('tg_upsample2x_concat_bwd', _p(gcat), _p(g0), _p(g1), n, H // 2, W // 2, c0, c1, gsz, pk, _dt(gcat), _stream(),
             work=('upcat_bwd' + _shape_tag(gcat), 0, (gcat.numel() + x0.numel() + x1.numel()) * _esize(gcat)))
    if ctx.needs_input_grad[2] and not _State.skip_param_grads:
      sink = GradSink.get(w)
      gw = sink if sink is not None else torch.empty(tuple(w.shape), dtype=torch.float32, device=gy.device)

      def wgrad(gw=gw):
        lib = _lib.load()
        nbytes = lib.tg_conv2d_upcat_bwd_weight_workspace(n, H, W, c0, c1, cout)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=gy.device)
        call('tg_conv2d_upcat_bwd_weight', _p(x0), _p(x1), _p(gy), _p(gw), 1 if sink is not None else 0, _p(ws), nbytes,
             n, H, W, c0, c1, cout, gsz, pk, _dt(gy), _stream(),
             work=lambda: ('wgrad:upcat:k3:c%d+%d>%d:hw%d:n%d' % (c0, c1, cout, H, n), 2 * n * H * W * cout * 9 * (c0 + c1),
                           2 * (x0.numel() + x1.numel() + gy.numel()) + 4 * w.numel()))
        _keep_for_aux(ws, sink is not None)
      wgrad()
      if sink is not None:
        gw = None
    return g0, g1, gw, None, None, None


def upcat_conv(x0, x1, w, gsz=0, perm=()):
  return UpcatConvFn.apply(x0, x1, w, gsz, tuple(perm))


def upcat_conv_stats(x0, x1, w, gsz=0, perm=()):
  """(y, ConvStats | None), as conv2d_stats."""
  holder = []
  y = UpcatConvFn.apply(x0, x1, w, gsz, tuple(perm), holder)
  return y, holder[0]


def upsample2x_concat(x0, x1=None, gsz=0, perm=()):
  return UpsampleConcatFn.apply(x0, x1, int(gsz), tuple(perm))


class Pool2Fn(torch.autograd.Function):
  """scale * (2x2 block sum), stride 2 (avg-pool with scale .25)."""

  @staticmethod
  def forward(ctx, x, scale):
    _chk(x)
    n, h, w, c = x.shape
    y = torch.empty((n, h // 2, w // 2, c), dtype=x.dtype, device=x.device)
    call('tg_pool2x2_fwd', _p(x), _p(y), n, h, w, c, scale, _dt(x), _stream(),
         work=('pool_fwd' + _shape_tag(x), 0, _nb(x, y)))
    ctx.scale, ctx.hw = scale, (h, w)
    return y

  @staticmethod
  def backward(ctx, gy):
    return Pool2BwdFn.apply(gy.contiguous(), ctx.scale, ctx.hw), None


class Pool2BwdFn(torch.autograd.Function):
  """scale * gy replicated 2x2 (adjoint of Pool2Fn; also the plain upsample with scale 1)."""

  @staticmethod
  def forward(ctx, gy, scale, hw):
    _chk(gy)
    n, _, _, c = gy.shape
    gx = torch.empty((n, hw[0], hw[1], c), dtype=gy.dtype, device=gy.device)
    call('tg_pool2x2_bwd', _p(gy), _p(gx), n, hw[0], hw[1], c, scale, _dt(gy), _stream(),
         work=('pool_bwd' + _shape_tag(gx), 0, _nb(gy, gx)))
    ctx.scale = scale
    return gx

  @staticmethod
  def backward(ctx, v):
    return Pool2Fn.apply(v.contiguous(), ctx.scale), None, None


def avg_pool2(x):
  return Pool2Fn.apply(x, 0.25)


class AxpbyFn(torch.autograd.Function):
  """a*x + b*y (fade-in lerp, nets/pggan.py:205,314,475)."""

  @staticmethod
  def forward(ctx, x, y, a, b):
    _chk(x, y)
    out = torch.empty_like(x)
    call('tg_axpby', _p(x), _p(y), _p(out), x.numel(), a, b, _dt(x), _stream(),
         work=('axpby:numel%d' % x.numel(), 0, _nb(x, y, out)))
    ctx.a, ctx.b, ctx.has_y = a, b, y is not None
    ctx.set_materialize_grads(False)      # no gradient, no work (see first_order_only)
    return out

  @staticmethod
  def backward(ctx, g):
    if g is None:
      return None, None, None, None
    g = g.contiguous()
    gx = AxpbyFn.apply(g, None, ctx.a, 0.0) if ctx.needs_input_grad[0] else None
    gy = AxpbyFn.apply(g, None, ctx.b, 0.0) if (ctx.has_y and ctx.needs_input_grad[1]) else None
    return gx, gy, None, None


def scale(x, a):
  """a * x (the input scaling of maybe_equalized_conv2d / maybe_equalized_fc, nets/pggan_utils.py:236-254)."""
  y = AxpbyFn.apply(x, None, float(a), 0.0)
  return first_order_only(y) if getattr(x, '_tg_first_order_only', False) else y


def add(x, y):
  """x + y (residual shortcut, nets/pggan_utils.py:261)."""
  return AxpbyFn.apply(x, y, 1.0, 1.0)


def lerp(new, old, alpha):
  """new * alpha + (1 - alpha) * old."""
  return AxpbyFn.apply(new, old, float(alpha), float(1.0 - alpha))


class GDropFn(torch.autograd.Function):
  """libs/gdrop.py:20-36 (mode 'prop'): x * (noise * strength * sqrt(C) + 1), noise [N, C] fp32 ~ N(0, 1) per image and
  channel (rnd_shape [N, 1, 1, C]).  ``strength``: a python float or a fp32 device scalar (the `gdrop_strength` variable).
  Linear in x with a constant factor, so the backward -- and the backward of that, under the gradient penalty -- is the
  same node applied to the incoming gradient."""

  @staticmethod
  def forward(ctx, x, noise, strength, c_logical):
    _chk(x, noise)
    n, c = x.shape[0], x.shape[-1]
    assert noise.dtype == torch.float32 and tuple(noise.shape) == (n, c), (tuple(noise.shape), n, c)
    out = torch.empty_like(x)
    dev = strength if isinstance(strength, torch.Tensor) else None
    call('tg_gdrop', _p(x), _p(noise), _p(dev), 0.0 if dev is not None else float(strength), int(c_logical), _p(out), n,
         x.numel() // (n * c), c, _dt(x), _stream(), work=('gdrop' + _shape_tag(x), 0, _nb(x, out)))
    ctx.noise, ctx.strength, ctx.c_logical = noise, strength, c_logical
    return out

  @staticmethod
  def backward(ctx, g):
    return GDropFn.apply(g.contiguous(), ctx.noise, ctx.strength, ctx.c_logical), None, None, None


def gdrop(x, strength, noise=None, c_logical=None):
  """ops.gdrop of the reference (libs/ops.py:31) on an NHWC tensor; ``noise`` [N, C] (drawn on the device when None)."""
  n, c = x.shape[0], x.shape[-1]
  if noise is None:
    noise = torch.randn(n, c, dtype=torch.float32, device=x.device)
  return GDropFn.apply(x.contiguous(), noise.contiguous(), strength, c_logical or c)


def uniform(n, seed, state, lo=0.0, hi=1.0):
  """-> fp32 [n], lo + (hi - lo) * U[0, 1): Philox4x32-10 keyed by (``seed``, state[0]).  ``state``: int32 [2] device tensor,
  zero at the start; the kernel advances state[0] (tg_uniform)."""
  assert state.dtype == torch.int32 and state.numel() == 2
  out = torch.empty(int(n), dtype=torch.float32, device=state.device)
  call('tg_uniform', _p(out), int(n), int(seed) & 0xffffffffffffffff, _p(state), float(lo), float(hi), _stream())
  return out


class _RowsJob(ctypes.Structure):
  _fields_ = [('src', ctypes.c_void_p * 4), ('dst_off', ctypes.c_int64), ('numel', ctypes.c_int64)]


_ROWS_MAX_JOBS = 8      # TG_ROWS_MAX_JOBS


def assemble_rows(dst, blocks):
  """dst[lo : lo + rows] = the sum of ``sources`` (contiguous [rows, ...] tensors of dst's dtype; none: zeros) for every
  (lo, rows, sources) of ``blocks`` -- tg_rows_assemble: one launch per eight blocks, whatever the mix of copies, sums and
  zero fills.  More than four sources of a block chain through dst."""
  assert dst.is_contiguous()
  per = dst[0].numel() if dst.shape[0] else 0
  waves, keep = [[]], []      # a block with more than four sources continues in a LATER launch: its jobs read what the first wrote
  for lo, rows, sources in blocks:
    if rows == 0:
      continue
    src = []
    for t in sources:
      assert t.dtype == dst.dtype and t.numel() == rows * per, (t.shape, t.dtype, dst.shape, dst.dtype, rows)
      t = t if t.is_contiguous() else t.contiguous()
      keep.append(t)
      src.append(t)
    wave = 0
    while wave == 0 or src:
      head, src = src[:3 if wave else 4], src[3 if wave else 4:]
      ptrs = ([dst.data_ptr() + lo * per * dst.element_size()] if wave else []) + [_p(t) for t in head]
      if wave == len(waves):
        waves.append([])
      waves[wave].append((ptrs, lo * per, rows * per))
      wave += 1
  for jobs in waves:
    for k in range(0, len(jobs), _ROWS_MAX_JOBS):
      part = jobs[k:k + _ROWS_MAX_JOBS]
      arr = (_RowsJob * len(part))()      # kept alive across the call
      moved = 0
      for j, (ptrs, off, numel) in enumerate(part):
        for q, ptr in enumerate(ptrs):
          arr[j].src[q] = ptr
        arr[j].dst_off, arr[j].numel = off, numel
        moved += (len(ptrs) + 1) * numel * dst.element_size()
      call('tg_rows_assemble', ctypes.addressof(arr), len(part), _p(dst), _dt(dst), _stream(),
           work=('rows_assemble:jobs%d:numel%d' % (len(part), sum(j[2] for j in part)), 0, moved))
  return dst


def _rows_out(d, spec):
  """One output of ``rows``: a view for a (lo, hi) range, a new tensor for a tuple of ranges."""
  if isinstance(spec[0], int):
    return d.narrow(0, spec[0], spec[1] - spec[0])
  out = torch.empty((sum(hi - lo for lo, hi in spec),) + tuple(d.shape[1:]), dtype=d.dtype, device=d.device)
  blocks, off = [], 0
  for lo, hi in spec:
    blocks.append((off, hi - lo, [d.narrow(0, lo, hi - lo)]))
    off += hi - lo
  return assemble_rows(out, blocks)


class RowsFn(torch.autograd.Function):
  """Row ranges of one batch x [N, ...], each output either a VIEW x[lo:hi] (spec (lo, hi)) or a new tensor made of several
  ranges one after the other (spec ((lo, hi), ...): a repeat, a reordering).

  The generator's batch [s_cyc; s'; t'; t_cyc] feeds the two discriminators ([s_cyc; s'] and [t'; t_cyc]), the re-encoding
  pass ([s'; t']) and the cycle losses (s_cyc, t_cyc); the encoder's batch [E(s); E(t)] feeds the content losses (each half)
  and, repeated, the four generator passes (twingan.py:198-288).  As framework chunks / cats that was a copy per consumer
  forward and, backward, a chain of framework adds, zero fills and the cat of the chunks' gradients.  Here the range
  consumers read the batch in place, the repeats are one launch, and the backward writes every row block of the gradient
  ONCE -- the sum of all the incoming gradients that cover it -- in one launch (assemble_rows)."""

  @staticmethod
  def forward(ctx, x, specs):
    assert x.is_contiguous()
    ctx.specs, ctx.shape = specs, tuple(x.shape)
    ctx.set_materialize_grads(False)      # an unused output contributes nothing: no zero tensor is made for it
    d = x.detach()      # the views are views of the storage, not autograd views of the input (nothing is modified in place)
    return tuple(_rows_out(d, spec) for spec in specs)

  @staticmethod
  def backward(ctx, *grads):
    n = ctx.shape[0]
    if torch.is_grad_enabled() and any(g is not None and g.requires_grad for g in grads):
      # a create_graph pass (the gradient penalty through the split of the two discriminators' features): differentiable when
      # the ranges tile the batch in order -- the gradient is then their concatenation (CatRowsFn)
      if all(isinstance(sp[0], int) for sp in ctx.specs) and all(g is not None for g in grads) and \
          [sp[0] for sp in ctx.specs] + [n] == [0] + [sp[1] for sp in ctx.specs]:
        return cat_rows([g.contiguous() for g in grads]), None
      raise NotImplementedError('second-order gradient through ops.rows with overlapping / repeated ranges')
    pieces = []      # (lo, hi, gradient rows)
    for spec, g in zip(ctx.specs, grads):
      if g is None:
        continue
      g = g.contiguous()
      if isinstance(spec[0], int):
        pieces.append((spec[0], spec[1], g))
      else:
        off = 0
        for lo, hi in spec:
          pieces.append((lo, hi, g.narrow(0, off, hi - lo)))
          off += hi - lo
    if not pieces:
      return None, None
    if len(pieces) == 1 and pieces[0][:2] == (0, n):
      return pieces[0][2], None
    cuts = sorted({0, n} | {lo for lo, _, _ in pieces} | {hi for _, hi, _ in pieces})
    gx = torch.empty(ctx.shape, dtype=pieces[0][2].dtype, device=pieces[0][2].device)
    assemble_rows(gx, [(a, b - a, [g.narrow(0, a - lo, b - a) for lo, hi, g in pieces if lo <= a and b <= hi])
                       for a, b in zip(cuts[:-1], cuts[1:])])
    return gx, None


def rows(x, specs):
  """-> a tuple, one tensor per spec: x[lo:hi] as a view for (lo, hi), the listed ranges one after the other as a new tensor
  for ((lo, hi), ...); see RowsFn."""
  specs = tuple(tuple(sp) if isinstance(sp[0], int) else tuple(tuple(r) for r in sp) for sp in specs)
  x = x.contiguous()
  if not (torch.is_grad_enabled() and x.requires_grad):
    return tuple(_rows_out(x, spec) for spec in specs)
  return RowsFn.apply(x, specs)


def row_views(x, ranges):
  """-> a tuple of views x[lo:hi]."""
  return rows(x, ranges)


class CatRowsFn(torch.autograd.Function):
  """tf.concat(tensors, 0): one launch forward; every input's gradient is a row range of the incoming one (a view)."""

  @staticmethod
  def forward(ctx, *tensors):
    ctx.rows = [t.shape[0] for t in tensors]
    return _cat_rows_copy(tensors)

  @staticmethod
  def backward(ctx, g):
    out, off = [], 0
    for k, r in enumerate(ctx.rows):
      out.append(g.narrow(0, off, r) if ctx.needs_input_grad[k] else None)
      off += r
    return tuple(out)


def _cat_rows_copy(tensors):
  t0 = tensors[0]
  out = torch.empty((sum(t.shape[0] for t in tensors),) + tuple(t0.shape[1:]), dtype=t0.dtype, device=t0.device)
  blocks, off = [], 0
  for t in tensors:
    blocks.append((off, t.shape[0], [t.detach()]))
    off += t.shape[0]
  return assemble_rows(out, blocks)


def cat_rows(tensors):
  """torch.cat(tensors, 0) -- a view of the rows when the tensors already sit one after the other in one allocation (the
  trainer's static input buffers under graph replay) and no tape is involved, else one tg_rows_assemble launch."""
  t0 = tensors[0]
  assert all(t.dtype == t0.dtype and t.shape[1:] == t0.shape[1:] for t in tensors), [(t.shape, t.dtype) for t in tensors]
  taped = torch.is_grad_enabled() and any(t.requires_grad for t in tensors)
  if not taped and all(t.is_contiguous() and t.untyped_storage().data_ptr() == t0.untyped_storage().data_ptr() for t in tensors):
    end = t0.data_ptr() + t0.numel() * t0.element_size()
    for t in tensors[1:]:
      if t.data_ptr() != end:
        break
      end += t.numel() * t.element_size()
    else:
      return torch.as_strided(t0, (sum(t.shape[0] for t in tensors),) + tuple(t0.shape[1:]), t0.stride())
  return CatRowsFn.apply(*tensors) if taped else _cat_rows_copy(tensors)


class CastFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x, dtype):
    ctx.src = x.dtype
    return cast_raw(x, dtype)

  @staticmethod
  def backward(ctx, g):
    return CastFn.apply(g.contiguous(), ctx.src), None


def cast(x, dtype):
  return x if x.dtype == dtype else CastFn.apply(x, dtype)


# ------------------------------------------------------------------------------------------------
# minibatch stddev (nets/pggan_utils.py:353-366)
# ------------------------------------------------------------------------------------------------
def _mbstd_eps(dtype):
  return 1e-8 if dtype == torch.float32 else 1e-6      # pggan_utils.py:359


class MbstdFn(torch.autograd.Function):
  """[n,h,w,c] -> [n,h,w,cpad]: x, the batch-stddev statistic in channel c, zeros above.  ``groups``
  consecutive sub-batches (discriminator calls batched along N) each keep their own statistic."""

  @staticmethod
  def forward(ctx, x, cpad, groups):
    _chk(x)
    n, h, w, c = x.shape
    out = torch.empty((n, h, w, cpad), dtype=x.dtype, device=x.device)
    call('tg_mbstd_fwd', _p(x), _p(out), None, n, groups, h * w, c, cpad, _mbstd_eps(x.dtype), _dt(x), _stream(),
         work=('mbstd_fwd' + _shape_tag(x), 0, _nb(x, out)))
    ctx.cpad, ctx.groups = cpad, groups
    ctx.set_materialize_grads(False)      # see Conv2dFn: the gradient penalty's second backward brings none
    ctx.save_for_backward(x)
    return out

  @staticmethod
  def backward(ctx, gout):
    if gout is None:
      return None, None, None
    x, = ctx.saved_tensors
    return MbstdBwdFn.apply(gout.contiguous(), x, ctx.cpad, ctx.groups), None, None


class MbstdBwdFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, gout, x, cpad, groups):
    n, h, w, c = x.shape
    gx = torch.empty_like(x)
    call('tg_mbstd_bwd', _p(gout), _p(x), _p(gx), n, groups, h * w, c, cpad, _mbstd_eps(x.dtype), _dt(x), _stream(),
         work=('mbstd_bwd' + _shape_tag(x), 0, _nb(gout, x, gx)))
    ctx.cpad, ctx.groups = cpad, groups
    ctx.save_for_backward(gout, x)
    return gx

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, v):
    gout, x = ctx.saved_tensors
    v = v.contiguous()
    n, h, w, c = x.shape
    ggout = torch.empty_like(gout) if ctx.needs_input_grad[0] else None
    gx2 = torch.empty_like(x) if ctx.needs_input_grad[1] else None
    call('tg_mbstd_bwd_bwd', _p(v), _p(gout), _p(x), _p(ggout), _p(gx2), n, ctx.groups, h * w, c, ctx.cpad,
         _mbstd_eps(x.dtype), _dt(x), _stream(), work=('mbstd_bwd_bwd' + _shape_tag(x), 0, _nb(v, gout, x, ggout, gx2)))
    return ggout, gx2, None, None


def minibatch_state_concat(x, cpad, groups=1):
  return MbstdFn.apply(x, cpad, int(groups))


# ------------------------------------------------------------------------------------------------
# SAGAN self-attention pieces (libs/self_attention.py:57-69): every op's backward is made of the same ops
# ------------------------------------------------------------------------------------------------
class BGemmFn(torch.autograd.Function):
  """c[i] = alpha * op(a[i]) @ op(b[i]) for 3-D contiguous a, b (tg_batched_gemm); both gradients are BGemmFn again."""

  @staticmethod
  def forward(ctx, a, b, ta, tb, alpha):
    _chk(a, b)
    assert a.dim() == 3 and b.dim() == 3 and a.shape[0] == b.shape[0] and a.dtype == b.dtype
    m = a.shape[2] if ta else a.shape[1]
    k = a.shape[1] if ta else a.shape[2]
    n = b.shape[1] if tb else b.shape[2]
    assert (b.shape[2] if tb else b.shape[1]) == k, (a.shape, b.shape, ta, tb)
    c = torch.empty((a.shape[0], m, n), dtype=a.dtype, device=a.device)
    call('tg_batched_gemm', _p(a), _p(b), _p(c), a.shape[0], m, n, k, int(ta), int(tb), a.shape[2], b.shape[2], n,
         a.shape[1] * a.shape[2], b.shape[1] * b.shape[2], m * n, alpha, 0, _dt(a), 0, _stream(),
         work=('bgemm:%dx%dx%d:b%d' % (m, n, k, a.shape[0]), 2 * a.shape[0] * m * n * k, _nb(a, b, c)))
    ctx.ta, ctx.tb, ctx.alpha = ta, tb, alpha
    ctx.save_for_backward(a, b)
    return c

  @staticmethod
  def backward(ctx, g):
    a, b = ctx.saved_tensors
    g = g.contiguous()
    ta, tb, al = ctx.ta, ctx.tb, ctx.alpha
    ga = gb = None
    if ctx.needs_input_grad[0]:
      ga = BGemmFn.apply(b, g, tb, True, al) if ta else BGemmFn.apply(g, b, False, not tb, al)
    if ctx.needs_input_grad[1]:
      gb = BGemmFn.apply(g, a, True, ta, al) if tb else BGemmFn.apply(a, g, not ta, False, al)
    return ga, gb, None, None, None


def bgemm(a, b, ta=False, tb=False, alpha=1.0):
  return BGemmFn.apply(a.contiguous(), b.contiguous(), bool(ta), bool(tb), float(alpha))


class SoftmaxRowsFn(torch.autograd.Function):
  """softmax over the last axis (tf.nn.softmax(s, axis=-1), libs/self_attention.py:63)."""

  @staticmethod
  def forward(ctx, s):
    _chk(s)
    p = torch.empty_like(s)
    cols = s.shape[-1]
    call('tg_softmax_rows_fwd', _p(s), _p(p), s.numel() // cols, cols, _dt(s), _stream(),
         work=('softmax_fwd:cols%d:rows%d' % (cols, s.numel() // cols), 0, _nb(s, p)))
    ctx.save_for_backward(p)
    return p

  @staticmethod
  def backward(ctx, dp):
    p, = ctx.saved_tensors
    return SoftmaxRowsBwdFn.apply(p, dp.contiguous())


class SoftmaxRowsBwdFn(torch.autograd.Function):
  """ds = p * (dp - sum(dp * p)); its gradient in dp is the same map of v, in p the tg_softmax_rows_bwd_bwd kernel."""

  @staticmethod
  def forward(ctx, p, dp):
    _chk(p, dp)
    ds = torch.empty_like(p)
    cols = p.shape[-1]
    call('tg_softmax_rows_bwd', _p(p), _p(dp), _p(ds), p.numel() // cols, cols, _dt(p), _stream(),
         work=('softmax_bwd:cols%d:rows%d' % (cols, p.numel() // cols), 0, _nb(p, dp, ds)))
    ctx.save_for_backward(p, dp)
    return ds

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, v):
    p, dp = ctx.saved_tensors
    v = v.contiguous()
    cols = p.shape[-1]
    gp = None
    if ctx.needs_input_grad[0]:
      gp = torch.empty_like(p)
      call('tg_softmax_rows_bwd_bwd', _p(p), _p(dp), _p(v), _p(gp), p.numel() // cols, cols, _dt(p), _stream())
    gdp = None
    if ctx.needs_input_grad[1]:
      gdp = torch.empty_like(p)
      call('tg_softmax_rows_bwd', _p(p), _p(v), _p(gdp), p.numel() // cols, cols, _dt(p), _stream())
    return gp, gdp


def transpose16(x):
  """[n, rows, cols] -> [n, cols, rows] contiguous, 16-bit tensors (tg_transpose16)."""
  _chk(x)
  n, r, c = x.shape
  out = torch.empty((n, c, r), dtype=x.dtype, device=x.device)
  call('tg_transpose16', _p(x), _p(out), n, r, c, _stream(), work=('transpose16:numel%d' % x.numel(), 0, 2 * _nb(x)))
  return out


def flash_attention_supported(q, v):
  return q.dtype in HALF_TYPES and bool(_lib.load().tg_flash_attention_supported(q.shape[1], q.shape[2], v.shape[2]))


def _flash_workspace(q, v, which):
  """which: 0 forward, 1 first-order backward, 2 second-order backward."""
  n, ln, dk = q.shape
  nbytes = int(_lib.load().tg_flash_attention_workspace_bytes(n, ln, dk, v.shape[2], int(which)))
  assert nbytes > 0, 'flash attention: unsupported shape %s / %s' % (tuple(q.shape), tuple(v.shape))
  return torch.empty(nbytes, dtype=torch.uint8, device=q.device)


def flash_attention_fwd_raw(q, k, v):
  """(o, lse): o = softmax(q k^T) v per image without the [len, len] map."""
  _chk(q, k, v)
  n, ln, dk = q.shape
  dv = v.shape[2]
  ws = _flash_workspace(q, v, 0)
  o = torch.empty((n, ln, dv), dtype=q.dtype, device=q.device)
  lse = torch.empty((n, ln), dtype=torch.float32, device=q.device)
  call('tg_flash_attention_fwd', _p(q), _p(k), _p(v), _p(o), _p(lse), _p(ws), n, ln, dk, dv, _dt(q), _stream(),
       work=('flash_fwd:len%d:dk%d:dv%d:n%d' % (ln, dk, dv, n), 2 * n * ln * ln * (dk + dv), _nb(q, k, v, o)))
  return o, lse


class _SecondOrder(object):
  depth = 0


class second_order(object):
  """Marks a forward pass whose backward is itself differentiated (the gradient-penalty pass of the discriminator,
  image_generation.py:414-439): a layer whose fused kernel has no second-order backward builds its differentiable
  composition there (flash attention with USE_FLASH_BWD_BWD = False (tests))."""

  def __enter__(self):
    _SecondOrder.depth += 1

  def __exit__(self, *exc):
    _SecondOrder.depth -= 1


def in_second_order():
  return _SecondOrder.depth > 0


def flash_attention_trainable(q, v):
  """Forward + first-order backward (+ second-order backward, USE_FLASH_BWD_BWD) exist as flash kernels for this shape."""
  return flash_attention_supported(q, v) and v.shape[2] <= 128 and (USE_FLASH_BWD_BWD or not in_second_order())


def _flash_bwd_raw(q, k, v, o, lse, go):
  n, ln, dk = q.shape
  dv = v.shape[2]
  gq, gk, gv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
  ws = _flash_workspace(q, v, 1)
  call('tg_flash_attention_bwd', _p(q), _p(k), _p(v), _p(go), _p(o), _p(lse), _p(ws), _p(gq), _p(gk), _p(gv), n, ln, dk,
       dv, _dt(q), _stream(),
       work=('flash_bwd:len%d:dk%d:dv%d:n%d' % (ln, dk, dv, n), 2 * n * ln * ln * (3 * dk + 3 * dv), _nb(q, k, v, o, go, gq, gk, gv)))
  return gq, gk, gv


class FlashAttnBwdFn(torch.autograd.Function):
  """The first-order backward of FlashAttnFn as a differentiable node (create_graph passes: the gradient penalty,
  image_generation.py:414-439): (q, k, v, dO) -> (dq, dk, dv) with o / lse as saved constants of the forward; its own
  backward is tg_flash_attention_bwd_bwd, which accounts for o / lse through q, k, v."""

  @staticmethod
  def forward(ctx, q, k, v, o, lse, go):
    ctx.save_for_backward(q, k, v, o, lse, go)
    return _flash_bwd_raw(q, k, v, o, lse, go)

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, aq, ak, av):
    q, k, v, o, lse, go = ctx.saved_tensors
    n, ln, dk = q.shape
    dv = v.shape[2]
    aq = torch.zeros_like(q) if aq is None else aq.contiguous()
    ak = torch.zeros_like(k) if ak is None else ak.contiguous()
    av = torch.zeros_like(v) if av is None else av.contiguous()
    _chk(aq, ak, av)
    adj_q, adj_k, adj_v, adj_go = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), torch.empty_like(go)
    ws = _flash_workspace(q, v, 2)
    call('tg_flash_attention_bwd_bwd', _p(q), _p(k), _p(v), _p(go), _p(o), _p(lse), _p(aq), _p(ak), _p(av), _p(ws),
         _p(adj_q), _p(adj_k), _p(adj_v), _p(adj_go), n, ln, dk, dv, _dt(q), _stream(),
         work=('flash_bwd_bwd:len%d:dk%d:dv%d:n%d' % (ln, dk, dv, n), 2 * n * ln * ln * (8 * dk + 10 * dv),
               _nb(q, k, v, o, go, aq, ak, av, adj_q, adj_k, adj_v, adj_go)))
    return adj_q, adj_k, adj_v, None, None, adj_go


class FlashAttnFn(torch.autograd.Function):
  """softmax(q k^T) v of libs/self_attention.py:56-63 without the [len, len] map in HBM (csrc/flash.hip): forward saves the
  per-query log-sum-exp; the backward recomputes the probabilities tile by tile.  Under create_graph the backward is the
  differentiable FlashAttnBwdFn (or, with USE_FLASH_BWD_BWD = False (tests), the batched-GEMM / softmax composition, which materialises
  the map)."""

  @staticmethod
  def forward(ctx, q, k, v):
    o, lse = flash_attention_fwd_raw(q, k, v)
    ctx.save_for_backward(q, k, v, o, lse)
    return o

  @staticmethod
  def backward(ctx, go):
    q, k, v, o, lse = ctx.saved_tensors
    go = go.contiguous()
    if torch.is_grad_enabled():
      if USE_FLASH_BWD_BWD:
        return FlashAttnBwdFn.apply(q, k, v, o.detach(), lse, go)
      p = softmax_rows(bgemm(q, k, False, True))
      gv = bgemm(p, go, True, False)
      gs = SoftmaxRowsBwdFn.apply(p, bgemm(go, v, False, True))
      return bgemm(gs, k, False, False), bgemm(gs, q, True, False), gv
    _chk(go)
    return _flash_bwd_raw(q, k, v, o, lse, go)


def flash_attention(q, k, v):
  return FlashAttnFn.apply(q.contiguous(), k.contiguous(), v.contiguous())


def softmax_rows(s):
  return SoftmaxRowsFn.apply(s.contiguous())


class TanhFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x):
    _chk(x)
    y = torch.empty_like(x)
    call('tg_tanh_fwd', _p(x), _p(y), x.numel(), _dt(x), _stream())
    ctx.save_for_backward(y)
    return y

  @staticmethod
  def backward(ctx, g):
    y, = ctx.saved_tensors
    return TanhBwdFn.apply(g.contiguous(), y)


class TanhBwdFn(torch.autograd.Function):
  """gx = g * (1 - y^2); d/dg = the same map of v, d/dy = -2 y g v (tg_mul3)."""

  @staticmethod
  def forward(ctx, g, y):
    _chk(g, y)
    gx = torch.empty_like(g)
    call('tg_tanh_bwd', _p(g), _p(y), _p(gx), g.numel(), _dt(g), _stream())
    ctx.save_for_backward(g, y)
    return gx

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, v):
    g, y = ctx.saved_tensors
    v = v.contiguous()
    gg = gy = None
    if ctx.needs_input_grad[0]:
      gg = torch.empty_like(g)
      call('tg_tanh_bwd', _p(v), _p(y), _p(gg), g.numel(), _dt(g), _stream())
    if ctx.needs_input_grad[1]:
      gy = torch.empty_like(g)
      call('tg_mul3', _p(y), _p(g), _p(v), _p(gy), -2.0, g.numel(), _dt(g), _stream())
    return gg, gy


def tanh(x):
  return TanhFn.apply(x.contiguous())


class ScaleDevFn(torch.autograd.Function):
  """x * s for a device fp32 scalar s [1] (sa_gamma): d/dx = g * s, d/ds = <g, x> (DotFn) -- bilinear, closed."""

  @staticmethod
  def forward(ctx, x, s):
    _chk(x, s)
    assert s.dtype == torch.float32 and s.numel() == 1
    out = torch.empty_like(x)
    call('tg_scale_dev', _p(x), _p(s), _p(out), x.numel(), _dt(x), _stream())
    ctx.save_for_backward(x, s)
    return out

  @staticmethod
  def backward(ctx, g):
    x, s = ctx.saved_tensors
    g = g.contiguous()
    gx = ScaleDevFn.apply(g, s) if ctx.needs_input_grad[0] else None
    gs = DotFn.apply(g, x).reshape(s.shape) if ctx.needs_input_grad[1] else None
    return gx, gs


class DotFn(torch.autograd.Function):
  """<a, b> over all elements -> fp32 [1] (two-stage sum in a fixed order)."""

  @staticmethod
  def forward(ctx, a, b):
    _chk(a, b)
    assert a.shape == b.shape and a.dtype == b.dtype
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    ws = torch.empty(1024, dtype=torch.float32, device=a.device)
    call('tg_dot', _p(a), _p(b), _p(out), _p(ws), a.numel(), _dt(a), _stream())
    ctx.save_for_backward(a, b)
    return out

  @staticmethod
  def backward(ctx, v):
    a, b = ctx.saved_tensors
    v = v.contiguous().float()
    ga = ScaleDevFn.apply(b, v) if ctx.needs_input_grad[0] else None
    gb = ScaleDevFn.apply(a, v) if ctx.needs_input_grad[1] else None
    return ga, gb


def scale_dev(x, s):
  return ScaleDevFn.apply(x.contiguous(), s)


# ------------------------------------------------------------------------------------------------
# spectral normalisation of a conv kernel (libs/sn.py:38-101)
# ------------------------------------------------------------------------------------------------
class SpectralNormFn(torch.autograd.Function):
  """(w_bar, u_new) = one power iteration on the fp32 master kernel w [kh,kw,cin,cout] from the persistent vector u
  [1,cout]: w_bar = w / sigma with sigma = v W u_new^T (tg_spectral_norm_fwd).  The backward (tg_spectral_norm_bwd) lets
  the gradient flow through sigma, v and u_new like the reference's graph; it is first order -- the gradient-penalty
  double backward reaches the master weight through w_bar, i.e. through ONE application of this node's backward."""

  out_buffer = None      # side channel of spectral_norm(): the persistent fp32 buffer this forward writes w_bar into

  @staticmethod
  def forward(ctx, w, u):
    _chk(w, u)
    cout = w.shape[-1]
    k_rows = w.numel() // cout
    buf, SpectralNormFn.out_buffer = SpectralNormFn.out_buffer, None
    # a fresh tensor object over the persistent storage: the buffer is rewritten by every run, the alias carries this
    # run's grad_fn
    w_bar = torch.empty_like(w) if buf is None else buf.view(w.shape)
    u_new = torch.empty_like(u)
    v = torch.empty(k_rows, dtype=torch.float32, device=w.device)
    stats = torch.empty(2, dtype=torch.float32, device=w.device)
    nbytes = _lib.load().tg_spectral_norm_workspace(k_rows, cout)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    call('tg_spectral_norm_fwd', _p(w), _p(u), _p(w_bar), _p(u_new), _p(v), _p(stats), k_rows, cout, _p(ws), nbytes,
         _stream(), work=('sn_fwd:k%d:c%d' % (k_rows, cout), 6 * w.numel(), 4 * _nb(w)))
    ctx.dims = (k_rows, cout, nbytes)
    ctx.save_for_backward(w, u, u_new, v, stats)
    ctx.mark_non_differentiable(u_new)
    return w_bar, u_new

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, g, _gu):
    w, u, u_new, v, stats = ctx.saved_tensors
    k_rows, cout, nbytes = ctx.dims
    g = g.contiguous()
    sink = None if _State.skip_param_grads else GradSink.get(w)
    gw = sink if sink is not None else torch.empty_like(w)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    call('tg_spectral_norm_bwd', _p(g), _p(w), _p(u), _p(u_new), _p(v), _p(stats), _p(gw), 1 if sink is not None else 0,
         k_rows, cout, _p(ws), nbytes, _stream(), work=('sn_bwd:k%d:c%d' % (k_rows, cout), 6 * w.numel(), 4 * _nb(w)))
    return (None if sink is not None else gw), None


class SpectralNormPreFn(torch.autograd.Function):
  """The node of SpectralNormFn around a power iteration that has ALREADY run (spectral_norm_multi: every kernel of a run
  in three launches): forward hands out the precomputed w_bar / u_new, backward is SpectralNormFn's."""

  pre = None      # side channel of spectral_norm_multi(): (w_bar buffer, u_new, v, stats, table) of the node being made.
                  # NOT inputs: an output that is a view of an INPUT rebases that input's history onto this node, and the
                  # persistent buffers would then drag the previous run's graph into the next one

  @staticmethod
  def forward(ctx, w, u):
    (w_bar, u_new, v, stats, table), SpectralNormPreFn.pre = SpectralNormPreFn.pre, None
    cout = w.shape[-1]
    k_rows = w.numel() // cout
    ctx.dims = (k_rows, cout, _lib.load().tg_spectral_norm_workspace(k_rows, cout))
    # u_new / v / stats are the table's PERSISTENT buffers (fixed addresses for graph replay), rewritten by raw kernel
    # writes that autograd's version counters do not see: the backward refuses to run on a later run's values
    ctx.table, ctx.generation = table, table.generation
    u1 = u_new.view_as(u)
    ctx.save_for_backward(w, u, u1, v, stats)
    ctx.mark_non_differentiable(u1)
    return w_bar.view(w.shape), u1

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, g, _gu):
    if ctx.table.generation != ctx.generation:
      raise RuntimeError('spectral_norm_multi: the power iteration ran again (generation %d -> %d) before the backward of the '
                         'graph built on its previous results; one outstanding graph per SnTable'
                         % (ctx.generation, ctx.table.generation))
    return SpectralNormFn.backward(ctx, g, _gu)



class SnTable:
  """Persistent per-kernel buffers (u_new, v, stats, workspace) and the device job table of spectral_norm_multi for one
  fixed list of (w, u, w_bar buffer) triples -- fixed addresses, so a captured hipGraph replays the three launches."""

  def __init__(self, items):
    lib = _lib.load()
    dev = items[0][0].device
    self.n = len(items)
    self.generation = 0      # bumped by every run(): SpectralNormPreFn.backward checks it
    self.key = tuple((w.data_ptr(), u.data_ptr(), out.data_ptr()) for w, u, out in items)
    host = ctypes.create_string_buffer(lib.tg_sn_table_bytes(self.n))
    totals = (ctypes.c_int32 * 3)(0, 0, 0)
    self.bufs = []
    for j, (w, u, out) in enumerate(items):
      _chk(w, u, out)
      cout = w.shape[-1]
      k_rows = w.numel() // cout
      nbytes = lib.tg_spectral_norm_workspace(k_rows, cout)
      u_new = torch.empty(cout, dtype=torch.float32, device=dev)
      v = torch.empty(k_rows, dtype=torch.float32, device=dev)
      stats = torch.empty(2, dtype=torch.float32, device=dev)
      ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
      call('tg_sn_table_fill', j, _p(w), _p(u), _p(out), _p(u_new), _p(v), _p(stats), _p(ws), nbytes, k_rows, cout,
           ctypes.addressof(host), totals)
      self.bufs.append((u_new, v, stats, ws))
    self.totals = tuple(int(t) for t in totals)
    self.table = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(dev)
    self.nbytes = sum(4 * _nb(w) for w, _, _ in items)

  def assign_u(self):
    """u <- u' for every kernel of the table (pggan.end_run), one launch."""
    call('tg_sn_assign_u', _p(self.table), self.n, _stream())

  def run(self):
    self.generation += 1
    call('tg_spectral_norm_fwd_multi', _p(self.table), self.n, self.totals[0], self.totals[1], self.totals[2], _stream(),
         work=('sn_fwd_multi:%d' % self.n, 0, self.nbytes))


def spectral_norm_multi(items, table=None):
  """items: [(w, u [1, cout] contiguous, out)] with persistent fp32 ``out`` buffers (as spectral_norm's) -> ([(w_bar, u_new)],
  table).  ``table``: the SnTable of a previous call with the same tensors (rebuilt when an address changed)."""
  items = [(w, u.contiguous(), out) for w, u, out in items]
  key = tuple((w.data_ptr(), u.data_ptr(), out.data_ptr()) for w, u, out in items)
  if table is None or table.key != key:
    table = SnTable(items)
  table.run()
  outs = []
  for (w, u, out), (u_new, v, stats, _) in zip(items, table.bufs):
    SpectralNormPreFn.pre = (out, u_new, v, stats, table)
    try:
      outs.append(SpectralNormPreFn.apply(w, u))
    finally:
      SpectralNormPreFn.pre = None
  return outs, table


def spectral_norm(w, u, out=None):
  """-> (w_bar, u_new); see SpectralNormFn.  ``out``: a persistent fp32 buffer of w's size that receives w_bar (its
  MFMA packs then live in PackCache like a master weight's and are rebuilt by PackCache.refresh, one launch for all
  normalised kernels of a run, instead of once per use)."""
  assert out is None or (out.dtype == torch.float32 and out.numel() == w.numel() and out.is_contiguous())
  SpectralNormFn.out_buffer = out
  try:
    return SpectralNormFn.apply(w, u.contiguous())
  finally:
    SpectralNormFn.out_buffer = None


# ------------------------------------------------------------------------------------------------
# small dense layer (layers.fully_connected, nets/pggan_utils.py:323-327)
# ------------------------------------------------------------------------------------------------
def _is_parameter(t):
  """Is ``t`` a trainable variable (ParamStore registers every one, and the per-run spectrally normalised kernels, with a
  gradient sink)?  Not "is it a leaf": the gradient penalty's interpolates and the leaves of a segmented backward (Cuts.cut)
  are leaves too, and their gradient is exactly what no_param_grads passes are run for."""
  ent = GradSink._sinks.get(GradSink._key(t))
  return ent is not None and ent[0]() is t


class GemmFn(torch.autograd.Function):
  """c = op(a) @ op(b), fp32 row-major; fully differentiable (every gradient is another GemmFn)."""

  @staticmethod
  def forward(ctx, a, b, ta, tb):
    _chk(a, b)
    m = a.shape[1] if ta else a.shape[0]
    k = a.shape[0] if ta else a.shape[1]
    n = b.shape[0] if tb else b.shape[1]
    assert (b.shape[1] if tb else b.shape[0]) == k
    c = torch.empty((m, n), dtype=torch.float32, device=a.device)
    call('tg_small_gemm', _p(a), _p(b), None, _p(c), m, n, k, int(ta), int(tb), 0, _stream())
    ctx.ta, ctx.tb = ta, tb
    ctx.save_for_backward(a, b)
    return c

  @staticmethod
  def backward(ctx, g):
    a, b = ctx.saved_tensors
    g = g.contiguous()
    ta, tb = ctx.ta, ctx.tb
    ga = gb = None
    # ops.no_param_grads (the gradient penalty's inner gradient): a leaf operand is a parameter -- its gradient arrives
    # through the double backward, this pass would compute it only to drop it
    skip = _State.skip_param_grads
    if ctx.needs_input_grad[0] and not (skip and _is_parameter(a)):
      ga = GemmFn.apply(b, g, tb, True) if ta else GemmFn.apply(g, b, False, not tb)
    if ctx.needs_input_grad[1] and not (skip and _is_parameter(b)):
      gb = GemmFn.apply(g, a, True, ta) if tb else GemmFn.apply(a, g, not ta, False)
    return ga, gb, None, None


class AddRowBiasFn(torch.autograd.Function):
  """x[m,n] + b[n] for fp32 2-D tensors (tiny: the [B,1] prediction)."""

  @staticmethod
  def forward(ctx, x, b):
    _chk(x, b)
    out = torch.empty_like(x)
    call('tg_bias_lrelu_fwd', _p(x), _p(b), _p(out), x.shape[0], x.shape[1], 1.0, TG_F32, _stream())
    return out

  @staticmethod
  def backward(ctx, g):
    g = g.contiguous()
    return g, (ChannelSumFn.apply(g) if (ctx.needs_input_grad[1] and not _State.skip_param_grads) else None)


class FcFn(torch.autograd.Function):
  """x [B,K] (the activations' type) @ w [K,N] + b -> fp32 [B,N] as ONE launch (tg_fc_fwd; the composition below is a cast,
  a GEMM and a bias add) and, backward, ONE launch for gx / gw / gb (tg_fc_bwd; four there), the parameter gradients
  straight into their sinks.  First-order passes only: the gradient penalty differentiates the composition."""

  @staticmethod
  def forward(ctx, x, w, b):
    _chk(x, w, b)
    m, k = x.shape
    n = w.shape[1]
    y = torch.empty((m, n), dtype=torch.float32, device=x.device)
    call('tg_fc_fwd', _p(x), _p(w), _p(b), _p(y), m, n, k, _dt(x), _stream())
    ctx.save_for_backward(x, w, b)
    return y

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, g):
    x, w, b = ctx.saved_tensors
    m, k = x.shape
    n = w.shape[1]
    g = g.contiguous()
    params = not _State.skip_param_grads
    gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
    sw = GradSink.get(w) if (ctx.needs_input_grad[1] and params) else None
    gw = sw if sw is not None else (torch.empty_like(w) if (ctx.needs_input_grad[1] and params) else None)
    sb = GradSink.get(b) if (b is not None and ctx.needs_input_grad[2] and params) else None
    gb = sb if sb is not None else (torch.empty_like(b) if (b is not None and ctx.needs_input_grad[2] and params) else None)
    if gx is not None or gw is not None or gb is not None:
      call('tg_fc_bwd', _p(x), _p(w), _p(g), _p(gx), _p(gw), _p(gb), m, n, k, 1 if sw is not None else 0,
           1 if sb is not None else 0, _dt(x), _stream())
    return gx, (None if sw is not None else gw), (None if sb is not None else gb)


def fully_connected(x, w, b):
  """x [B,K] (any dtype) @ w [K,N] + b -> fp32 [B,N]."""
  if (not in_second_order() and not deterministic() and x.dim() == 2 and x.is_contiguous()
      and w.dtype == torch.float32 and w.is_contiguous() and w.shape[1] <= w.shape[0]):
    return FcFn.apply(x, w, b)
  y = GemmFn.apply(cast(x, torch.float32), w, False, False)
  return AddRowBiasFn.apply(y, b) if b is not None else y


def latent_conv(noise, w):
  """The plain PGGAN generator's first layer (nets/pggan.py:135-153): a k x k VALID conv over the [B, 1, 1, C] latent noise
  zero-padded to (2k - 1) x (2k - 1), whose only non-zero pixel meets tap (k-1-oy, k-1-ox) at output pixel (oy, ox):
      y[b, oy, ox, :] = noise[b, :] @ w[k-1-oy, k-1-ox]        i.e.  [B, C] @ [C, k*k*C'] of the flipped kernel,
  a GEMM with 1 / (2k-1)^2 of the padded conv's multiplies, every one of them on a non-zero (the conv kernels have no
  k = 4 VALID form off the dense k x k-input case: the layer fell to the one-thread-per-output kernel, 62 % of BASELINE
  configs[0]'s step).  fp32 accumulation over the fp32 master kernel; the flip / regrouping of the kernel and its adjoint
  (into the kernel's gradient) are framework view ops of a 4 MB tensor.  -> [B, k, k, C'] of noise's dtype."""
  b = noise.shape[0]
  k, _, c, co = w.shape
  wm = w.flip(0, 1).permute(2, 0, 1, 3).reshape(c, k * k * co)
  y = GemmFn.apply(cast(noise.reshape(b, c), torch.float32), wm.contiguous(), False, False)
  return cast(y, noise.dtype).view(b, k, k, co)


# ------------------------------------------------------------------------------------------------
# losses (twingan.py:464,502; image_generation.py:333,350,431-436)
# ------------------------------------------------------------------------------------------------
def _scalar_sum_into(x, y, out, scale):
  """out[0] = scale * sum(x) (y None) or scale * sum|x - y|; the ordered two-stage form in deterministic mode."""
  tag = ('abs_diff_sum' if y is not None else 'sum') + ':numel%d' % x.numel()
  if deterministic():
    ws = torch.empty(_ORDERED_ROWS, dtype=torch.float32, device=x.device)
    call('tg_sum_ordered', _p(x), _p(y), _p(out), x.numel(), scale, 0, _p(ws), ws.numel(), _dt(x), _stream(),
         work=(tag, 0, _nb(x, y)))
  elif y is not None:
    call('tg_abs_diff_sum', _p(x), _p(y), _p(out), x.numel(), scale, 0, _dt(x), _stream(), work=(tag, 0, _nb(x, y)))
  else:
    call('tg_sum', _p(x), _p(out), x.numel(), scale, 0, _dt(x), _stream(), work=(tag, 0, _nb(x)))


class MeanFn(torch.autograd.Function):
  """mean(x) * weight -> fp32 [1]."""

  @staticmethod
  def forward(ctx, x, weight):
    _chk(x)
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    _scalar_sum_into(x, None, out, weight / x.numel())
    ctx.meta = (x.shape, x.dtype, weight / x.numel())
    return out

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, g):
    shape, dtype, k = ctx.meta
    return fill(shape, k, dtype, g.device, scalar=g.contiguous()), None


class AbsDiffMeanFn(torch.autograd.Function):
  """weight * mean|a - b| (tf.losses.absolute_difference) -> fp32 [1]."""

  @staticmethod
  def forward(ctx, a, b, weight):
    _chk(a, b)
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    _scalar_sum_into(a, b, out, weight / a.numel())
    ctx.k = weight / a.numel()
    ctx.save_for_backward(a, b)
    return out

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, g):
    a, b = ctx.saved_tensors
    ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
    gb = torch.empty_like(b) if ctx.needs_input_grad[1] else None
    call('tg_abs_diff_bwd', _p(a), _p(b), _p(g.contiguous()), _p(ga), _p(gb), a.numel(), ctx.k, _dt(a), _stream(),
         work=('abs_diff_bwd:numel%d' % a.numel(), 0, _nb(a, b, ga, gb)))
    return ga, gb, None


class GradPenaltyFn(torch.autograd.Function):
  """lambda * mean_b (||g_b||_2 - 1)^2 (image_generation.py:431-436) -> fp32 [1]."""

  @staticmethod
  def forward(ctx, g, lam):
    _chk(g)
    b = g.shape[0]
    ss = torch.empty(b, dtype=torch.float32, device=g.device)
    call('tg_sample_sumsq', _p(g), _p(ss), b, g.numel() // b, _dt(g), _stream(),
         work=('sample_sumsq:numel%d' % g.numel(), 0, _nb(g)))
    loss = torch.empty(1, dtype=torch.float32, device=g.device)
    coef = torch.empty(b, dtype=torch.float32, device=g.device)
    call('tg_gp_penalty', _p(ss), _p(loss), _p(coef), b, lam, _stream())
    ctx.save_for_backward(g, coef)
    return loss

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, gl):
    g, coef = ctx.saved_tensors
    out = torch.empty_like(g)
    b = g.shape[0]
    call('tg_sample_scale', _p(g), _p(coef), _p(gl.contiguous()), _p(out), b, g.numel() // b, _dt(g), _stream(),
         work=('sample_scale:numel%d' % g.numel(), 0, _nb(g, out)))
    return out, None


class PredLossFn(torch.autograd.Function):
  """weight * mean_i f(x_i) over the fp32 discriminator predictions; f by ``mode`` (0 identity, 1 relu(a + b x),
  2 sigmoid cross entropy against label a, 3 square) -- image_generation.py:331-400.  First order."""

  @staticmethod
  def forward(ctx, x, mode, a, b, weight):
    _chk(x)
    assert x.dtype == torch.float32
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    k = weight / x.numel()
    call('tg_pred_loss_fwd', _p(x), _p(out), x.numel(), mode, a, b, k, 0, _stream())
    ctx.meta = (mode, a, b, k)
    ctx.save_for_backward(x)
    return out

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, g):
    x, = ctx.saved_tensors
    mode, a, b, k = ctx.meta
    gx = torch.empty_like(x)
    call('tg_pred_loss_bwd', _p(x), _p(g.contiguous()), _p(gx), x.numel(), mode, a, b, k, _stream())
    return gx, None, None, None, None


class _PredJob(ctypes.Structure):
  _fields_ = [('group', ctypes.c_int32), ('term', ctypes.c_int32), ('mode', ctypes.c_int32), ('a', ctypes.c_float),
              ('b', ctypes.c_float), ('coef', ctypes.c_float)]


def _pred_jobs(jobs):
  arr = (_PredJob * len(jobs))()
  for k, (group, term, mode, a, b, coef) in enumerate(jobs):
    arr[k] = _PredJob(group, term, mode, a, b, coef)
  return arr


class PredLossesFn(torch.autograd.Function):
  """Every prediction loss of ONE batched discriminator call: pred fp32 [groups * group_size, 1] (e.g. [real; cycle; prime]);
  ``jobs``: tuples (group, term, mode, a, b, coef) -- term += coef * mean_i f_mode(pred[group]_i; a, b), f as PredLossFn's --
  -> ``nterms`` fp32 [1] tensors.  One launch forward (all means, all terms) and one backward (the gradient of the whole
  prediction) instead of a sum / fill launch per term and group plus the framework's sub / neg / add / cat glue
  (image_generation.py:331-400).  First order."""

  @staticmethod
  def forward(ctx, pred, group_size, jobs, nterms):
    _chk(pred)
    assert pred.dtype == torch.float32 and pred.numel() % group_size == 0
    groups = pred.numel() // group_size
    out = torch.empty(nterms, dtype=torch.float32, device=pred.device)
    arr = _pred_jobs(jobs)      # kept alive across the call
    call('tg_pred_losses_fwd', _p(pred), group_size, groups, ctypes.addressof(arr), len(jobs), _p(out), nterms, _stream())
    ctx.meta = (group_size, groups, jobs, nterms)
    ctx.save_for_backward(pred)
    return tuple(out[t:t + 1] for t in range(nterms))

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, *gterms):
    pred, = ctx.saved_tensors
    group_size, groups, jobs, nterms = ctx.meta
    keep = [g.contiguous() if g is not None else None for g in gterms]
    ptrs = (ctypes.c_void_p * nterms)(*[(_p(g) if g is not None else None) for g in keep])
    gpred = torch.empty_like(pred)
    arr = _pred_jobs(jobs)
    call('tg_pred_losses_bwd', _p(pred), group_size, groups, ctypes.addressof(arr), len(jobs), ctypes.addressof(ptrs), nterms,
         _p(gpred), _stream())
    return gpred, None, None, None


def pred_losses(pred, group_size, jobs, nterms):
  return PredLossesFn.apply(pred.contiguous(), int(group_size), tuple(jobs), int(nterms))


class SumScalarsFn(torch.autograd.Function):
  """tf.add_n over fp32 [1] loss terms (model/model_inheritor.py): one launch; every term's gradient IS the incoming one."""

  @staticmethod
  def forward(ctx, *terms):
    _chk(*terms)
    assert all(t.dtype == torch.float32 and t.numel() == 1 for t in terms) and len(terms) <= 24
    out = torch.empty(1, dtype=torch.float32, device=terms[0].device)
    keep = [t.contiguous() for t in terms]
    ptrs = (ctypes.c_void_p * len(keep))(*[_p(t) for t in keep])
    call('tg_sum_scalars', ctypes.addressof(ptrs), len(keep), _p(out), _stream())
    ctx.n = len(terms)
    return out

  @staticmethod
  def backward(ctx, g):
    return (g,) * ctx.n


def sum_scalars(terms):
  terms = list(terms)
  return terms[0] if len(terms) == 1 else SumScalarsFn.apply(*terms)


class CosineDistanceFn(torch.autograd.Function):
  """weight * mean_b(1 - l2n(expected_b) . l2n(embedding_b)) -> fp32 [1] (twingan.py:507-521: the encoder-distillation
  loss; `expected` is the dataset's embedding, no gradient).  First order."""

  @staticmethod
  def forward(ctx, expected, embedding, weight):
    _chk(expected, embedding)
    assert expected.dtype == embedding.dtype == torch.float32 and expected.shape == embedding.shape
    b, d = embedding.shape
    out = torch.empty(1, dtype=torch.float32, device=embedding.device)
    call('tg_cosine_distance_fwd', _p(expected), _p(embedding), _p(out), b, d, weight, _stream())
    ctx.weight = weight
    ctx.save_for_backward(expected, embedding)
    return out

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, g):
    expected, embedding = ctx.saved_tensors
    b, d = embedding.shape
    ge = torch.empty_like(embedding)
    call('tg_cosine_distance_bwd', _p(expected), _p(embedding), _p(g.contiguous()), _p(ge), b, d, ctx.weight, _stream())
    return None, ge, None


def cosine_distance(expected, embedding, weight=1.0):
  return CosineDistanceFn.apply(expected.contiguous(), embedding.contiguous(), float(weight))


def hinge_mean(x, a, b, weight=1.0):
  """weight * mean(relu(a + b*x))."""
  return PredLossFn.apply(x.contiguous(), 1, float(a), float(b), float(weight))


def sigmoid_xent_mean(x, label, weight=1.0):
  """tf.losses.sigmoid_cross_entropy(label * ones, x) * weight."""
  return PredLossFn.apply(x.contiguous(), 2, float(label), 0.0, float(weight))


def square_mean(x, weight=1.0):
  return PredLossFn.apply(x.contiguous(), 3, 0.0, 0.0, float(weight))


def batch_variance(x):
  """Variance over every element of the minibatch as a device fp32 [1] (no autograd): DRAGAN's noise scale."""
  _chk(x)
  b = x.shape[0]
  s1 = torch.empty(1, dtype=torch.float32, device=x.device)
  ss = torch.empty(b, dtype=torch.float32, device=x.device)
  out = torch.empty(1, dtype=torch.float32, device=x.device)
  call('tg_sum', _p(x), _p(s1), x.numel(), 1.0, 0, _dt(x), _stream())
  call('tg_sample_sumsq', _p(x), _p(ss), b, x.numel() // b, _dt(x), _stream())
  call('tg_var_from_sums', _p(s1), _p(ss), _p(out), b, x.numel(), _stream())
  return out


def dragan_interpolates(real, noise, alpha):
  """real + alpha[b] * (perturbed - real) with perturbed = real + 0.5 * Var(real) * noise
  (image_generation.py:441-449,455-460).  No autograd (the interpolates are a leaf)."""
  _chk(real, noise, alpha)
  var = batch_variance(real)
  b = real.shape[0]
  delta = torch.empty_like(real)
  half = (alpha * 0.5).contiguous()
  call('tg_sample_scale', _p(noise), _p(half), _p(var), _p(delta), b, real.numel() // b, _dt(real), _stream())
  out = torch.empty_like(real)
  call('tg_axpby', _p(real), _p(delta), _p(out), real.numel(), 1.0, 1.0, _dt(real), _stream())
  return out


def mean(x, weight=1.0):
  return MeanFn.apply(x, float(weight))


def abs_diff_mean(a, b, weight=1.0):
  return AbsDiffMeanFn.apply(a, b, float(weight))


def gradient_penalty(g, lam):
  return GradPenaltyFn.apply(g, float(lam))
