"""Parameter store: every variable of one optimiser group lives in ONE flat fp32 HBM buffer (plus
flat grad / Adam m / Adam v buffers), so a step needs one fused Adam launch and one RCCL all-reduce
per group instead of one per variable (the reference applies Adam and tf.add_n per variable:
model/model_inheritor.py:537-542, deployment/model_deploy.py:473-503).

Variables are addressed by the reference's TF names and keep TF layouts
(conv HWIO, fc [in, out]) so reference checkpoints map 1:1.  A variable may have a *physical*
shape larger than its logical one (the D tail conv sees the minibatch-stddev tensor padded to a
multiple of 8 channels); the padding rows are zero and stay zero under Adam.
"""
import math
from collections import OrderedDict

import torch

from .ops import GradSink, PackCache

ALIGN = 64   # elements; keeps every variable 256-byte aligned inside the flat buffer


def get_num_channels(stage, max_num_channels=256):
  """nets/pggan_utils.py:369-372 (python-2 integer division)."""
  return min(1024 // (2 ** stage), max_num_channels)


def max_stage_of(hw):
  """nets/pggan.py:126,218,425."""
  return int(math.log2(hw)) - 2


def mbstd_cpad(c):
  """Physical channel count of the minibatch-stddev output (c + 1 rounded up to a multiple of 8)."""
  return (c + 1 + 7) // 8 * 8


class _ParamDict(dict):
  """name -> parameter tensor; ``state`` holds the non-trainable variables (BatchNorm moving statistics); ``pairs`` the
  stacked [2, ...] views of the two discriminators' twin variables (ParamStore.build)."""
  state = None
  pairs = None


class ParamStore:
  GROUPS = ('g', 'd')    # g: encoder_content + generator (twingan.py:526-527); d: discriminator_* (image_generation.py:484-485)

  def __init__(self, device):
    self.device = torch.device(device)
    self.specs = OrderedDict()      # name -> dict(shape, phys, group, kind)
    self.P = _ParamDict()           # name -> physical leaf tensor (requires_grad)
    self.flat = {}
    self.grad = {}
    self.m = {}
    self.v = {}
    self.offsets = {}
    self.state_specs = OrderedDict()   # name -> (numel, init value): non-trainable variables
    self.state = {}
    self.scalar_state = set()          # names of state variables whose logical (TF) shape is ()
    self.weights_init_stddev = 0.02    # 1.0 under equalized_learning_rate (nets/pggan_utils.py:82-84, pggan.py:364)
    self.renorm = False                # generator_norm_type=batch_renorm: extra non-trainable renorm_* variables
    self.phase_of = None               # name -> backward segment at whose end the gradient is final (grad_phase)
    self.phase_bounds = {}             # group -> {phase: (lo, hi)} element range of each phase in the flat buffers
    self.phase = {}                    # name -> phase
    self.pairs = {}                    # 'discriminator_*/<rest>' -> [2, *phys] view over the two domains' adjacent variables

  # ---- declaration ----------------------------------------------------------------------------
  def add(self, name, shape, group, kind, phys=None):
    assert name not in self.specs, name
    self.specs[name] = dict(shape=tuple(shape), phys=tuple(phys or shape), group=group, kind=kind)

  def add_conv(self, scope, k, cin, cout, group, bias, norm_domains, phys_cin=None, norm_scope='InstanceNorm',
               cond_dim=0):
    self.add(scope + '/weights', (k, k, cin, cout), group, 'conv_w', (k, k, phys_cin or cin, cout))
    if bias:
      self.add(scope + '/biases', (cout,), group, 'bias')
    for d in norm_domains:
      if cond_dim:      # gamma = 1 + FC(cond), beta = FC(cond)  (libs/instance_norm.py:93-120, batch_norm.py:34-38)
        assert norm_scope != NATIVE_NORM      # nets/pggan_utils.py:177,191: contrib's layers take no conditional layer
        for nm in ('gamma', 'beta'):
          self.add(norm_var(scope, norm_scope, nm, d) + '/weights', (cond_dim, cout), group, 'xavier_w')
          self.add(norm_var(scope, norm_scope, nm, d) + '/biases', (cout,), group, 'bias')
      else:
        self.add(norm_var(scope, norm_scope, 'gamma', d), (cout,), group, 'gamma')
        self.add(norm_var(scope, norm_scope, 'beta', d), (cout,), group, 'beta')
      if norm_scope == 'BatchNorm' or (norm_scope == NATIVE_NORM and self.renorm):
        # non-trainable moving statistics (libs/batch_norm.py:184-196; tf.layers.BatchNormalization.build)
        self.state_specs[norm_var(scope, norm_scope, 'moving_mean', d)] = (cout, 0.0)
        self.state_specs[norm_var(scope, norm_scope, 'moving_variance', d)] = (cout, 1.0)
        if self.renorm:                  # batch renorm training statistics (libs/batch_norm.py:209-246), zero-initialised
          self.state_specs[norm_var(scope, norm_scope, 'renorm_mean', d)] = (cout, 0.0)
          self.state_specs[norm_var(scope, norm_scope, 'renorm_mean_weight', d)] = (1, 0.0)
          self.state_specs[norm_var(scope, norm_scope, 'renorm_stddev', d)] = (cout, 0.0)
          self.state_specs[norm_var(scope, norm_scope, 'renorm_stddev_weight', d)] = (1, 0.0)
          # TF creates the two weights as SCALARS (libs/batch_norm.py:237,246; tf.layers.BatchNormalization): the device
          # buffer keeps one element, state_dict / checkpoints carry shape ()
          self.scalar_state.add(norm_var(scope, norm_scope, 'renorm_mean_weight', d))
          self.scalar_state.add(norm_var(scope, norm_scope, 'renorm_stddev_weight', d))

  # ---- allocation -----------------------------------------------------------------------------
  def build(self, seed=0):
    """Lays the variables out phase by phase (``phase_of``): the gradients that one backward segment completes are
    one contiguous range of the group's flat buffer, so the clone all-reduce of that range (dp.GradReducer) can start
    while the next segment is still running.  Initial values are drawn in declaration order whatever the layout."""
    sizes = {g: 0 for g in self.GROUPS}
    phase = {name: (int(self.phase_of(name)) if self.phase_of else 0) for name in self.specs}
    order = sorted(self.specs, key=lambda k: phase[k])      # stable: declaration order inside a phase
    # The two discriminators are towers of identical layers (twingan.py:105-110): a variable of discriminator_t goes right
    # behind its discriminator_s twin, so that the pair is ONE dense [2, ...] tensor (self.pairs) -- what the grouped convs
    # (TgConvDesc.groups: both discriminators' layer as one launch) read their two weight sets from.  Values are drawn in
    # declaration order whatever the layout, names / shapes / checkpoints are untouched.
    twins = {}
    for name in order:
      if name.startswith('discriminator_s/'):
        t = 'discriminator_t/' + name[len('discriminator_s/'):]
        n = int(math.prod(self.specs[name]['phys']))
        if t in self.specs and self.specs[t]['phys'] == self.specs[name]['phys'] and phase[t] == phase[name] and n % 4 == 0:
          twins[name] = t
    moved = set(twins.values())
    order = [k for name in order if name not in moved for k in ((name, twins[name]) if name in twins else (name,))]
    marks = {g: {} for g in self.GROUPS}
    second = set(twins.values())
    for name in order:
      s = self.specs[name]
      n = int(math.prod(s['phys']))
      self.offsets[name] = sizes[s['group']]
      marks[s['group']].setdefault(phase[name], sizes[s['group']])
      # a first twin is followed by its second without padding (16-byte aligned: n % 4 == 0); the pair is padded as a whole
      sizes[s['group']] += n if name in twins else (n + ALIGN - 1) // ALIGN * ALIGN
      if name in second:
        sizes[s['group']] = (sizes[s['group']] + ALIGN - 1) // ALIGN * ALIGN
    for g in self.GROUPS:
      ids = sorted(marks[g]) or [0]
      starts = [marks[g].get(p, 0) for p in ids]
      self.phase_bounds[g] = dict(zip(ids, zip(starts, starts[1:] + [max(sizes[g], ALIGN)])))
    self.phase = phase
    for g in self.GROUPS:
      n = max(sizes[g], ALIGN)
      self.flat[g] = torch.zeros(n, dtype=torch.float32, device=self.device)
      self.grad[g] = torch.zeros(n, dtype=torch.float32, device=self.device)
      self.m[g] = torch.zeros(n, dtype=torch.float32, device=self.device)
      self.v[g] = torch.zeros(n, dtype=torch.float32, device=self.device)
    gen = torch.Generator().manual_seed(seed)
    for name, s in self.specs.items():
      g, off, n = s['group'], self.offsets[name], int(math.prod(s['phys']))
      p = self.flat[g][off:off + n].view(s['phys'])
      self._init(p, s, gen)
      p.requires_grad_(True)
      p.grad = self.grad[g][off:off + n].view(s['phys'])
      GradSink.register(p, p.grad)          # backward kernels accumulate straight into the flat buffer
      self.P[name] = p
      if s['kind'] == 'conv_w':
        PackCache.register(p)
    for name, t in twins.items():
      s = self.specs[name]
      g, off, n = s['group'], self.offsets[name], int(math.prod(s['phys']))
      assert self.offsets[t] == off + n, (name, t)
      pr = self.flat[g][off:off + 2 * n].view((2,) + s['phys'])
      pr.requires_grad_(True)
      pr.grad = self.grad[g][off:off + 2 * n].view((2,) + s['phys'])
      GradSink.register(pr, pr.grad)      # keyed by (address, numel): distinct from the first twin's own sink
      # no PackCache.register: the pair starts at the first twin's address, whose registration covers its packs too
      self.pairs['discriminator_*/' + name[len('discriminator_s/'):]] = pr
    self.P.pairs = self.pairs
    for name, (n, init) in self.state_specs.items():
      if init == 'trunc_normal':      # tf.truncated_normal_initializer(): N(0,1) redrawn outside 2 sigma
        t = torch.empty(n, dtype=torch.float32)
        torch.nn.init.trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=gen)
        self.state[name] = t.to(self.device)
      else:
        self.state[name] = torch.full((n,) if isinstance(n, int) else tuple(n), init, dtype=torch.float32,
                                      device=self.device)
    if self.renorm:      # clipping bounds of the current global step (nets/pggan_utils.py:207-223), set by the trainer
      for k, v in (('renorm/rmax', 1.1), ('renorm/rmin', 0.9), ('renorm/dmax', 0.1)):
        self.state[k] = torch.full((1,), v, dtype=torch.float32, device=self.device)
    if 'gdrop_strength' in self.state_specs:      # the controller's coefficient of the current global step (Trainer._set_gdrop_coef)
      self.state['gdrop/coef'] = torch.zeros(1, dtype=torch.float32, device=self.device)
    self.P.state = self.state
    PackCache.version += 1
    return self

  def _init(self, p, s, gen):
    """weights ~ N(0, 0.02) (nets/pggan_utils.py:56,93; pggan.py:364-368; N(0,1) when equalized), biases/beta 0,
    gamma 1."""
    with torch.no_grad():
      if s['kind'] in ('conv_w', 'fc_w'):
        p.zero_()
        w = torch.randn(s['shape'], generator=gen, dtype=torch.float32) * self.weights_init_stddev
        self._logical(p, s).copy_(w.to(p.device))
      elif s['kind'] == 'xavier_w':      # layers.fully_connected default: xavier_initializer (uniform)
        lim = math.sqrt(6.0 / (s['shape'][0] + s['shape'][1]))
        p.copy_(((torch.rand(s['shape'], generator=gen, dtype=torch.float32) * 2.0 - 1.0) * lim).to(p.device))
      elif s['kind'] == 'gamma':
        p.fill_(1.0)
      else:
        p.zero_()

  @staticmethod
  def _logical(p, s):
    if s['phys'] == s['shape']:
      return p
    return p[tuple(slice(0, d) for d in s['shape'])]

  # ---- access ---------------------------------------------------------------------------------
  def __getitem__(self, name):
    return self.P[name]

  def __contains__(self, name):
    return name in self.P

  def names(self, group=None):
    return [k for k, s in self.specs.items() if group is None or s['group'] == group]

  def state_dict(self, include_state=False):
    """Logical (reference-shaped) copies keyed by TF variable names.  ``include_state``: also the non-trainable
    variables (BatchNorm moving / renorm statistics, spectral-norm u) -- what a TF checkpoint of the stage holds."""
    sd = {k: self._logical(self.P[k].detach(), s).clone() for k, s in self.specs.items()}
    if include_state:
      sd.update({k: (v.reshape(()) if k in self.scalar_state else v).clone() for k, v in self.state.items()
                 if k in self.state_specs})
    return sd

  def state_shape(self, k):
    """Logical (TF checkpoint) shape of a non-trainable variable."""
    return () if k in self.scalar_state else tuple(self.state[k].shape)

  def adam_dict(self):
    """{variable name: (m, v)} -- logical views of the shared Adam optimiser's slot variables (TF: <var>/Adam,
    <var>/Adam_1)."""
    out = {}
    for k, s in self.specs.items():
      g, off, n = s['group'], self.offsets[k], int(math.prod(s['phys']))
      out[k] = (self._logical(self.m[g][off:off + n].view(s['phys']), s).clone(),
                self._logical(self.v[g][off:off + n].view(s['phys']), s).clone())
    return out

  def load_adam_dict(self, slots):
    """Inverse of adam_dict for the names given ({name: (m, v)} of logical shape)."""
    with torch.no_grad():
      for k, (m, v) in slots.items():
        s = self.specs[k]
        g, off, n = s['group'], self.offsets[k], int(math.prod(s['phys']))
        for dst, src in ((self.m[g], m), (self.v[g], v)):
          self._logical(dst[off:off + n].view(s['phys']), s).copy_(torch.as_tensor(src, dtype=torch.float32).reshape(s['shape']))

  def grad_dict(self):
    GradSink.flush()                             # filter gradients held back for pairing
    if self.device.type == 'cuda':
      torch.cuda.synchronize(self.device)      # gradient sinks are written by kernels on several streams
    return {k: self._logical(self.P[k].grad, s).clone() for k, s in self.specs.items()}

  def load_state_dict(self, sd, strict=True):
    """``ignore_missing_vars`` semantics of pggan_runner.py:136-146 when strict is False."""
    with torch.no_grad():
      for k, s in self.specs.items():
        if k not in sd:
          if strict:
            raise KeyError(k)
          continue
        src = torch.as_tensor(sd[k]).to(device=self.device, dtype=torch.float32)
        assert tuple(src.shape) == s['shape'], (k, tuple(src.shape), s['shape'])
        self.P[k].zero_() if s['phys'] != s['shape'] else None
        self._logical(self.P[k], s).copy_(src)
      for k in self.state_specs:      # non-trainable variables, when the dict carries them
        if k in sd and tuple(torch.as_tensor(sd[k]).shape) in (tuple(self.state[k].shape), self.state_shape(k)):
          self.state[k].copy_(torch.as_tensor(sd[k]).to(device=self.device, dtype=torch.float32).reshape(self.state[k].shape))
    PackCache.version += 1

  def zero_grad(self, group):
    if self.grad[group].is_cuda:
      from . import ops
      ops.zero_(self.grad[group])
    else:
      self.grad[group].zero_()

  def close(self):
    """Drops this store's entries from the process-wide pack / gradient-sink registries (they hold strong references
    to the parameter and gradient views): call when a stage's trainer is done (runner.run_progressive)."""
    for p in self.P.values():
      GradSink.unregister(p)
      PackCache.unregister(p)
    for p in self.pairs.values():
      GradSink.unregister(p)
    for buf in self.P.__dict__.pop('sn_wbar', {}).values():      # persistent spectrally-normalised kernels (pggan._sn_compute)
      PackCache.unregister(buf)

  def numel(self, group):
    return sum(int(math.prod(s['shape'])) for s in self.specs.values() if s['group'] == group)


_HW_IN_NAME = __import__('re').compile(r'/(?:encoder_block|from_rgb|self_attention|block|generator_to_rgb)_(\d+)x\d+')


def is_model_variable(name):
  """Is ``name`` in slim's MODEL_VARIABLES collection, i.e. among what a stage's warm start restores
  (slim.get_model_variables() in model/model_inheritor.py:612-614)?  Every variable of the path is created through slim
  layers / variables.model_variable (libs/instance_norm.py:101,121; libs/batch_norm.py:142-224), and so is the
  spectral-norm vector ``u``: libs/sn.py:56 asks tf.get_variable for ``collections=tf.GraphKeys.MODEL_VARIABLES`` from
  inside the layer's variable scope, whose custom getter (libs/sn.py:199-204, layers._build_variable_getter) routes the
  request through slim's model_variable, which appends MODEL_VARIABLES itself -- a stage's warm start restores ``u``.
  The one exception is the attention gate ``sa_gamma`` (libs/self_attention.py:68): plain tf.get_variable outside any
  layer scope, so every stage of the reference starts it from 0 again; a full Saver restore (resuming a run,
  inference) still loads it."""
  return not name.endswith('/sa_gamma')


def grad_phase(name, cfg):
  """Backward segment (twingan.Trainer._grad_segments) at whose end the gradient of variable ``name`` is complete,
  for the cut resolution ``cfg.overlap_cut_hw``:

    generator step  0: generator/*  (re-encode pass, discriminators and the generator are differentiated first)
                    1: encoder layers at <= cut_hw        2: encoder layers above cut_hw (their backward runs last)
    discriminator   0: layers at <= cut_hw, the tail and the FC (gradient penalty passes + low-resolution part of the
                       batched pass)                      1: layers above cut_hw

  Growing stages: the shrink path (from_rgb at hw / 2, nets/pggan.py:233-240,395-399) is blended in ABOVE the cut, so its
  variables complete with the full-resolution block whatever hw / 2 is.  The style encoder (twingan.py:201-223) is not cut:
  its gradients are final after segment 0 and travel with the encoder ranges (later than necessary, never too early)."""
  if not cfg.overlap_cut_hw or cfg.hw <= cfg.overlap_cut_hw:
    return 0
  top = name.split('/', 1)[0]
  if top == 'generator':
    return 0
  m = _HW_IN_NAME.search(name)
  high = bool(m) and int(m.group(1)) > cfg.overlap_cut_hw
  if cfg.is_growing and ('/from_rgb_%dx%d/' % (cfg.hw // 2, cfg.hw // 2)) in name:
    high = True
  if top.startswith('encoder'):
    return 2 if high else 1
  return 1 if high else 0


NATIVE_NORM = '@native'
NORM_SCOPE = {'instance_norm': 'InstanceNorm', 'batch_norm': 'BatchNorm', 'batch_renorm': 'BatchNorm', 'none': '',
              'batch_renorm_native': NATIVE_NORM, 'layer_norm_native': NATIVE_NORM}


def norm_var(scope, norm_scope, name, domain):
  """TF name of a normaliser variable of the conv at ``scope``.  The reference's own layers open '<conv>/InstanceNorm' |
  '<conv>/BatchNorm' and append the domain postfix to the VARIABLE name (libs/instance_norm.py:66-120,
  libs/batch_norm.py:80,130-246: 'gamma_s'); tf.contrib's layers behind 'batch_renorm_native' / 'layer_norm_native' get
  the postfix as their SCOPE (nets/pggan_utils.py:187,196) and keep contrib's plain variable names: '<conv>/_s/gamma'."""
  pf = '_' + domain if domain else ''
  if norm_scope == NATIVE_NORM:
    return '%s/%s/%s' % (scope, pf, name)
  return '%s/%s/%s%s' % (scope, norm_scope, name, pf)


def declare_pggan(store, cfg):
  """The plain PGGAN trainer's variables (image_generation.py:194-316: scopes 'generator' and 'discriminator', latent
  noise input, no encoder, no domain postfix on the normaliser variables) -- BASELINE configs[0]."""
  return declare_twingan(store, cfg, model='pggan')


def declare_twingan(store, cfg, model='twingan'):
  """All TwinGAN variables of one progressive stage (scopes twingan.py:105-110; layer lists
  SURVEY.md Appendix A; nets/pggan.py:93-211,242-376,403-479).  ``model='pggan'``: see declare_pggan."""
  hw, mc = cfg.hw, cfg.max_ch
  pggan_model = model == 'pggan'
  store.phase_of = (lambda name: 0) if pggan_model else (lambda name: grad_phase(name, cfg))
  ms = max_stage_of(hw)
  store.renorm = cfg.generator_norm_type in ('batch_renorm', 'batch_renorm_native')
  if cfg.generator_norm_type not in NORM_SCOPE:
    raise NotImplementedError('generator_norm_type=%s' % cfg.generator_norm_type)
  if NORM_SCOPE[cfg.generator_norm_type] == NATIVE_NORM and pggan_model:
    # image_generation.py leaves conditional_layer_var_scope_postfix at '' (nets/pggan_utils.py:102-113), and contrib's
    # layers would open variable_scope('') -- TF then names the variables '<conv>//gamma'; not restated, not built
    raise NotImplementedError('generator_norm_type=%s in the plain PGGAN trainer (empty scope postfix)' % cfg.generator_norm_type)
  # generator_norm_type=none (nets/pggan_utils.py:198-200): no normaliser, so slim's conv2d adds a bias instead
  g_bias = cfg.generator_norm_type == 'none'
  nd = () if g_bias else (('',) if pggan_model else ('s', 't'))
  ns = NORM_SCOPE[cfg.generator_norm_type]
  rk = min(7, hw // 2) if cfg.use_larger_filter_at_rgb_layer else 1      # nets/pggan.py:172-175,194-197
  if cfg.equalized_learning_rate:
    store.weights_init_stddev = 1.0

  if cfg.use_gdrop:      # twingan.py:861-865 / image_generation.py:1034-1038: slim.model_variable('gdrop_strength', shape=[], zeros)
    store.state_specs['gdrop_strength'] = (1, 0.0)
    store.scalar_state.add('gdrop_strength')

  def sn_state(scope, cout, is_disc):
    """--spectral_norm: the power-iteration vector 'u' [1, cout] of a conv (libs/sn.py:56-57, truncated normal)."""
    if cfg.spectral_norm and (is_disc or cfg.spectral_norm_in_non_discriminator):
      store.state_specs[scope + '/u'] = ((1, cout), 'trunc_normal')

  _add_conv = store.add_conv

  def add_conv(scope, k, cin, cout, group, bias, norm_domains, **kw):
    _add_conv(scope, k, cin, cout, group, bias, norm_domains, **kw)
    sn_state(scope, cout, group == 'd')

  store.add_conv = add_conv

  def attention(top, hw_, c_, name_c, group, bias, norm_domains, **kw):
    """--do_self_attention: sa_f / sa_g (c -> c/8), sa_h (c -> c) 1x1 convs under the scope's arg-scope (normaliser
    in G/E, bias in D) and the scalar sa_gamma (libs/self_attention.py:24-70; nets/pggan_utils.py:301-308).  They go
    through libs.sn.convolution with do_spec_norm False: never spectrally normed."""
    if not (cfg.do_self_attention and hw_ == cfg.self_attention_hw):
      return
    sc = '%s/self_attention_%dx%dx%d' % (top, hw_, hw_, name_c)
    for nm, co in (('sa_f', c_ // 8), ('sa_g', c_ // 8), ('sa_h', c_)):
      _add_conv('%s/%s' % (sc, nm), 1, c_, co, group, bias, norm_domains, norm_scope=ns, **kw)
    store.add(sc + '/sa_gamma', (1,), group, 'beta')

  def shortcut(blk, cin, cout, group):
    """--use_res_block: 1x1 'shortcut' conv (+bias, no norm) where a block changes the channel count
    (nets/pggan_utils.py:334-342)."""
    if cfg.use_res_block and cin != cout:
      store.add_conv(blk + '/shortcut', 1, cin, cout, group, True, ())

  def enc_skeleton(top, group, bias, norm_domains, mc=mc):
    if cfg.is_growing:
      store.add_conv('%s/from_rgb_%dx%d/Conv' % (top, hw // 2, hw // 2), 1, 3, get_num_channels(ms - 1, mc), group, bias,
                     norm_domains, norm_scope=ns)
      shortcut('%s/from_rgb_%dx%d' % (top, hw // 2, hw // 2), 3, get_num_channels(ms - 1, mc), group)
    c = get_num_channels(ms, mc)
    store.add_conv('%s/from_rgb_%dx%d/Conv' % (top, hw, hw), 1, 3, c, group, bias, norm_domains, norm_scope=ns)
    shortcut('%s/from_rgb_%dx%d' % (top, hw, hw), 3, c, group)
    for stage in range(ms, 0, -1):
      cur = hw // (2 ** (ms - stage))
      nc = get_num_channels(stage - 1, mc)
      attention(top, cur, c, nc, group, bias, norm_domains)
      blk = '%s/encoder_block_%dx%dx%d' % (top, cur, cur, nc)
      store.add_conv(blk + '/Conv', 3, c, c, group, bias, norm_domains, norm_scope=ns)
      store.add_conv(blk + '/Conv_1', 3, c, nc, group, bias, norm_domains, norm_scope=ns)
      shortcut(blk, c, nc, group)
      c = nc

  if not pggan_model:
    enc_skeleton('encoder_content', 'g', g_bias, nd)
    if cfg.do_encoder_distillation:      # twingan.py:207-230: encoder_classification heads under encoder_content, one per domain
      assert cfg.distill_embed_dim > 0, 'do_encoder_distillation needs distill_embed_dim (the dataset embedding width)'
      c0 = get_num_channels(0, mc)
      for head, d in (('encoder_content/encoder_distillation_source', 's'), ('encoder_content/encoder_distillation_target', 't')):
        hd = () if g_bias else (d,)
        store.add_conv('%s/before_fc_1x1x%d/Conv' % (head, mc), 3, c0, mc, 'g', g_bias, hd, norm_scope=ns)
        store.add_conv('%s/before_fc_1x1x%d/Conv_1' % (head, mc), 4, mc, mc, 'g', g_bias, hd, norm_scope=ns)
        store.add(head + '/prediction/fully_connected/weights', (mc, cfg.distill_embed_dim), 'g', 'fc_w')
        store.add(head + '/prediction/fully_connected/biases', (cfg.distill_embed_dim,), 'g', 'bias')
  gen_kw = {}
  if cfg.use_style_embedding and not pggan_model:      # twingan.py:47-51,201-223: the style encoder (pggan.encoder) and conditional generator norms
    if cfg.generator_norm_type not in ('instance_norm', 'batch_norm', 'batch_renorm'):
      raise NotImplementedError('use_style_embedding with generator_norm_type=%s' % cfg.generator_norm_type)
    enc_skeleton('encoder_style', 'g', g_bias, nd)
    c0 = get_num_channels(0, mc)
    store.add_conv('encoder_style/before_fc_1x1x%d/Conv' % mc, 3, c0, mc, 'g', g_bias, nd, norm_scope=ns)
    store.add_conv('encoder_style/before_fc_1x1x%d/Conv_1' % mc, 4, mc, mc, 'g', g_bias, nd, norm_scope=ns)
    store.add('encoder_style/prediction/fully_connected/weights', (mc, cfg.style_embed_size), 'g', 'fc_w')
    store.add('encoder_style/prediction/fully_connected/biases', (cfg.style_embed_size,), 'g', 'bias')
    gen_kw = dict(cond_dim=cfg.style_embed_size)
  # generator
  c = get_num_channels(0, mc)
  blk = 'generator/block_4x4x%d' % c
  if pggan_model:      # latent noise [B,1,1,get_num_channels(1)] padded to 7x7, 4x4 VALID (nets/pggan.py:135-153)
    store.add_conv(blk + '/Conv', 4, get_num_channels(1, mc), c, 'g', g_bias, nd, norm_scope=ns)
  else:
    store.add_conv(blk + '/Conv', 3, c, c, 'g', g_bias, nd, norm_scope=ns, **gen_kw)
  store.add_conv(blk + '/Conv_1', 3, c, c, 'g', g_bias, nd, norm_scope=ns, **gen_kw)
  attention('generator', 4, c, c, 'g', g_bias, nd, **gen_kw)
  for stage in range(1, ms + 1):
    cur = 2 ** (stage + 2)
    oc = get_num_channels(stage, mc)
    if stage == ms and cfg.is_growing:
      store.add_conv('generator/generator_to_rgb_%dx%d/Conv' % (cur // 2, cur // 2), rk, c, 3, 'g', g_bias, nd, norm_scope=ns, **gen_kw)
    skip = (cfg.use_unet and not pggan_model and
            not (cfg.unet_max_concat_hw and cur > cfg.unet_max_concat_hw))      # pggan_utils.py:287-289
    cin = c + (get_num_channels(stage - 1, mc) if skip else 0)
    blk = 'generator/block_%dx%dx%d' % (cur, cur, oc)
    store.add_conv(blk + '/Conv', 3, cin, oc, 'g', g_bias, nd, norm_scope=ns, **gen_kw)
    store.add_conv(blk + '/Conv_1', 3, oc, oc, 'g', g_bias, nd, norm_scope=ns, **gen_kw)
    shortcut(blk, cin, oc, 'g')
    attention('generator', cur, oc, oc, 'g', g_bias, nd, **gen_kw)
    c = oc
  store.add_conv('generator/generator_to_rgb_%dx%d/Conv' % (hw, hw), rk, c, 3, 'g', g_bias, nd, norm_scope=ns, **gen_kw)
  # discriminators
  for top in (('discriminator',) if pggan_model else ('discriminator_s', 'discriminator_t')):
    md = cfg.max_ch_dis or mc      # get_discriminator_max_num_channels (nets/pggan_utils.py:375-380)
    enc_skeleton(top, 'd', True, (), md)
    blk = '%s/before_fc_1x1x%d' % (top, md)
    store.add_conv(blk + '/Conv', 3, md + 1, md, 'd', True, (), phys_cin=mbstd_cpad(md))
    store.add_conv(blk + '/Conv_1', 4, md, md, 'd', True, ())
    store.add(top + '/prediction/fully_connected/weights', (md, 1), 'd', 'fc_w')
    store.add(top + '/prediction/fully_connected/biases', (1,), 'd', 'bias')
  store.add_conv = _add_conv
  return store
