"""torch-CPU restatement of the TwinGAN training graph (oracle, test-only; also the timed CPU baseline).

Parity status: PINNED against the reference's own graph code (oracle/ref_runner.py, tests/golden/twingan_*.npz;
TensorFlow-op semantics restated) -- see ``oracle/__init__.py``.  Stock ``torch.nn.functional`` ops only; autograd
supplies the reference gradients (incl. the WGAN-GP double backward).  Works in float32 or float64
(dtype follows the parameters).  Public tensors are NHWC; parameters are keyed by the reference's
TF variable names with TF layouts (conv HWIO, fc [in, out]).

Restated files (all relative to /root/reference): nets/pggan.py, nets/pggan_utils.py,
libs/instance_norm.py, util_misc.py:68-86, twingan.py:146-521,820-891,
image_generation.py:318-439,543-662,1001-1006, model/model_inheritor.py:537-542,
deployment/model_deploy.py:242-315.
"""
import math

import numpy as np
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class Config:
  """The flags that shape the hot path (defaults = BASELINE.json / SURVEY.md section 8d)."""
  hw: int = 256                      # train_image_size
  max_ch: int = 256                  # pggan_max_num_channels            (nets/pggan.py:51-53)
  max_ch_dis: object = None          # pggan_max_num_channels_dis        (nets/pggan.py:54-56; pggan_utils.py:375-380)
  norm: str = 'instance_norm'        # generator_norm_type               (nets/pggan.py:24)
  do_pixel_norm: bool = True         # nets/pggan.py:34-38
  use_unet: bool = True              # twingan.py:53-56
  unet_max_concat_hw: object = None  # pggan_unet_max_concat_hw (nets/pggan.py:57-59): no skip above this hw
  is_growing: bool = False           # image_generation.py:69-72
  alpha_grow: float = 0.0
  loss: str = 'wgan_gp'              # loss_architecture                 (image_generation.py:81-83)
  gp_lambda: float = 10.0            # image_generation.py:92-95
  gan_weight: float = 1.0            # image_generation.py:84-86
  drift: float = 0.0                 # wgan_drift_loss_weight            (image_generation.py:96-98)
  l_cyc: float = 1.0                 # twingan.py:73-76
  l_content: float = 0.1             # twingan.py:80-82
  do_l_cyc_gan: bool = True          # twingan.py:77-79
  lr: float = 1e-4                   # docs/training.md:24-25
  beta1: float = 0.5
  beta2: float = 0.99
  adam_eps: float = 1e-8
  use_ttur: bool = False             # image_generation.py:554-561: own optimizer (rate, beta powers) for the discriminator
  d_lr: float = 4e-4                 # discriminator_learning_rate
  in_eps: float = 1e-6               # libs/instance_norm.py:37
  pn_eps: float = 1e-6               # nets/pggan_utils.py:330
  lrelu: float = 0.2                 # util_misc.py:68
  bn_state: object = None            # dict collecting the BatchNorm moving (and renorm) statistics when set
  do_encoder_distillation: bool = False   # twingan.py:58-65: content encoder distils dataset-provided embeddings
  distillation_weight: float = 1.0
  distillation_start_hw: int = 16
  distill_embed_dim: int = 0         # width of the 'a_embedding' / 'b_embedding' dataset fields
  distill_embed_s: object = None     # [B, D] embeddings of the source / target batch (None: that domain has none)
  distill_embed_t: object = None
  is_training: bool = True           # False: the inference branch (twingan.py:300-363): BatchNorm uses the moving statistics
  global_step: int = 0               # batch renorm clipping schedule (nets/pggan_utils.py:207-223)
  spectral_norm: bool = False        # nets/pggan.py:28-30 (discriminator convs; libs/sn.py:38-101)
  sn_non_disc: bool = False          # spectral_norm_in_non_discriminator (nets/pggan.py:31-33): encoder / generator convs too
  larger_rgb: bool = False           # use_larger_filter_at_rgb_layer (nets/pggan.py:47-50,172-175,194-197)
  sn_state: object = None            # dict scope -> u [1, cout] (libs/sn.py:56-57); updated by end_run()
  sn_cache: object = None            # per-run normalised kernels: every use in a run sees the pre-run u
  do_self_attention: bool = False    # image_generation.py:62-64
  self_attention_hw: int = 64        # image_generation.py:65-67
  use_style_embedding: bool = False  # twingan.py:47-49
  style_embed_size: int = 16         # twingan.py:50-51
  # gdrop (libs/gdrop.py:20-36; nets/pggan.py:340-355): the layer runs only under do_dgrop (no reference trainer sets it)
  do_dgrop: bool = False
  gdrop_strength: float = 0.0
  gdrop_noise: object = None         # list of [N, C] N(0,1) draws consumed in call order (drawn when None / exhausted)
  style_noise: object = None         # the N(0,1) random_style_embed [B, E] of twingan.py:232-235 (drawn when None)
  equalized: bool = False            # equalized_learning_rate           (nets/pggan.py:40; pggan_utils.py:236-254)
  res_block: bool = False            # use_res_block                     (nets/pggan.py:44; pggan_utils.py:257-264,334-342)


def get_num_channels(stage, max_num_channels=256):
  """nets/pggan_utils.py:369-372 (py2 integer division)."""
  return min(1024 // (2 ** stage), max_num_channels)


def max_stage_of(hw):
  """nets/pggan.py:126,218,425."""
  return int(math.log2(hw)) - 2


# ------------------------------------------------------------------------------------------------
# parameter construction (names: SURVEY.md Appendix C; init: nets/pggan_utils.py:56,93, pggan.py:364-368)
# ------------------------------------------------------------------------------------------------
NATIVE = '@native'  # tf.contrib's own layers, called with scope=<postfix> (nets/pggan_utils.py:175-197): '<conv>/_s/gamma'
NORM_SCOPE = {'instance_norm': 'InstanceNorm', 'batch_norm': 'BatchNorm', 'batch_renorm': 'BatchNorm',      # libs/instance_norm.py:66, batch_norm.py:80
              'batch_renorm_native': NATIVE, 'layer_norm_native': NATIVE}
LN_EPS = 1e-12      # tf.contrib.layers.layer_norm's variance_epsilon (TF 1.8 contrib/layers/python/layers/layers.py)
BN_EPS = 1e-3       # libs/batch_norm.py:48 (max(epsilon, 1.001e-5), :464-468)
BN_DECAY = 0.999    # libs/batch_norm.py:45


def _pf(domain):
  """Variable-name postfix of a domain: '_s' / '_t' in TwinGAN, '' in the plain PGGAN trainer
  (conditional_layer_var_scope_postfix, nets/pggan_utils.py:102-113)."""
  return '_' + domain if domain else ''


def norm_var(scope, norm_scope, name, domain):
  """Name of a normaliser variable of the conv at ``scope``.  The reference's own layers (libs/instance_norm.py:66,
  libs/batch_norm.py:80,130-196) open '<conv>/InstanceNorm' | '<conv>/BatchNorm' and append the domain postfix to the
  VARIABLE name; tf.contrib's layers, which 'batch_renorm_native' / 'layer_norm_native' call with scope=<postfix>
  (nets/pggan_utils.py:175-197), have no postfix argument: the postfix IS the scope, the variables keep contrib's plain
  names ('<conv>/_s/gamma').  The plain PGGAN trainer's postfix '' would make that scope the empty string."""
  if norm_scope == NATIVE:
    if not domain:
      raise NotImplementedError("a native normaliser with an empty postfix opens variable_scope(''): not restated")
    return '%s/%s/%s' % (scope, _pf(domain), name)
  return '%s/%s/%s%s' % (scope, norm_scope, name, _pf(domain))


def _conv_p(P, g, scope, k, cin, cout, norm_domains, bias, dtype, std=0.02, norm_scope='InstanceNorm'):
  if std == 'he':                    # test-only: O(1) activations so parity errors are visible
    std = math.sqrt(2.0 / (k * k * cin))
  P[scope + '/weights'] = torch.randn(k, k, cin, cout, generator=g, dtype=torch.float32).to(dtype) * std
  if bias:
    P[scope + '/biases'] = torch.zeros(cout, dtype=dtype)
  for d in norm_domains:
    P[norm_var(scope, norm_scope, 'gamma', d)] = torch.ones(cout, dtype=dtype)
    P[norm_var(scope, norm_scope, 'beta', d)] = torch.zeros(cout, dtype=dtype)


def encoder_param_specs(top, hw, max_ch, growing=False):
  """[(scope, k, cin, cout)] for nets/pggan.py:403-479 (also the D skeleton, :242-315)."""
  ms = max_stage_of(hw)
  specs = []
  if growing:
    specs.append(('%s/from_rgb_%dx%d/Conv' % (top, hw // 2, hw // 2), 1, 3, get_num_channels(ms - 1, max_ch)))
  c = get_num_channels(ms, max_ch)
  specs.append(('%s/from_rgb_%dx%d/Conv' % (top, hw, hw), 1, 3, c))
  for stage in range(ms, 0, -1):
    cur = hw // (2 ** (ms - stage))
    nc = get_num_channels(stage - 1, max_ch)
    blk = '%s/encoder_block_%dx%dx%d' % (top, cur, cur, nc)
    specs.append((blk + '/Conv', 3, c, c))
    specs.append((blk + '/Conv_1', 3, c, nc))
    c = nc
  return specs


def rgb_kernel_size(larger_rgb, hw):
  """nets/pggan.py:172-175,194-197: the to-RGB kernel is min(7, hw / 2) with --use_larger_filter_at_rgb_layer, else 1 --
  computed from the CURRENT stage's hw for the grown layer and for the previous-resolution layer alike (Python-2 integer
  division; 8 x 8 gives an EVEN 4 x 4 SAME kernel: TF pads 1 before, 2 after)."""
  return min(7, hw // 2) if larger_rgb else 1


def generator_param_specs(top, hw, max_ch, use_unet, growing=False, unet_max_hw=None, larger_rgb=False):
  """[(scope, k, cin, cout)] for nets/pggan.py:93-211 with a [B,4,4,C] source."""
  ms = max_stage_of(hw)
  specs = []
  c = get_num_channels(0, max_ch)
  blk = '%s/block_4x4x%d' % (top, c)
  specs.append((blk + '/Conv', 3, c, c))      # source channels == ch(0) in TwinGAN
  specs.append((blk + '/Conv_1', 3, c, c))
  for stage in range(1, ms + 1):
    cur = 2 ** (stage + 2)
    oc = get_num_channels(stage, max_ch)
    if stage == ms and growing:
      specs.append(('%s/generator_to_rgb_%dx%d/Conv' % (top, cur // 2, cur // 2), rgb_kernel_size(larger_rgb, cur), c, 3))
    cin = c + (get_num_channels(stage - 1, max_ch) if (use_unet and not (unet_max_hw and cur > unet_max_hw)) else 0)
    blk = '%s/block_%dx%dx%d' % (top, cur, cur, oc)
    specs.append((blk + '/Conv', 3, cin, oc))
    specs.append((blk + '/Conv_1', 3, oc, oc))
    c = oc
  specs.append(('%s/generator_to_rgb_%dx%d/Conv' % (top, hw, hw), rgb_kernel_size(larger_rgb, hw), c, 3))
  return specs


def shortcut_specs(specs, hw, max_ch, use_unet=False):
  """[(scope, 1, cin, cout)] of the 1x1 'shortcut' convs --use_res_block adds where a block changes the channel
  count (nets/pggan_utils.py:334-342): from_rgb blocks (3 -> C), encoder / discriminator two-layer blocks (block
  input vs out_channels) and generator three-layer blocks (upsampled [+ UNet concat] input vs out_channels)."""
  out = []
  for scope, k, cin, cout in specs:
    blk, leaf = scope.rsplit('/', 1)
    name = blk.rsplit('/', 1)[1]
    if name.startswith('from_rgb_') and cin != cout:
      out.append((blk + '/shortcut', 1, cin, cout))
    elif name.startswith('encoder_block_') and leaf == 'Conv':
      nc = int(name.rsplit('x', 1)[1])
      if cin != nc:
        out.append((blk + '/shortcut', 1, cin, nc))
    elif name.startswith('block_') and leaf == 'Conv' and not name.startswith('block_4x4x') and cin != cout:
      out.append((blk + '/shortcut', 1, cin, cout))
  return out


def discriminator_tail_specs(top, max_ch):
  blk = '%s/before_fc_1x1x%d' % (top, max_ch)
  return [(blk + '/Conv', 3, max_ch + 1, max_ch), (blk + '/Conv_1', 4, max_ch, max_ch)]


def init_params(cfg, seed=0, dtype=torch.float32, std=0.02):
  """All TwinGAN variables for one progressive stage (twingan.py:105-110 scopes).  ``std=0.02`` is the
  reference initialiser (1.0 with equalized_learning_rate, nets/pggan_utils.py:82-84, pggan.py:364); ``std='he'``
  is a test-only variant (also randomises biases / gamma / beta)."""
  g = torch.Generator().manual_seed(seed)
  P = {}
  he = std == 'he'
  if cfg.equalized:
    std = 1.0
  for s in encoder_param_specs('encoder_content', cfg.hw, cfg.max_ch, cfg.is_growing):
    _conv_p(P, g, s[0], s[1], s[2], s[3], ('s', 't') if cfg.norm in NORM_SCOPE else (), cfg.norm not in NORM_SCOPE, dtype,
            std, NORM_SCOPE.get(cfg.norm, ''))      # no normaliser: slim's conv2d owns a bias instead
  for s in generator_param_specs('generator', cfg.hw, cfg.max_ch, cfg.use_unet, cfg.is_growing, cfg.unet_max_concat_hw,
                                    cfg.larger_rgb):
    _conv_p(P, g, s[0], s[1], s[2], s[3], ('s', 't') if cfg.norm in NORM_SCOPE else (), cfg.norm not in NORM_SCOPE, dtype,
            std, NORM_SCOPE.get(cfg.norm, ''))
  md = cfg.max_ch_dis or cfg.max_ch
  for top in ('discriminator_s', 'discriminator_t'):
    for s in encoder_param_specs(top, cfg.hw, md, cfg.is_growing) + discriminator_tail_specs(top, md):
      _conv_p(P, g, s[0], s[1], s[2], s[3], (), True, dtype, std)
    P[top + '/prediction/fully_connected/weights'] = \
        torch.randn(md, 1, generator=g, dtype=torch.float32).to(dtype) * \
        (math.sqrt(1.0 / md) if std == 'he' else std)
    P[top + '/prediction/fully_connected/biases'] = torch.zeros(1, dtype=dtype)
  if cfg.do_self_attention:      # libs/self_attention.py:24-70 under each network's arg-scope; sa_gamma starts at 0
    ms = max_stage_of(cfg.hw)
    nd = ('s', 't') if cfg.norm in NORM_SCOPE else ()

    def att(top, hw_, c_, name_c, bias, domains):
      if hw_ != cfg.self_attention_hw:
        return
      sc = '%s/self_attention_%dx%dx%d' % (top, hw_, hw_, name_c)
      for nm, co in (('sa_f', c_ // 8), ('sa_g', c_ // 8), ('sa_h', c_)):
        _conv_p(P, g, '%s/%s' % (sc, nm), 1, c_, co, domains, bias, dtype, std, NORM_SCOPE.get(cfg.norm, ''))
      P[sc + '/sa_gamma'] = torch.zeros(1, dtype=dtype)

    tops = [('encoder_content', False, nd), ('discriminator_s', True, ()), ('discriminator_t', True, ())]
    if cfg.use_style_embedding:
      tops.append(('encoder_style', False, nd))
    for top, bias, domains in tops:
      mt = md if top.startswith('discriminator') else cfg.max_ch
      c = get_num_channels(ms, mt)
      for stage in range(ms, 0, -1):
        att(top, cfg.hw // (2 ** (ms - stage)), c, get_num_channels(stage - 1, mt), bias, domains)
        c = get_num_channels(stage - 1, mt)
    for stage in range(0, ms + 1):
      att('generator', 2 ** (stage + 2), get_num_channels(stage, cfg.max_ch), get_num_channels(stage, cfg.max_ch), False, nd)
  if cfg.use_style_embedding:    # twingan.py:47-51: style encoder (pggan.encoder) + generator norms conditioned on its output
    nd = ('s', 't')
    for sp in encoder_param_specs('encoder_style', cfg.hw, cfg.max_ch, cfg.is_growing):
      _conv_p(P, g, sp[0], sp[1], sp[2], sp[3], nd, False, dtype, std, NORM_SCOPE[cfg.norm])
    blk = 'encoder_style/before_fc_1x1x%d' % cfg.max_ch
    _conv_p(P, g, blk + '/Conv', 3, get_num_channels(0, cfg.max_ch), cfg.max_ch, nd, False, dtype, std, NORM_SCOPE[cfg.norm])
    _conv_p(P, g, blk + '/Conv_1', 4, cfg.max_ch, cfg.max_ch, nd, False, dtype, std, NORM_SCOPE[cfg.norm])
    P['encoder_style/prediction/fully_connected/weights'] = \
        torch.randn(cfg.max_ch, cfg.style_embed_size, generator=g, dtype=torch.float32).to(dtype) * \
        (math.sqrt(1.0 / cfg.max_ch) if he else std)
    P['encoder_style/prediction/fully_connected/biases'] = torch.zeros(cfg.style_embed_size, dtype=dtype)
    # generator: gamma_<d> / beta_<d> vectors become fully connected layers of the embedding (xavier uniform, zero bias)
    E = cfg.style_embed_size
    for k in [k for k in P if k.startswith('generator/') and ('/gamma_' in k or '/beta_' in k)]:
      c = P.pop(k).shape[0]
      lim = math.sqrt(6.0 / (E + c))
      P[k + '/weights'] = ((torch.rand(E, c, generator=g, dtype=torch.float32) * 2 - 1) * lim).to(dtype)
      P[k + '/biases'] = torch.zeros(c, dtype=dtype)
  if cfg.do_encoder_distillation:      # twingan.py:207-230: two heads under encoder_content, source / target norm postfix
    for head, d in (('encoder_content/encoder_distillation_source', 's'), ('encoder_content/encoder_distillation_target', 't')):
      nd1 = (d,) if cfg.norm in NORM_SCOPE else ()
      blk = '%s/before_fc_1x1x%d' % (head, cfg.max_ch)
      _conv_p(P, g, blk + '/Conv', 3, get_num_channels(0, cfg.max_ch), cfg.max_ch, nd1, cfg.norm not in NORM_SCOPE, dtype, std,
              NORM_SCOPE.get(cfg.norm, ''))
      _conv_p(P, g, blk + '/Conv_1', 4, cfg.max_ch, cfg.max_ch, nd1, cfg.norm not in NORM_SCOPE, dtype, std,
              NORM_SCOPE.get(cfg.norm, ''))
      P[head + '/prediction/fully_connected/weights'] = \
          torch.randn(cfg.max_ch, cfg.distill_embed_dim, generator=g, dtype=torch.float32).to(dtype) * \
          (math.sqrt(1.0 / cfg.max_ch) if he else std)
      P[head + '/prediction/fully_connected/biases'] = torch.zeros(cfg.distill_embed_dim, dtype=dtype)
  if cfg.res_block:      # after everything else so the other variables keep their seeded values
    ge = encoder_param_specs('encoder_content', cfg.hw, cfg.max_ch, cfg.is_growing) + \
        generator_param_specs('generator', cfg.hw, cfg.max_ch, cfg.use_unet, cfg.is_growing, cfg.unet_max_concat_hw,
                                    cfg.larger_rgb)
    dd = [sp for top in ('discriminator_s', 'discriminator_t')
          for sp in encoder_param_specs(top, cfg.hw, md, cfg.is_growing)]
    for sp in shortcut_specs(ge, cfg.hw, cfg.max_ch) + shortcut_specs(dd, cfg.hw, md):
      _conv_p(P, g, sp[0], 1, sp[2], sp[3], (), True, dtype, std)
  if he:
    for k in sorted(P):
      if k.endswith('/biases') or '/beta_' in k:
        P[k] = (torch.randn(P[k].shape, generator=g, dtype=torch.float32) * 0.1).to(dtype)
      elif '/gamma_' in k:
        P[k] = (1.0 + torch.randn(P[k].shape, generator=g, dtype=torch.float32) * 0.1).to(dtype)
      elif k.endswith('/sa_gamma'):
        P[k] = (0.5 + torch.randn(P[k].shape, generator=g, dtype=torch.float32) * 0.1).to(dtype)
  return P


def generator_var_names(P):
  """twingan.py:526-527: G step trains encoder_content, encoder_style, generator."""
  return sorted(k for k in P if k.startswith(('encoder_content/', 'encoder_style/', 'generator/')))


def discriminator_var_names(P):
  """image_generation.py:484-485: every scope starting with 'discriminator'."""
  return sorted(k for k in P if k.startswith('discriminator'))


# ------------------------------------------------------------------------------------------------
# primitives
# ------------------------------------------------------------------------------------------------
def conv2d(x, w, padding):
  """NHWC x HWIO stride-1 conv via F.conv2d (TF SAME pads (k-1)//2 low, k//2 high)."""
  kh, kw = w.shape[0], w.shape[1]
  xc = x.permute(0, 3, 1, 2)
  if padding == 'SAME':
    pl, ph = (kh - 1) // 2, kh // 2
    if pl != ph:
      xc = F.pad(xc, ((kw - 1) // 2, kw // 2, pl, ph))
      pad = 0
    else:
      pad = pl
  else:
    pad = 0
  y = F.conv2d(xc, w.permute(3, 2, 0, 1), padding=pad)
  return y.permute(0, 2, 3, 1)


def leaky_relu(x, alpha=0.2):
  return torch.maximum(alpha * x, x)             # util_misc.py:86


def pixel_norm(x, eps=1e-6):
  return x / torch.sqrt(torch.mean(x * x, dim=3, keepdim=True) + eps)   # pggan_utils.py:330-331


def instance_norm(x, gamma, beta, eps=1e-6):
  mean = x.mean(dim=(1, 2), keepdim=True)        # libs/instance_norm.py:131
  var = ((x - mean) ** 2).mean(dim=(1, 2), keepdim=True)
  inv = torch.rsqrt(var + eps) * gamma           # tf.nn.batch_normalization form, :134
  return x * inv + (beta - mean * inv)


def batch_norm_train(x, gamma, beta, eps=BN_EPS):
  """libs/batch_norm.py:430,464-470 (training mode, no renorm): biased moments over (N,H,W),
  tf.nn.batch_normalization form.  Returns (y, batch mean, batch variance)."""
  mean = x.mean(dim=(0, 1, 2), keepdim=True)
  var = ((x - mean) ** 2).mean(dim=(0, 1, 2), keepdim=True)
  inv = torch.rsqrt(var + eps) * gamma
  return x * inv + (beta - mean * inv), mean.reshape(-1), var.reshape(-1)


def batch_norm_inference(x, gamma, beta, state, key, postfix, eps=BN_EPS):
  """conditional_batch_norm(is_training=False) (libs/batch_norm.py:403-470): tf.nn.batch_normalization with the moving
  mean / variance (initial values 0 / 1 when the state does not hold them yet)."""
  c = x.shape[-1]
  mean = state.get(key + 'moving_mean' + postfix, torch.zeros(c, dtype=x.dtype))
  var = state.get(key + 'moving_variance' + postfix, torch.ones(c, dtype=x.dtype))
  inv = torch.rsqrt(var + eps) * gamma
  return x * inv + (beta - mean * inv)


RENORM_MOMENTUM = 0.99                                     # libs/batch_norm.py:62; pggan_utils.py:163 sets decay the same
RENORM_BOUNDARIES = (10000, 20000, 30000)                  # nets/pggan_utils.py:43-47
RENORM_RMAX, RENORM_RMIN, RENORM_DMAX = (1.1, 1.5, 2.0, 4.0), (0.9, 0.66, 0.5, 0.25), (0.1, 0.3, 0.5, 1.0)


def renorm_clipping(global_step):
  """get_renorm_clipping_params (nets/pggan_utils.py:207-223): tf.train.piecewise_constant of the global step."""
  i = sum(1 for b in RENORM_BOUNDARIES if global_step > b)
  return dict(rmax=RENORM_RMAX[i], rmin=RENORM_RMIN[i], dmax=RENORM_DMAX[i])


def batch_renorm_train(x, gamma, beta, state, key, postfix, clipping, eps=BN_EPS, momentum=RENORM_MOMENTUM):
  """conditional_batch_norm(renorm=True, is_training=True): _batch_norm_aux + _renorm_correction_and_moments
  (libs/batch_norm.py:329-470).  ``state`` holds renorm_mean / renorm_mean_weight / renorm_stddev /
  renorm_stddev_weight / moving_mean / moving_variance under ``key + name + postfix`` (zero / one initialised,
  libs/batch_norm.py:184-246) and is updated in place."""
  mean = x.mean(dim=(0, 1, 2))
  var = x.var(dim=(0, 1, 2), unbiased=False)
  c = x.shape[-1]

  def get(name, shape, init):
    k = key + name + postfix
    if k not in state:
      state[k] = torch.full(shape, init, dtype=x.dtype)
    return k

  k_rm, k_rmw = get('renorm_mean', (c,), 0.0), get('renorm_mean_weight', (1,), 0.0)
  k_rs, k_rsw = get('renorm_stddev', (c,), 0.0), get('renorm_stddev_weight', (1,), 0.0)
  k_mm, k_mv = get('moving_mean', (c,), 0.0), get('moving_variance', (c,), 1.0)
  with torch.no_grad():
    stddev = torch.sqrt(var + eps)
    mixed_mean = state[k_rm] + (1.0 - state[k_rmw]) * mean
    mixed_std = state[k_rs] + (1.0 - state[k_rsw]) * stddev
    r = (stddev / mixed_std).clamp(clipping['rmin'], clipping['rmax'])          # stop_gradient (:443-444)
    d = ((mean - mixed_mean) / mixed_std).clamp(-clipping['dmax'], clipping['dmax'])
    state[k_rm] = state[k_rm] * momentum + mean * (1 - momentum)
    state[k_rmw] = state[k_rmw] * momentum + (1 - momentum)
    state[k_rs] = state[k_rs] * momentum + stddev * (1 - momentum)
    state[k_rsw] = state[k_rsw] * momentum + (1 - momentum)
    new_mean = state[k_rm] / state[k_rmw]
    new_std = state[k_rs] / state[k_rsw]
    state[k_mm] = state[k_mm] * momentum + new_mean * (1 - momentum)
    state[k_mv] = state[k_mv] * momentum + (new_std * new_std - eps) * (1 - momentum)
  scale, offset = r * gamma, d * gamma + beta                                    # _compose_transforms (:431-437)
  return (x - mean) * torch.rsqrt(var + eps) * scale + offset


def moving_average_update(moving, value, decay=BN_DECAY):
  """moving_averages.assign_moving_average(zero_debias=False), libs/batch_norm.py:283-300."""
  return moving - (1.0 - decay) * (moving - value)


def upsample2x(x):
  return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)      # pggan_utils.py:349-350


def avg_pool2(x):
  n, h, w, c = x.shape
  return x.reshape(n, h // 2, 2, w // 2, 2, c).mean(dim=(2, 4))          # tf.nn.avg_pool 2x2 s2 VALID


def minibatch_state_concat(x):
  eps = 1e-8 if x.dtype in (torch.float32, torch.float64) else 1e-6      # pggan_utils.py:359
  mean = x.mean(dim=0, keepdim=True)
  std = torch.sqrt(((x - mean) ** 2).mean(dim=0, keepdim=True) + eps)
  val = std.mean()
  tile = val.reshape(1, 1, 1, 1).expand(x.shape[0], 4, 4, 1)
  return torch.cat([x, tile], dim=3)


# ------------------------------------------------------------------------------------------------
# layers = arg-scoped conv (nets/pggan_utils.py:54-127,236-245): conv -> norm -> act, then pixel-norm
# ------------------------------------------------------------------------------------------------
def equalize(x, cfg, k):
  """maybe_equalized_conv2d / maybe_equalized_fc (nets/pggan_utils.py:236-254): the INPUT is scaled by
  sqrt(2 / (in_ch * k^2)) (k = 1 for the fully connected layer)."""
  if not cfg.equalized:
    return x
  return x * math.sqrt(2.0 / (x.shape[-1] * k * k))


def resblock(P, blk, input_layer, out_channels, conv_out, cfg):
  """maybe_resblock (nets/pggan_utils.py:257-264,334-342): identity, or a 1x1 'shortcut' conv (+bias, no norm, no
  activation) when the channel count changes, added to the block's conv output."""
  if not cfg.res_block:
    return conv_out
  if input_layer.shape[-1] == out_channels:
    return input_layer + conv_out
  w = spectral_normed_weight(P, blk + '/shortcut', cfg, blk.startswith('discriminator'))
  sc = conv2d(equalize(input_layer, cfg, 1), w, 'SAME') + P[blk + '/shortcut/biases']
  return sc + conv_out


def l2_normalize(x):
  """tf.nn.l2_normalize over all elements (epsilon 1e-12 under the square root's argument)."""
  return x / x.pow(2).sum().clamp_min(1e-12).sqrt()


def spectral_normed_weight(P, scope, cfg, is_discriminator=True):
  """libs/sn.py:38-101 with num_iters=1: v = l2n(u W^T), u' = l2n(v W), sigma = v W u'^T, W_bar = W / sigma; the
  gradient flows through v, u' and sigma.  The reference assigns u at every use of W in unspecified order within a
  session.run; this restatement fixes the schedule "every use in a run reads the pre-run u" (see end_run)."""
  w = P[scope + '/weights']
  if not (cfg.spectral_norm and (is_discriminator or cfg.sn_non_disc)):      # nets/pggan_utils.py:316-320
    return w
  if cfg.sn_cache is None:
    cfg.sn_cache = {}
  if scope in cfg.sn_cache:
    return cfg.sn_cache[scope][0]
  u = cfg.sn_state[scope + '/u'].to(w.dtype)
  w2 = w.reshape(-1, w.shape[-1])
  v = l2_normalize(u @ w2.t())
  u1 = l2_normalize(v @ w2)
  sigma = (v @ w2 @ u1.t()).reshape(())
  w_bar = (w2 / sigma).reshape(w.shape)
  cfg.sn_cache[scope] = (w_bar, u1.detach())
  return w_bar


def end_run(cfg):
  """The tf.assign(u, u_final) of libs/sn.py:84-86 for every kernel used in this run."""
  if cfg.sn_cache:
    for scope, (_, u1) in cfg.sn_cache.items():
      cfg.sn_state[scope + '/u'] = u1.clone()
    cfg.sn_cache.clear()


def init_sn_state(P, seed=0, non_disc=False):
  """u ~ truncated normal [1, cout] for every discriminator conv kernel (libs/sn.py:56-57); the attention convs and
  the fully connected layer are not spectrally normed (libs/self_attention.py:33-58; libs/ops.py:37-40)."""
  g = torch.Generator().manual_seed(seed)
  st = {}
  for k in sorted(P):
    if (non_disc or k.startswith('discriminator')) and k.endswith('/weights') and P[k].dim() == 4 and \
        '/self_attention_' not in k:
      t = torch.empty(1, P[k].shape[-1], dtype=torch.float32)
      torch.nn.init.trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=g)
      st[k[:-len('/weights')] + '/u'] = t.double()
  return st


def self_attention(P, sc, layer, domain, cfg, is_discriminator, cond=None):
  """libs/self_attention.py:24-70."""
  n, hh, ww, c = layer.shape
  outs = []
  for nm in ('sa_f', 'sa_g', 'sa_h'):
    scope = '%s/%s' % (sc, nm)
    if is_discriminator:
      y = conv2d(layer, P[scope + '/weights'], 'SAME') + P[scope + '/biases']
    else:
      y = ge_conv(P, scope, layer, domain, cfg, k=1, act=False, pixnorm=False, equalized=False, cond=cond,
                  spectral=False)      # libs/self_attention.py:33-58 calls sn.convolution without do_spec_norm
    outs.append(torch.tanh(y) if nm != 'sa_h' else y)
  f, g, h = outs
  npos = hh * ww
  s_ = torch.bmm(f.reshape(n, npos, -1), g.reshape(n, npos, -1).transpose(1, 2))
  beta = torch.softmax(s_, dim=-1)
  o = torch.bmm(beta, h.reshape(n, npos, c)).reshape(layer.shape)
  return P[sc + '/sa_gamma'] * o + layer


def maybe_self_attention(P, top, hw, name_c, net, ep, domain, cfg, is_discriminator=False, cond=None):
  """nets/pggan_utils.py:301-308."""
  if cfg.do_self_attention and hw == cfg.self_attention_hw:
    name = 'self_attention_%dx%dx%d' % (hw, hw, name_c)
    net = self_attention(P, '%s/%s' % (top, name), net, domain, cfg, is_discriminator, cond)
    ep[name] = net
  return net


def ge_conv(P, scope, x, domain, cfg, k=3, padding='SAME', act=True, pixnorm=True, equalized=True, cond=None,
            spectral=True):
  """Generator/encoder conv: no bias (a normalizer is set), per-domain instance norm,
  LeakyReLU, optional pixel norm (nets/pggan.py:78-81,387-391)."""
  w = spectral_normed_weight(P, scope, cfg, False) if spectral else P[scope + '/weights']
  y = conv2d(equalize(x, cfg, k) if equalized else x, w, padding)
  if cfg.norm == 'instance_norm' and cond is not None:
    # conditional parameters (libs/instance_norm.py:93-120): gamma = 1 + FC(cond), beta = FC(cond), one row per image
    pre = scope + '/InstanceNorm/'
    gamma = 1.0 + cond @ P[pre + 'gamma%s/weights' % _pf(domain)] + P[pre + 'gamma%s/biases' % _pf(domain)]
    beta = cond @ P[pre + 'beta%s/weights' % _pf(domain)] + P[pre + 'beta%s/biases' % _pf(domain)]
    y = instance_norm(y, gamma[:, None, None, :], beta[:, None, None, :], cfg.in_eps)
  elif cfg.norm == 'instance_norm':
    y = instance_norm(y, P[scope + '/InstanceNorm/gamma' + _pf(domain)], P[scope + '/InstanceNorm/beta' + _pf(domain)],
                      cfg.in_eps)
  elif cfg.norm == 'batch_norm':       # the reference's default generator_norm_type (nets/pggan.py:24)
    if cond is not None:
      # conditional batch norm (libs/batch_norm.py:82-85,152-159,403-424): the embedding is l2-normalised per row,
      # gamma = 1 + FC, beta = FC, one row per image, broadcast as [B,1,1,C] against the BATCH statistics
      cn = cond / cond.pow(2).sum(dim=1, keepdim=True).clamp_min(1e-12).sqrt()
      pre = scope + '/BatchNorm/'
      gamma = (1.0 + cn @ P[pre + 'gamma%s/weights' % _pf(domain)] + P[pre + 'gamma%s/biases' % _pf(domain)])[:, None, None, :]
      beta = (cn @ P[pre + 'beta%s/weights' % _pf(domain)] + P[pre + 'beta%s/biases' % _pf(domain)])[:, None, None, :]
    else:
      gamma, beta = P[scope + '/BatchNorm/gamma' + _pf(domain)], P[scope + '/BatchNorm/beta' + _pf(domain)]
    if not cfg.is_training:
      y = batch_norm_inference(y, gamma, beta, cfg.bn_state or {}, scope + '/BatchNorm/', _pf(domain))
      bm = bv = None
    else:
      y, bm, bv = batch_norm_train(y, gamma, beta)
    if cfg.is_training and cfg.bn_state is not None:       # moving statistics per domain postfix (libs/batch_norm.py:184-196)
      for nm, val, init in (('moving_mean_', bm, 0.0), ('moving_variance_', bv, 1.0)):
        key = scope + '/BatchNorm/' + nm.rstrip('_') + _pf(domain)
        cur = cfg.bn_state.get(key, torch.full_like(val, init))
        cfg.bn_state[key] = moving_average_update(cur, val.detach())
  elif cfg.norm == 'batch_renorm' and not cfg.is_training:      # inference: the moving statistics, no r / d correction
    y = batch_norm_inference(y, P[scope + '/BatchNorm/gamma' + _pf(domain)], P[scope + '/BatchNorm/beta' + _pf(domain)],
                             cfg.bn_state or {}, scope + '/BatchNorm/', _pf(domain))
  elif cfg.norm == 'batch_renorm':     # the configuration of docs/training.md:17
    if cond is not None:      # conditional parameters as for batch norm: l2-normalised embedding, one row per image
      cn = cond / cond.pow(2).sum(dim=1, keepdim=True).clamp_min(1e-12).sqrt()
      pre = scope + '/BatchNorm/'
      gamma = (1.0 + cn @ P[pre + 'gamma%s/weights' % _pf(domain)] + P[pre + 'gamma%s/biases' % _pf(domain)])[:, None, None, :]
      beta = (cn @ P[pre + 'beta%s/weights' % _pf(domain)] + P[pre + 'beta%s/biases' % _pf(domain)])[:, None, None, :]
    else:
      gamma, beta = P[scope + '/BatchNorm/gamma' + _pf(domain)], P[scope + '/BatchNorm/beta' + _pf(domain)]
    y = batch_renorm_train(y, gamma, beta, cfg.bn_state, scope + '/BatchNorm/', _pf(domain),
                           renorm_clipping(cfg.global_step))
  elif cfg.norm == 'batch_renorm_native':
    # tf.contrib.layers.batch_norm(decay=0.99, renorm=True, renorm_clipping=..., scope=<postfix>) (nets/pggan_utils.py:
    # 175-188) -> tf.layers.BatchNormalization, non-fused: the arithmetic libs/batch_norm.py says it was copied from
    # (its docstring, :64-66), without the conditional layer; variables '<conv>/_s/{gamma, beta, moving_*, renorm_*}'
    assert cond is None, 'Tensorflow implementation does not support `conditional_layer`.'      # pggan_utils.py:177
    pre = '%s/%s/' % (scope, _pf(domain))
    if not domain:
      raise NotImplementedError("a native normaliser with an empty postfix opens variable_scope(''): not restated")
    if cfg.is_training:
      y = batch_renorm_train(y, P[pre + 'gamma'], P[pre + 'beta'], cfg.bn_state, pre, '', renorm_clipping(cfg.global_step))
    else:
      y = batch_norm_inference(y, P[pre + 'gamma'], P[pre + 'beta'], cfg.bn_state or {}, pre, '')
  elif cfg.norm == 'layer_norm_native':
    # tf.contrib.layers.layer_norm(center, scale, scope=<postfix>) (nets/pggan_utils.py:189-197), TF 1.8: moments over
    # axes [1, 2, 3] (begin_norm_axis=1) of each image, gamma / beta over the last axis (begin_params_axis=-1),
    # tf.nn.batch_normalization with variance_epsilon 1e-12
    assert cond is None, 'Tensorflow implementation does not support `conditional_layer`.'      # pggan_utils.py:191
    mean = y.mean(dim=(1, 2, 3), keepdim=True)
    var = ((y - mean) ** 2).mean(dim=(1, 2, 3), keepdim=True)
    y = (y - mean) * torch.rsqrt(var + LN_EPS) * P[norm_var(scope, NATIVE, 'gamma', domain)] + P[norm_var(scope, NATIVE, 'beta', domain)]
  elif cfg.norm in ('none', None):      # nets/pggan_utils.py:198-200: normalizer_fn None -> slim's conv2d adds its bias
    y = y + P[scope + '/biases']
  else:
    raise NotImplementedError(cfg.norm)
  if act:
    y = leaky_relu(y, cfg.lrelu)
  if pixnorm and cfg.do_pixel_norm:
    y = pixel_norm(y, cfg.pn_eps)
  return y


def d_conv(P, scope, x, cfg, k=3, padding='SAME'):
  """Discriminator conv: bias, no norm, LeakyReLU (nets/pggan_utils.py:116; slim bias rule)."""
  y = conv2d(equalize(x, cfg, k), spectral_normed_weight(P, scope, cfg), padding) + P[scope + '/biases']
  return leaky_relu(y, cfg.lrelu)


# ------------------------------------------------------------------------------------------------
# networks
# ------------------------------------------------------------------------------------------------
def encoder(P, x, domain, cfg, top='encoder_content'):
  """nets/pggan.py:403-479 (encoder_before_classification).  Returns (net, end_points)."""
  hw = x.shape[1]
  ms = max_stage_of(hw)
  ep = {'source': x}
  shr = None
  if cfg.is_growing:
    shr = avg_pool2(x)
    name = 'from_rgb_%dx%d' % (hw // 2, hw // 2)
    pooled = shr
    shr = ge_conv(P, '%s/%s/Conv' % (top, name), shr, domain, cfg, k=1)
    shr = resblock(P, '%s/%s' % (top, name), pooled, shr.shape[-1], shr, cfg)
    ep[name] = shr
  name = 'from_rgb_%dx%d' % (hw, hw)
  net = ge_conv(P, '%s/%s/Conv' % (top, name), x, domain, cfg, k=1)
  net = resblock(P, '%s/%s' % (top, name), x, net.shape[-1], net, cfg)
  ep[name] = net
  for stage in range(ms, 0, -1):
    nc = get_num_channels(stage - 1, cfg.max_ch)
    cur = hw // (2 ** (ms - stage))
    net = maybe_self_attention(P, top, cur, nc, net, ep, domain, cfg)
    name = 'encoder_block_%dx%dx%d' % (cur, cur, nc)
    blk_in = net
    net = ge_conv(P, '%s/%s/Conv' % (top, name), net, domain, cfg)
    net = ge_conv(P, '%s/%s/Conv_1' % (top, name), net, domain, cfg)
    net = resblock(P, '%s/%s' % (top, name), blk_in, nc, net, cfg)
    ep[name] = net
    cur //= 2
    net = avg_pool2(net)
    ep['downsample_to_%dx%dx%d' % (cur, cur, nc)] = net
    if stage == ms and cfg.is_growing:
      net = net * cfg.alpha_grow + (1 - cfg.alpha_grow) * shr
      ep['encoder_block_interpolated_%dx%dx%d' % (cur, cur, nc)] = net
  ep['before_classification'] = net
  return net, ep


def encoder_classification(P, net, domain, cfg, top):
  """nets/pggan.py:482-507: conv3x3 SAME, conv4x4 VALID (norm + LeakyReLU, no pixel norm), squeeze, fully connected."""
  blk = '%s/before_fc_1x1x%d' % (top, cfg.max_ch)
  net = ge_conv(P, blk + '/Conv', net, domain, cfg, pixnorm=False)
  net = ge_conv(P, blk + '/Conv_1', net, domain, cfg, k=4, padding='VALID', pixnorm=False)
  feat = net.reshape(net.shape[0], -1)
  return equalize(feat, cfg, 1) @ P[top + '/prediction/fully_connected/weights'] + P[top + '/prediction/fully_connected/biases']


def cosine_distance(expected, embedding, weight):
  """twingan.py:515-519: tf.losses.cosine_distance(l2n(expected), l2n(embedding), axis=-1, weights=w) =
  w * mean_b(1 - sum_c e_c p_c)."""
  def l2n(x):
    return x * torch.rsqrt(x.pow(2).sum(dim=-1, keepdim=True).clamp_min(1e-12))
  return weight * (1.0 - (l2n(expected) * l2n(embedding)).sum(dim=-1)).mean()


def encoder_full(P, x, domain, cfg, top='encoder_style'):
  """nets/pggan.py:482-541 (pggan.encoder): encoder_before_classification, conv3x3 SAME, conv4x4 VALID (norm +
  LeakyReLU, no pixel norm), squeeze, fully connected -> [B, output_dim]."""
  net, ep = encoder(P, x, domain, cfg, top)
  blk = '%s/before_fc_1x1x%d' % (top, cfg.max_ch)
  net = ge_conv(P, blk + '/Conv', net, domain, cfg, pixnorm=False)
  net = ge_conv(P, blk + '/Conv_1', net, domain, cfg, k=4, padding='VALID', pixnorm=False)
  feat = net.reshape(net.shape[0], -1)
  pred = equalize(feat, cfg, 1) @ P[top + '/prediction/fully_connected/weights'] + \
      P[top + '/prediction/fully_connected/biases']
  ep['prediction'] = pred
  return pred, ep


def _concat_unet(layer, unet_ep, max_ch, max_hw=None):
  """nets/pggan_utils.py:281-298."""
  if unet_ep is None:
    return layer
  hw = layer.shape[1]
  if max_hw and hw > max_hw:      # pggan_unet_max_concat_hw (:287-289)
    return layer
  nc = get_num_channels(max_stage_of(hw) - 1, max_ch)
  name = 'encoder_block_interpolated_%dx%dx%d' % (hw, hw, nc)
  if name not in unet_ep:
    name = 'encoder_block_%dx%dx%d' % (hw, hw, nc)
  if name not in unet_ep:
    raise ValueError('%s not in unet_end_points' % name)
  return torch.cat((layer, unet_ep[name]), dim=3)


def generator(P, source, domain, cfg, unet_ep=None, top='generator', cond=None):
  """nets/pggan.py:93-211 with a [B,4,4,C] source (TwinGAN mode).  Returns (output, end_points)."""
  ms = max_stage_of(cfg.hw)
  ep = {'source': source}
  net = source
  before_growth = None
  hw = 4
  for stage in range(0, ms + 1):
    hw = 2 ** (stage + 2)
    oc = get_num_channels(stage, cfg.max_ch)
    name = 'block_%dx%dx%d' % (hw, hw, oc)
    if hw == 4:
      if source.shape[1] == 1 and source.shape[2] == 1:      # latent noise (nets/pggan.py:135-153): pad to 7x7, 4x4 VALID
        net = F.pad(net, (0, 0, 3, 3, 3, 3))
        net = ge_conv(P, '%s/%s/Conv' % (top, name), net, domain, cfg, k=4, padding='VALID', cond=cond)
      else:
        assert source.shape[1] == 4 and source.shape[2] == 4
        net = ge_conv(P, '%s/%s/Conv' % (top, name), net, domain, cfg, cond=cond)
      net = ge_conv(P, '%s/%s/Conv_1' % (top, name), net, domain, cfg, cond=cond)
    else:
      if stage == ms and cfg.is_growing:
        rgb = 'generator_to_rgb_%dx%d' % (hw // 2, hw // 2)
        before_growth = ge_conv(P, '%s/%s/Conv' % (top, rgb), net, domain, cfg, k=rgb_kernel_size(cfg.larger_rgb, hw),
                                act=False, pixnorm=False, cond=cond)
        before_growth = upsample2x(before_growth)
        ep[rgb] = before_growth
      net = upsample2x(net)
      net = _concat_unet(net, unet_ep, cfg.max_ch, cfg.unet_max_concat_hw)
      blk_in = net
      net = ge_conv(P, '%s/%s/Conv' % (top, name), net, domain, cfg, cond=cond)
      net = ge_conv(P, '%s/%s/Conv_1' % (top, name), net, domain, cfg, cond=cond)
      net = resblock(P, '%s/%s' % (top, name), blk_in, oc, net, cfg)
    ep[name] = net
    net = maybe_self_attention(P, top, hw, oc, net, ep, domain, cfg, cond=cond)      # nets/pggan.py:188-190
  rgb = 'generator_to_rgb_%dx%d' % (hw, hw)
  to_rgb = ge_conv(P, '%s/%s/Conv' % (top, rgb), net, domain, cfg, k=rgb_kernel_size(cfg.larger_rgb, hw), act=False,
                   pixnorm=False, cond=cond)
  if cfg.is_growing:
    out = to_rgb * cfg.alpha_grow + (1 - cfg.alpha_grow) * before_growth
  else:
    out = to_rgb
  ep['output'] = out
  return out, ep


def gdrop(layer, strength, noise):
  """libs/gdrop.py:20-36, mode 'prop', NHWC: layer * (noise[N,1,1,C] * strength * sqrt(C) + 1)."""
  n, c = layer.shape[0], layer.shape[-1]
  coef = strength * float(np.sqrt(np.float32(c)))      # np.sqrt(np.float32(int(layer.shape[-1]))): a float32 square root
  return layer * (noise.reshape(n, 1, 1, c).to(layer.dtype) * coef + 1.0)


def maybe_gdrop(layer, cfg):
  """nets/pggan.py:351-355."""
  if cfg.do_dgrop and cfg.is_training and cfg.gdrop_strength:
    if cfg.gdrop_noise:
      noise = cfg.gdrop_noise.pop(0)
    else:
      noise = torch.randn(layer.shape[0], layer.shape[-1], dtype=layer.dtype)
    return gdrop(layer, cfg.gdrop_strength, noise)
  return layer


def discriminator(P, x, cfg, top):
  """nets/pggan.py:242-376.  Returns (prediction [B,1], end_points)."""
  hw = x.shape[1]
  ms = max_stage_of(hw)
  ep = {}
  shr = None
  if cfg.is_growing:
    shr = avg_pool2(x)
    name = 'from_rgb_%dx%d' % (hw // 2, hw // 2)
    pooled = shr
    shr = d_conv(P, '%s/%s/Conv' % (top, name), shr, cfg, k=1)
    shr = resblock(P, '%s/%s' % (top, name), pooled, shr.shape[-1], shr, cfg)
    ep[name] = shr
  name = 'from_rgb_%dx%d' % (hw, hw)
  net = d_conv(P, '%s/%s/Conv' % (top, name), x, cfg, k=1)
  net = resblock(P, '%s/%s' % (top, name), x, net.shape[-1], net, cfg)
  ep[name] = net
  for stage in range(ms, 0, -1):
    nc = get_num_channels(stage - 1, cfg.max_ch_dis or cfg.max_ch)
    cur = hw // (2 ** (ms - stage))
    net = maybe_self_attention(P, top, cur, nc, net, ep, None, cfg, True)   # nets/pggan.py:294-296
    name = 'encoder_block_%dx%dx%d' % (cur, cur, nc)
    blk_in = net
    net = d_conv(P, '%s/%s/Conv' % (top, name), maybe_gdrop(net, cfg), cfg)      # nets/pggan.py:221-231
    net = d_conv(P, '%s/%s/Conv_1' % (top, name), maybe_gdrop(net, cfg), cfg)
    net = resblock(P, '%s/%s' % (top, name), blk_in, nc, net, cfg)
    ep[name] = net
    net = avg_pool2(net)
    if stage == ms and cfg.is_growing:
      net = net * cfg.alpha_grow + (1 - cfg.alpha_grow) * shr
  blk = '%s/before_fc_1x1x%d' % (top, cfg.max_ch_dis or cfg.max_ch)
  net = minibatch_state_concat(net)
  net = d_conv(P, blk + '/Conv', maybe_gdrop(net, cfg), cfg, k=3, padding='SAME')      # nets/pggan.py:328-331
  net = d_conv(P, blk + '/Conv_1', maybe_gdrop(net, cfg), cfg, k=4, padding='VALID')
  ep['before_fc'] = net
  feat = net.reshape(net.shape[0], -1)
  pred = equalize(feat, cfg, 1) @ P[top + '/prediction/fully_connected/weights'] + \
      P[top + '/prediction/fully_connected/biases']
  ep['prediction'] = pred
  return pred, ep


# ------------------------------------------------------------------------------------------------
# the plain PGGAN trainer (image_generation.GanModel, image_generation.py:194-476) -- BASELINE configs[0]
# ------------------------------------------------------------------------------------------------
def init_pggan_params(cfg, seed=0, dtype=torch.float32, std=0.02):
  """Scopes 'generator' (latent-noise input: block_4x4 Conv is 4x4 VALID over get_num_channels(1) channels) and
  'discriminator'; normaliser variables without a domain postfix."""
  g = torch.Generator().manual_seed(seed)
  P = {}
  he = std == 'he'
  ns = NORM_SCOPE.get(cfg.norm, '')
  nd = ('',) if cfg.norm in NORM_SCOPE else ()
  specs = generator_param_specs('generator', cfg.hw, cfg.max_ch, False, cfg.is_growing, None, cfg.larger_rgb)
  c0 = get_num_channels(0, cfg.max_ch)
  for sp in specs:
    if sp[0] == 'generator/block_4x4x%d/Conv' % c0:
      sp = (sp[0], 4, get_num_channels(1, cfg.max_ch), c0)
    _conv_p(P, g, sp[0], sp[1], sp[2], sp[3], nd, False, dtype, std, ns)
  md = cfg.max_ch_dis or cfg.max_ch
  top = 'discriminator'
  for sp in encoder_param_specs(top, cfg.hw, md, cfg.is_growing) + discriminator_tail_specs(top, md):
    _conv_p(P, g, sp[0], sp[1], sp[2], sp[3], (), True, dtype, std)
  P[top + '/prediction/fully_connected/weights'] = \
      torch.randn(md, 1, generator=g, dtype=torch.float32).to(dtype) * (math.sqrt(1.0 / md) if he else std)
  P[top + '/prediction/fully_connected/biases'] = torch.zeros(1, dtype=dtype)
  if he:
    for k in sorted(P):
      if k.endswith('/biases') or k.endswith('/beta'):
        P[k] = (torch.randn(P[k].shape, generator=g, dtype=torch.float32) * 0.1).to(dtype)
      elif k.endswith('/gamma'):
        P[k] = (1.0 + torch.randn(P[k].shape, generator=g, dtype=torch.float32) * 0.1).to(dtype)
  return P


def pggan_generator_loss(P, targets, cfg, noise):
  """GENERATOR_LOSSES of image_generation.GanModel (add_gan_loss, :331-344).  Returns (total, terms)."""
  fake, _ = generator(P, noise, '', cfg, None, 'generator')
  pred, _ = discriminator(P, fake, cfg, 'discriminator')
  terms = {'generator_fool_loss': _fool_loss(pred, cfg)}
  return sum(terms.values()), terms


def pggan_discriminator_loss(P, targets, cfg, noise, gp_alpha, dragan_noise=None):
  """DISCRIMINATOR_LOSSES of image_generation.GanModel (:348-476); the generator runs without a tape."""
  with torch.no_grad():
    if cfg.is_growing:
      targets = growing_image(targets, cfg.alpha_grow)
    fake, _ = generator(P, noise, '', cfg, None, 'generator')
  pr, _ = discriminator(P, targets, cfg, 'discriminator')
  pf, _ = discriminator(P, fake, cfg, 'discriminator')
  terms = {}
  _real_fake_losses(terms, '', pf, pr, cfg)
  if cfg.drift and cfg.loss in ('wgan_gp', 'wgan'):
    terms['discriminator_drift_loss'] = cfg.drift * (pr ** 2).mean()
  if cfg.loss in ('wgan_gp', 'dragan'):
    if cfg.loss == 'dragan':
      interp = targets + gp_alpha * (targets + 0.5 * targets.var(unbiased=False) * dragan_noise - targets)
    else:
      interp = targets + gp_alpha * (fake - targets)
    interp = interp.detach().requires_grad_(True)
    pi, _ = discriminator(P, interp, cfg, 'discriminator')
    gi, = torch.autograd.grad(pi.sum(), interp, create_graph=True)
    slopes = torch.sqrt((gi ** 2).sum(dim=(1, 2, 3)))
    terms['discriminator_gradient_penalty'] = cfg.gp_lambda * ((slopes - 1.0) ** 2).mean()
  return sum(terms.values()), terms


def growing_image(img, alpha):
  """image_generation.py:1001-1006."""
  return alpha * img + (1 - alpha) * upsample2x(avg_pool2(img))


# ------------------------------------------------------------------------------------------------
# the per-clone graph and its losses (twingan.py:146-521)
# ------------------------------------------------------------------------------------------------
def forward_generators(P, sources, targets, cfg):
  """twingan.py:198-288: E(s), E(t), the four generator passes, and the two re-encodes."""
  es, es_ep = encoder(P, sources, 's', cfg)
  et, et_ep = encoder(P, targets, 't', cfg)
  unet = cfg.use_unet
  rand = style_s = style_t = None
  if cfg.use_style_embedding:      # twingan.py:201-235
    style_s, _ = encoder_full(P, sources, 's', cfg)
    style_t, _ = encoder_full(P, targets, 't', cfg)
    rand = cfg.style_noise if cfg.style_noise is not None else torch.randn_like(style_s)
  s_prime, _ = generator(P, et, 's', cfg, et_ep if unet else None, cond=rand)      # target content -> source domain
  s_cycle, _ = generator(P, es, 's', cfg, es_ep if unet else None, cond=style_s)
  t_prime, _ = generator(P, es, 't', cfg, es_ep if unet else None, cond=rand)
  t_cycle, _ = generator(P, et, 't', cfg, et_ep if unet else None, cond=style_t)
  return dict(es=es, et=et, s_prime=s_prime, s_cycle=s_cycle, t_prime=t_prime, t_cycle=t_cycle, random_style_embed=rand)


LOSSES = ('wgan_gp', 'wgan', 'hinge', 'gan', 'dragan')


def translate(P, images, cfg, to='t', style=None):
  """The inference branch of twingan.GanModel._clone_fn (twingan.py:300-363): `custom_generated_<to>_style_*` =
  G_<to>(E_<from>(images)) with is_training=False, UNet skips from the same encoder pass.  ``style``: the [B, E]
  conditional embedding (None without --use_style_embedding)."""
  import dataclasses
  ci = dataclasses.replace(cfg, is_training=False)
  frm = 's' if to == 't' else 't'
  net, ep = encoder(P, images, frm, ci)
  out, _ = generator(P, net, to, ci, ep if ci.use_unet else None, cond=style)
  return out


def _fool_loss(pred, cfg):
  """image_generation.py:331-344: -mean D(fake) for wgan / wgan_gp / hinge, sigmoid-xent against ones otherwise."""
  if cfg.loss in ('wgan_gp', 'wgan', 'hinge'):
    return -pred.mean() * cfg.gan_weight
  return F.binary_cross_entropy_with_logits(pred, torch.ones_like(pred)) * cfg.gan_weight


def _real_fake_losses(terms, name, pf, pr, cfg):
  """The real/fake part of the discriminator loss (image_generation.py:348-357,370-394)."""
  if cfg.loss in ('wgan_gp', 'wgan'):
    terms['discriminator_loss' + name] = (pf.mean() - pr.mean()) * cfg.gan_weight
  elif cfg.loss == 'hinge':
    terms['discriminator_loss' + name] = (F.relu(1 + pf).mean() + F.relu(1 - pr).mean()) * cfg.gan_weight
  else:
    terms['discriminator_fake_loss' + name] = F.binary_cross_entropy_with_logits(pf, torch.zeros_like(pf)) * cfg.gan_weight
    terms['discriminator_real_loss' + name] = F.binary_cross_entropy_with_logits(pr, torch.ones_like(pr)) * cfg.gan_weight


def generator_loss(P, sources, targets, cfg):
  """Sum of GENERATOR_LOSSES (twingan.py:464-505; image_generation.py:331-344)."""
  assert cfg.loss in LOSSES
  if cfg.is_growing:
    sources, targets = growing_image(sources, cfg.alpha_grow), growing_image(targets, cfg.alpha_grow)
  o = forward_generators(P, sources, targets, cfg)
  e_tp, _ = encoder(P, o['t_prime'], 't', cfg)
  e_sp, _ = encoder(P, o['s_prime'], 's', cfg)
  terms = {}
  for d, orig, prime, cyc, enc_orig, enc_opp_prime in (
      ('s', sources, o['s_prime'], o['s_cycle'], o['es'], e_tp),
      ('t', targets, o['t_prime'], o['t_cycle'], o['et'], e_sp)):
    top = 'discriminator_' + d
    terms['l_cyc_' + d] = (orig - cyc).abs().mean() * cfg.l_cyc
    if cfg.hw >= 64 and cfg.do_l_cyc_gan:
      pc, _ = discriminator(P, cyc, cfg, top)
      terms['generator_fool_loss_cycle_' + d] = _fool_loss(pc, cfg)
    pp, _ = discriminator(P, prime, cfg, top)
    terms['generator_fool_loss_prime_' + d] = _fool_loss(pp, cfg)
    if cfg.l_content:
      terms['l_content_' + d] = (enc_orig - enc_opp_prime).abs().mean() * cfg.l_content
      if cfg.use_style_embedding:      # twingan.py:495-505: |random_style_embed - E_style(d_prime)|
        st_prime, _ = encoder_full(P, prime, d, cfg)
        terms['l_style_' + d] = (o['random_style_embed'] - st_prime).abs().mean() * cfg.l_content
  if cfg.do_encoder_distillation:
    # twingan.py:207-230,290-298: the graph applies both heads to the original and to the re-encoded content whether or
    # not a dataset carries embeddings (BatchNorm statistics move accordingly); source head first on E(s), then E(s')
    hs, ht = 'encoder_content/encoder_distillation_source', 'encoder_content/encoder_distillation_target'
    d_s = encoder_classification(P, o['es'], 's', cfg, hs)
    d_t = encoder_classification(P, o['et'], 't', cfg, ht)
    d_sp = encoder_classification(P, e_sp, 's', cfg, hs)
    d_tp = encoder_classification(P, e_tp, 't', cfg, ht)
    if cfg.hw >= cfg.distillation_start_hw:      # twingan.py:507-521
      if cfg.distill_embed_s is not None:      # dataset a: the source image, and t' (generated from its content)
        terms['l_source_distillation'] = cosine_distance(cfg.distill_embed_s, d_s, cfg.distillation_weight)
        terms['l_t_prime_distillation'] = cosine_distance(cfg.distill_embed_s, d_tp, cfg.distillation_weight)
      if cfg.distill_embed_t is not None:
        terms['l_target_distillation'] = cosine_distance(cfg.distill_embed_t, d_t, cfg.distillation_weight)
        terms['l_s_prime_distillation'] = cosine_distance(cfg.distill_embed_t, d_sp, cfg.distillation_weight)
  return sum(terms.values()), terms


def discriminator_loss(P, sources, targets, cfg, gp_alpha_s, gp_alpha_t, dragan_noise_s=None, dragan_noise_t=None):
  """Sum of DISCRIMINATOR_LOSSES (image_generation.py:348-412,414-476).  ``gp_alpha_*`` are the
  per-sample U[0,1) draws, shape [B,1,1,1]; ``dragan_noise_*`` the U(-1,1) draws of get_perturbed_batch
  (image shape).  E/G run without grad: only D variables are in the var_list (image_generation.py:605-610)."""
  assert cfg.loss in LOSSES
  if cfg.is_growing:
    sources, targets = growing_image(sources, cfg.alpha_grow), growing_image(targets, cfg.alpha_grow)
  with torch.no_grad():
    o = forward_generators(P, sources, targets, cfg)
  terms = {}
  for d, real, prime, cyc, a, noise in (('s', sources, o['s_prime'], o['s_cycle'], gp_alpha_s, dragan_noise_s),
                                        ('t', targets, o['t_prime'], o['t_cycle'], gp_alpha_t, dragan_noise_t)):
    top = 'discriminator_' + d
    pr, _ = discriminator(P, real, cfg, top)
    if cfg.hw >= 64 and cfg.do_l_cyc_gan:                 # only_real_fake_loss=True (twingan.py:466-474)
      pc, _ = discriminator(P, cyc, cfg, top)
      _real_fake_losses(terms, '_cycle_' + d, pc, pr, cfg)
    pp, _ = discriminator(P, prime, cfg, top)
    _real_fake_losses(terms, '_prime_' + d, pp, pr, cfg)
    if cfg.drift and cfg.loss in ('wgan_gp', 'wgan'):     # image_generation.py:360-367
      terms['discriminator_drift_loss_prime_' + d] = cfg.drift * (pr ** 2).mean()
    interp = None
    if cfg.loss == 'wgan_gp':                             # image_generation.py:414-439
      interp = (real + a * (prime - real)).detach().requires_grad_(True)
    elif cfg.loss == 'dragan':                            # image_generation.py:441-476
      perturbed = real + 0.5 * real.var(unbiased=False) * noise
      interp = (real + a * (perturbed - real)).detach().requires_grad_(True)
    if interp is not None:
      pi, _ = discriminator(P, interp, cfg, top)
      gi, = torch.autograd.grad(pi.sum(), interp, create_graph=True)     # tf.gradients(pred, interp)
      slopes = torch.sqrt((gi ** 2).sum(dim=(1, 2, 3)))
      terms['discriminator_gradient_penalty_prime_' + d] = ((slopes - 1.0) ** 2).mean() * cfg.gp_lambda
  return sum(terms.values()), terms


# ------------------------------------------------------------------------------------------------
# optimisation (image_generation.py:587-662; TF Adam)
# ------------------------------------------------------------------------------------------------
class AdamState:
  """One tf.train.AdamOptimizer shared by G and D (image_generation.py:554-561): a single pair of
  beta-power accumulators that advances on every apply."""

  def __init__(self, P, cfg):
    self.m = {k: torch.zeros_like(v) for k, v in P.items()}
    self.v = {k: torch.zeros_like(v) for k, v in P.items()}
    self.t = 0
    self.cfg = cfg

  def apply(self, P, grads, group='g'):
    c = self.cfg
    # --use_ttur builds a second optimizer (image_generation.py:554-561) that only COMPUTES the discriminator gradients
    # (:603-608); both gradient sets are applied by the generator's optimizer (:640-646): one rate, one pair of powers
    self.t += 1
    t, lr = self.t, c.lr
    lr_t = lr * math.sqrt(1.0 - c.beta2 ** t) / (1.0 - c.beta1 ** t)
    with torch.no_grad():
      for k, g in grads.items():
        self.m[k].mul_(c.beta1).add_(g, alpha=1 - c.beta1)
        self.v[k].mul_(c.beta2).addcmul_(g, g, value=1 - c.beta2)
        P[k].sub_(lr_t * self.m[k] / (self.v[k].sqrt() + c.adam_eps))


def grads_of(loss, P, names):
  ps = [P[k] for k in names]
  gs = torch.autograd.grad(loss, ps, allow_unused=True)
  return {k: (g if g is not None else torch.zeros_like(P[k])) for k, g in zip(names, gs)}


def train_step(P, opt, sources, targets, cfg, gp_alpha_s, gp_alpha_t, counter, literal_schedule=False):
  """One ``session.run(train_op)`` (image_generation.py:640-652): counter % n_critic == 0 -> apply G
  grads, else D grads.  ``literal_schedule`` also computes the un-applied gradient set, as the
  reference graph does (image_generation.py:631-639)."""
  for v in P.values():
    v.requires_grad_(True)
  g_names, d_names = generator_var_names(P), discriminator_var_names(P)
  is_g = (counter % 2 == 0)
  out = {}
  if is_g or literal_schedule:
    gl, _ = generator_loss(P, sources, targets, cfg)
    gg = grads_of(gl, P, g_names)
    out['g_loss'] = float(gl.detach())
  if (not is_g) or literal_schedule:
    dl, _ = discriminator_loss(P, sources, targets, cfg, gp_alpha_s, gp_alpha_t)
    dg = grads_of(dl, P, d_names)
    out['d_loss'] = float(dl.detach())
  for v in P.values():
    v.requires_grad_(False)
  opt.apply(P, gg if is_g else dg, 'g' if is_g else 'd')
  return out
