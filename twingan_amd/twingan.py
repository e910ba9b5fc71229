"""TwinGAN training graph, losses and the alternating G/D step on MI355X.

Mirrors twingan.GanModel._clone_fn / add_loss (twingan.py:146-521), image_generation.add_gan_loss /
_add_wgan_gp_loss / _add_optimization (image_generation.py:318-439,587-662), the Adam of
model/model_inheritor.py:537-542 and the data-parallel clone reduction of
deployment/model_deploy.py:242-315,473-503 -- with two deliberate scheduling differences:

  * only the gradient set that is applied is computed (the reference graph computes both every
    session.run -- its own note at image_generation.py:631-639);
  * one process per GPU; the per-variable tf.add_n over clones becomes ONE sum all-reduce (RCCL over
    xGMI) of the group's flat gradient buffer, with the loss pre-divided by the world size
    (model_deploy.py:265-268).
"""
import ctypes
import os

import dataclasses

import torch

from . import ops, pggan
from ._lib import call
from .dp import GradReducer, loss_scale_for_clones
from .ops import PackCache
from .params import ParamStore, declare_twingan


class _DomainStreams:
  """The two discriminators (discriminator_s / discriminator_t) are independent networks: their passes --
  forward, backward and the WGAN-GP double backward -- are enqueued on two HIP streams so the many small
  (launch- and latency-bound) kernels of the 4x4..32x32 stages of one overlap with the other.  Autograd runs
  every backward node on the stream its forward ran on, so the backward overlaps the same way; both forks and
  the join are stream-wait edges, which hipGraph capture records as graph dependencies."""
  _pool = {}

  NSTREAMS = 2      # measured: 2 streams 676 img/s, 4 streams (GP passes on their own) 615-623

  def __init__(self, device, enabled):
    self.enabled = enabled and torch.cuda.is_available() and os.environ.get('TG_DOMAIN_STREAMS', '1') != '0'
    if self.enabled:
      key = torch.device(device).index
      if key not in self._pool:
        self._pool[key] = tuple(torch.cuda.Stream(device=device) for _ in range(self.NSTREAMS))
      self.side = self._pool[key]
      self.main = torch.cuda.current_stream(device)
      for st in self.side:
        st.wait_stream(self.main)

  def domain(self, i):
    import contextlib
    return torch.cuda.stream(self.side[i]) if self.enabled else contextlib.nullcontext()

  def join(self):
    if self.enabled:
      for st in self.side:
        self.main.wait_stream(st)

  @classmethod
  def join_all(cls, device):
    """The current stream waits for the domain streams.  Needed after ``backward()``: parameter gradients are
    accumulated into the flat buffers by the backward kernels themselves (ops.GradSink), on the stream of
    the forward op, and autograd only joins streams of gradients it accumulates itself."""
    key = torch.device(device).index
    if key in cls._pool:
      main = torch.cuda.current_stream(device)
      for st in cls._pool[key]:
        main.wait_stream(st)


LOSSES = ('wgan_gp', 'wgan', 'hinge', 'gan', 'dragan')      # --loss_architecture, image_generation.py:81-83


def select_network(generator_network='pggan'):
  """GanModel._select_network (twingan.py:122-140): the network-function registry of the hot path -- the same seven
  entries, bound to the MI355X implementations (signature (P, tensor, domain, cfg, ...) -> (output, end_points))."""
  if generator_network != 'pggan':
    raise NotImplementedError('Generator network %s is not implemented.' % generator_network)
  return {'generator_network_fn': pggan.generator,
          'discriminator_network_fn': pggan.discriminator,
          'encoder_network_fn': pggan.encoder_before_classification,
          'encoder_style_network_fn': pggan.encoder,
          'encoder_classification_fn': pggan.encoder_classification,
          # "Intentionally encoder_distillation_fn is the same as classification" (twingan.py:130-131)
          'encoder_distillation_fn': pggan.encoder_classification,
          'get_noise_shape': pggan.get_noise_shape}


def _fool_loss(pred, cfg):
  """image_generation.py:331-344."""
  pred = pred.contiguous()
  if cfg.loss_architecture in ('wgan_gp', 'wgan', 'hinge'):
    return ops.mean(pred, -cfg.gan_weight)
  return ops.sigmoid_xent_mean(pred, 1.0, cfg.gan_weight)


def _fool_jobs(group, term, cfg):
  """_fool_loss as a job of ops.pred_losses (group, term, mode, a, b, coef)."""
  if cfg.loss_architecture in ('wgan_gp', 'wgan', 'hinge'):
    return [(group, term, 0, 0.0, 0.0, -cfg.gan_weight)]
  return [(group, term, 2, 1.0, 0.0, cfg.gan_weight)]


def _real_fake_losses(terms, name, pf, pr, cfg, mean_real=None):
  """image_generation.py:348-357 (wgan), :370-379 (hinge), :380-394 (gan / dragan)."""
  la = cfg.loss_architecture
  if la in ('wgan_gp', 'wgan'):
    mr = mean_real if mean_real is not None else ops.mean(pr, cfg.gan_weight)
    terms['discriminator_loss' + name] = ops.mean(pf, cfg.gan_weight) - mr
  elif la == 'hinge':
    terms['discriminator_loss' + name] = ops.hinge_mean(pf, 1.0, 1.0, cfg.gan_weight) + \
        ops.hinge_mean(pr, 1.0, -1.0, cfg.gan_weight)
  else:
    terms['discriminator_fake_loss' + name] = ops.sigmoid_xent_mean(pf, 0.0, cfg.gan_weight)
    terms['discriminator_real_loss' + name] = ops.sigmoid_xent_mean(pr, 1.0, cfg.gan_weight)


def _real_fake_jobs(names, jobs, name, gf, gr, cfg):
  """_real_fake_losses as jobs of ops.pred_losses over the groups gf (fake) / gr (real) of one batched prediction; appends
  the term names to ``names`` (term index = position)."""
  la, w = cfg.loss_architecture, cfg.gan_weight
  if la in ('wgan_gp', 'wgan'):
    t = len(names)
    names.append('discriminator_loss' + name)
    jobs += [(gf, t, 0, 0.0, 0.0, w), (gr, t, 0, 0.0, 0.0, -w)]
  elif la == 'hinge':
    t = len(names)
    names.append('discriminator_loss' + name)
    jobs += [(gf, t, 1, 1.0, 1.0, w), (gr, t, 1, 1.0, -1.0, w)]
  else:
    t = len(names)
    names += ['discriminator_fake_loss' + name, 'discriminator_real_loss' + name]
    jobs += [(gf, t, 2, 0.0, 0.0, w), (gr, t + 1, 2, 1.0, 0.0, w)]


def _sum_terms(terms):
  """tf.add_n over the loss collection (model/model_inheritor.py): ONE launch over the scalar terms (ops.sum_scalars;
  backward: every term receives the incoming gradient itself)."""
  return ops.sum_scalars(v.reshape(1) for v in terms.values())


def act_dtype(cfg):
  return {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}[cfg.precision]


def get_growing_image(img, alpha):
  """image_generation.py:1001-1006: alpha * img + (1 - alpha) * up(avgpool(img))."""
  low = ops.upsample2x_concat(ops.avg_pool2(img), None)
  return ops.lerp(img, low, alpha)


def forward_generators(P, sources, targets, cfg, style_noise=None):
  """twingan.py:198-269: E(s), E(t) and the four generator passes (shared conv weights, per-domain norm
  parameters, UNet skips from the encoder whose content is decoded).

  The reference builds six separate towers; here passes that share conv weights run as ONE batch along N --
  E([s; t]) with domains (s, t), and G([E(s); E(t); E(s); E(t)]) with domains (s, s, t, t) = s_cyc, s', t', t_cyc --
  because the low-resolution layers cannot fill 256 CUs with 16 images (instance-norm statistics are per image
  and the norm kernels pick gamma/beta per image, so the results are those of the separate passes).  The batch order
  makes every later consumer a ROW RANGE of the generator's output (ops.row_views: no copies): discriminator_s reads
  [s_cyc; s'], the re-encoding pass [s'; t'], discriminator_t [t'; t_cyc]."""
  b = sources.shape[0]
  x = ops.cat_rows([sources, targets])
  # cuts (segmented backward of a data-parallel generator step, no-ops otherwise): every tensor this encoder pass
  # hands to the generator / the losses becomes a leaf; its low-resolution half resumes in segment 1, the rest in 2
  e, ep = pggan.encoder_before_classification(P, x, ('s', 't', b, 2), cfg, cuts=(1, 2))
  e = ops.Cuts.cut(e, 1)
  # Batch order of the generator pass.  With a stateless normaliser (instance / layer norm) the four passes commute and run
  # as [s_cyc; s'; t'; t_cyc], which makes the re-encoding batch [s'; t'] a row range too.  Batch norm / renorm update their
  # moving statistics pass by pass, so there the reference's tower order stays ([s'; s_cyc; t'; t_cyc], twingan.py:233-269)
  # and [s'; t'] is the one copy left.
  cyc_first = 'batch' not in cfg.generator_norm_type
  # E(s), E(t) for the content losses (views) and the generator's batch of contents (one launch; its gradient and the two
  # halves' meet in ONE launch of the backward, ops.RowsFn)
  s_rows, t_rows = (0, b), (b, 2 * b)
  es, et, content = ops.rows(e, [s_rows, t_rows, (s_rows, t_rows, s_rows, t_rows) if cyc_first else (t_rows, s_rows, s_rows, t_rows)])
  cond = rand = None
  if cfg.use_style_embedding:
    # twingan.py:201-267: style encoder on s / t (one batch, domains s|t); s' and t' are generated with ONE random
    # N(0,1) embedding, the cycle images with the encoded style of their own input
    st, _ = pggan.encoder(P, x, ('s', 't', b, 2), cfg, 'encoder_style')
    style_s, style_t = ops.rows(st, [s_rows, t_rows])
    rand = style_noise if style_noise is not None else torch.randn(b, cfg.style_embed_size, dtype=torch.float32,
                                                                  device=x.device)
    cond = ops.cat_rows([style_s, rand, rand, style_t] if cyc_first else [rand, style_s, rand, style_t])
  # UNet skips: generator group k reads encoder group (s, t, s, t)[k] / (t, s, s, t)[k] of the [s; t] encoder batch -- no copies
  out, _ = pggan.generator(P, content, ('s', 't', 2 * b, 4), cfg, ep if cfg.use_unet else None,
                           unet_groups=(b, (0, 1, 0, 1) if cyc_first else (1, 0, 0, 1)), cond=cond)
  out = out.contiguous()
  rows = [(0, b), (b, 2 * b), (2 * b, 3 * b), (3 * b, 4 * b), (0, 2 * b), (2 * b, 4 * b)] + ([(b, 3 * b)] if cyc_first else [])
  v = ops.row_views(out, rows)
  s_cycle, s_prime = (v[0], v[1]) if cyc_first else (v[1], v[0])
  t_prime, t_cycle = v[2], v[3]
  primes = v[6] if cyc_first else ops.cat_rows([s_prime, t_prime])
  # both_s / both_t: (the cycle and the prime image of a domain as one tensor, "the cycle image comes first")
  return dict(es=es, et=et, s_prime=s_prime, s_cycle=s_cycle, t_prime=t_prime, t_cycle=t_cycle, random_style_embed=rand,
              both_s=(v[4], cyc_first), primes=primes, both_t=(v[5], False))


def translate(P, images, cfg, to='t', style=None):
  """The inference branch of GanModel._clone_fn (twingan.py:300-363), i.e. what inference/image_translation_infer.py
  fetches: `custom_generated_<to>_style_<...>` = G_<to>(E_<from>(images)) with is_training=False (BatchNorm on its
  moving statistics), the UNet skips from the same encoder pass.  ``style``: [B, E] conditional embedding
  (--use_style_embedding: the encoded style of the input, a random one, or a supplied one), else None.
  The serving signature's default (twingan.py:777-805) is sources -> custom_generated_t_style_source."""
  ci = dataclasses.replace(cfg, is_training=False)
  frm = 's' if to == 't' else 't'
  with torch.no_grad():
    net, ep = pggan.encoder_before_classification(P, images, frm, ci)
    out, _ = pggan.generator(P, net, to, ci, ep if ci.use_unet else None, cond=style)
  return out


def encode_style(P, images, cfg, domain):
  """encoder_style_network_fn(images_ph, is_training=False) -> [B, style_embed_size] (twingan.py:330-337)."""
  ci = dataclasses.replace(cfg, is_training=False)
  with torch.no_grad():
    return pggan.encoder(P, images, domain, ci, 'encoder_style')[0]


def generator_loss(P, sources, targets, cfg, style_noise=None, distill_embed_s=None, distill_embed_t=None):
  """GENERATOR_LOSSES (twingan.py:464-521; image_generation.py:331-337).  Returns (total [1], terms).
  distill_embed_*: the datasets' fp32 [B, D] embeddings of --do_encoder_distillation (None: that dataset has none)."""
  assert cfg.loss_architecture in LOSSES, cfg.loss_architecture
  pggan.prepare_run(P, cfg)
  if cfg.is_growing:
    sources, targets = get_growing_image(sources, cfg.alpha_grow), get_growing_image(targets, cfg.alpha_grow)
  b = sources.shape[0]
  o = forward_generators(P, sources, targets, cfg, style_noise)
  cyc_gan = cfg.hw >= 64 and cfg.do_l_cyc_gan
  terms = {}
  # fork: the two discriminators run on their own streams while the main stream re-encodes s' / t'
  streams = _DomainStreams(sources.device, cfg.domain_streams)
  # D(cyc) and D(prime) of one domain share weights: one batch, two minibatch-stddev groups -- rows of the generator's batch
  # (``cyc_first``: which chunk of the prediction is the cycle image's)
  reenc = {}

  def reencode():
    # re-encode s' in domain s and t' in domain t as one batch (twingan.py:275-288): rows [s'; t'] of the generator's batch
    reenc['e2'] = pggan.encoder_before_classification(P, o['primes'], ('s', 't', b, 2), cfg)[0]
  if pggan.discriminator_pair_supported(P, cfg, cfg.hw):
    # both discriminators: heads per domain on two streams (the main stream re-encodes meanwhile), everything from
    # pggan.PAIR_HW down as grouped launches on the main stream
    doms = (('s', sources, o['s_prime'], o['s_cycle'], o['both_s']), ('t', targets, o['t_prime'], o['t_cycle'], o['both_t']))
    for i, (d, orig, prime, cyc, _) in enumerate(doms):
      with streams.domain(i):
        terms['l_cyc_' + d] = ops.abs_diff_mean(orig, cyc, cfg.l_cyc_weight)
    if cyc_gan:
      preds = pggan.discriminator_pair(P, doms[0][4][0], doms[1][4][0], cfg, groups=2, streams=streams, meanwhile=reencode)
    else:
      preds = pggan.discriminator_pair(P, doms[0][2], doms[1][2], cfg, streams=streams, meanwhile=reencode)
    for (d, _, _, _, (_, cyc_first)), pred in zip(doms, preds):
      if cyc_gan:
        gc, gp = (0, 1) if cyc_first else (1, 0)
        tc, tp = ops.pred_losses(pred, pred.shape[0] // 2, _fool_jobs(gc, 0, cfg) + _fool_jobs(gp, 1, cfg), 2)
        terms['generator_fool_loss_cycle_' + d] = tc
        terms['generator_fool_loss_prime_' + d] = tp
      else:
        terms['generator_fool_loss_prime_' + d] = _fool_loss(pred, cfg)
    doms = ()
  else:
    doms = (('s', sources, o['s_prime'], o['s_cycle'], o['both_s']), ('t', targets, o['t_prime'], o['t_cycle'], o['both_t']))
  for i, (d, orig, prime, cyc, (both, cyc_first)) in enumerate(doms):
    top = 'discriminator_' + d
    with streams.domain(i):
      terms['l_cyc_' + d] = ops.abs_diff_mean(orig, cyc, cfg.l_cyc_weight)
      if cyc_gan:      # both fool losses of the domain from the one batched prediction: one launch each way
        pred, _ = pggan.discriminator(P, both, cfg, top, groups=2, block_end_points=False)
        gc, gp = (0, 1) if cyc_first else (1, 0)
        tc, tp = ops.pred_losses(pred, pred.shape[0] // 2, _fool_jobs(gc, 0, cfg) + _fool_jobs(gp, 1, cfg), 2)
        terms['generator_fool_loss_cycle_' + d] = tc
        terms['generator_fool_loss_prime_' + d] = tp
      else:
        pp, _ = pggan.discriminator(P, prime, cfg, top, block_end_points=False)
        terms['generator_fool_loss_prime_' + d] = _fool_loss(pp, cfg)
  if 'e2' not in reenc:
    reencode()
  primes, e2 = o['primes'], reenc['e2']
  e_sp, e_tp = ops.rows(e2, [(0, b), (b, 2 * b)])
  if cfg.l_content_weight:
    terms['l_content_s'] = ops.abs_diff_mean(o['es'], e_tp, cfg.l_content_weight)
    terms['l_content_t'] = ops.abs_diff_mean(o['et'], e_sp, cfg.l_content_weight)
    if cfg.use_style_embedding:      # twingan.py:495-505: the style read back from s' / t' must be the random one
      st2, _ = pggan.encoder(P, primes, ('s', 't', b, 2), cfg, 'encoder_style')
      st_sp, st_tp = ops.rows(st2, [(0, b), (b, 2 * b)])
      terms['l_style_s'] = ops.abs_diff_mean(o['random_style_embed'], st_sp.contiguous(), cfg.l_content_weight)
      terms['l_style_t'] = ops.abs_diff_mean(o['random_style_embed'], st_tp.contiguous(), cfg.l_content_weight)
  if cfg.do_encoder_distillation:
    # twingan.py:207-230,290-298: both heads on the original and the re-encoded content (each head one batch: original
    # then prime); the cosine-distance terms only for the datasets that carry embeddings (:507-521)
    hs, ht = 'encoder_content/encoder_distillation_source', 'encoder_content/encoder_distillation_target'
    d_s, d_sp = ops.rows(pggan.encoder_classification(P, ops.cat_rows([o['es'], e_sp]), ('s', 's', b, 2), cfg, hs)[0], [(0, b), (b, 2 * b)])
    d_t, d_tp = ops.rows(pggan.encoder_classification(P, ops.cat_rows([o['et'], e_tp]), ('t', 't', b, 2), cfg, ht)[0], [(0, b), (b, 2 * b)])
    if cfg.hw >= cfg.distillation_start_hw:
      w = cfg.distillation_weight
      if distill_embed_s is not None:
        terms['l_source_distillation'] = ops.cosine_distance(distill_embed_s, d_s, w)
        terms['l_t_prime_distillation'] = ops.cosine_distance(distill_embed_s, d_tp, w)
      if distill_embed_t is not None:
        terms['l_target_distillation'] = ops.cosine_distance(distill_embed_t, d_t, w)
        terms['l_s_prime_distillation'] = ops.cosine_distance(distill_embed_t, d_sp, w)
  streams.join()
  return _sum_terms(terms), terms


def discriminator_loss(P, sources, targets, cfg, gp_alpha_s, gp_alpha_t, dragan_noise_s=None, dragan_noise_t=None,
                       style_noise=None):
  """DISCRIMINATOR_LOSSES (image_generation.py:348-412,414-476).  gp_alpha_*: fp32 [B] U[0,1) draws;
  dragan_noise_*: the U(-1,1) draws of get_perturbed_batch (image shaped; drawn on the device when None).
  E/G run without a tape: only discriminator variables are in the var_list (image_generation.py:605-610)."""
  assert cfg.loss_architecture in LOSSES, cfg.loss_architecture
  pggan.prepare_run(P, cfg)      # with the tape on: the discriminators' kernels are differentiated through sigma
  with torch.no_grad():
    if cfg.is_growing:
      sources, targets = get_growing_image(sources, cfg.alpha_grow), get_growing_image(targets, cfg.alpha_grow)
    o = forward_generators(P, sources, targets, cfg, style_noise)
  cyc_gan = cfg.hw >= 64 and cfg.do_l_cyc_gan
  terms = {}
  streams = _DomainStreams(sources.device, cfg.domain_streams)
  doms = (('s', sources, o['s_prime'], o['s_cycle'], gp_alpha_s, dragan_noise_s, o['both_s']),
          ('t', targets, o['t_prime'], o['t_cycle'], gp_alpha_t, dragan_noise_t, o['both_t']))
  if pggan.discriminator_pair_supported(P, cfg, cfg.hw):
    _d_pair_terms(P, cfg, terms, doms, cyc_gan, streams)
    if cfg.loss_architecture in ('wgan_gp', 'dragan'):
      _d_pair_gp(P, cfg, terms, doms, streams)
    doms = ()
  for i, (d, real, prime, cyc, a, noise, (both, cyc_first)) in enumerate(doms):
    top = 'discriminator_' + d
    with streams.domain(i):
      _d_domain_terms(P, cfg, terms, d, top, real, prime, cyc, a, cyc_gan, both, cyc_first)
      if cfg.loss_architecture in ('wgan_gp', 'dragan'):
        _d_domain_gp(P, cfg, terms, d, top, real, prime, a, noise)
  streams.join()
  return _sum_terms(terms), terms


def _d_domain_terms(P, cfg, terms, d, top, real, prime, cyc, a, cyc_gan, both=None, cyc_first=True):
  """One domain's DISCRIMINATOR_LOSSES (the body of the loop in image_generation.py:348-379,414-439).  ``both``: the cycle
  and the prime image as ONE tensor (rows of the generator's batch), the cycle image first or last (``cyc_first``)."""
  if True:
    # D(real), D(cyc), D(prime) of one domain share weights: one batch, one minibatch-stddev group per call
    if cyc_gan:
      if both is None:
        both, cyc_first = ops.cat_rows([cyc, prime]), True
      pred, _ = pggan.discriminator(P, ops.cat_rows([real, both]), cfg, top, groups=3, cut_seg=1, block_end_points=False)
      gc, gp = (1, 2) if cyc_first else (2, 1)      # groups of the batched prediction: 0 = real
      names, jobs = [], []
      _real_fake_jobs(names, jobs, '_cycle_' + d, gc, 0, cfg)      # only_real_fake_loss=True (twingan.py:466-474)
      _real_fake_jobs(names, jobs, '_prime_' + d, gp, 0, cfg)
    else:
      pred, _ = pggan.discriminator(P, ops.cat_rows([real, prime]), cfg, top, groups=2, cut_seg=1, block_end_points=False)
      names, jobs = [], []
      _real_fake_jobs(names, jobs, '_prime_' + d, 1, 0, cfg)
    if cfg.wgan_drift_loss_weight and cfg.loss_architecture in ('wgan_gp', 'wgan'):      # image_generation.py:360-367
      jobs.append((0, len(names), 3, 0.0, 0.0, cfg.wgan_drift_loss_weight))
      names.append('discriminator_drift_loss_prime_' + d)
    # every prediction loss of the domain from the one batched prediction: one launch each way (ops.PredLossesFn)
    groups = 3 if cyc_gan else 2
    for k, v in zip(names, ops.pred_losses(pred, pred.shape[0] // groups, jobs, len(names))):
      terms[k] = v


def _d_pair_terms(P, cfg, terms, doms, cyc_gan, streams):
  """_d_domain_terms for both domains with the discriminators' low-resolution layers as grouped launches
  (pggan.discriminator_pair): the per-domain batches [real; cyc; prime] go through their own heads, one tail."""
  srcs, metas = [], []
  for i, (d, real, prime, cyc, a, noise, (both, cyc_first)) in enumerate(doms):
    with streams.domain(i):
      if cyc_gan:
        if both is None:
          both, cyc_first = ops.cat_rows([cyc, prime]), True
        srcs.append(ops.cat_rows([real, both]))
      else:
        srcs.append(ops.cat_rows([real, prime]))
    metas.append((d, cyc_first))
  groups = 3 if cyc_gan else 2
  preds = pggan.discriminator_pair(P, srcs[0], srcs[1], cfg, groups=groups, cut_seg=1, streams=streams)
  for (d, cyc_first), pred in zip(metas, preds):
    names, jobs = [], []
    if cyc_gan:
      gc, gp = (1, 2) if cyc_first else (2, 1)      # groups of the batched prediction: 0 = real
      _real_fake_jobs(names, jobs, '_cycle_' + d, gc, 0, cfg)      # only_real_fake_loss=True (twingan.py:466-474)
      _real_fake_jobs(names, jobs, '_prime_' + d, gp, 0, cfg)
    else:
      _real_fake_jobs(names, jobs, '_prime_' + d, 1, 0, cfg)
    if cfg.wgan_drift_loss_weight and cfg.loss_architecture in ('wgan_gp', 'wgan'):      # image_generation.py:360-367
      jobs.append((0, len(names), 3, 0.0, 0.0, cfg.wgan_drift_loss_weight))
      names.append('discriminator_drift_loss_prime_' + d)
    for k, v in zip(names, ops.pred_losses(pred, pred.shape[0] // groups, jobs, len(names))):      # main stream, as the tail
      terms[k] = v


def _d_pair_gp(P, cfg, terms, doms, streams):
  """_d_domain_gp for both domains: one grouped pass of the two discriminators over the two domains' interpolates, ONE
  inner tf.gradients call for both (image_generation.py:414-439 per domain)."""
  interps = []
  for i, (d, real, prime, cyc, a, noise, _) in enumerate(doms):
    with streams.domain(i):
      if cfg.loss_architecture == 'dragan':
        if noise is None:
          noise = (torch.rand(real.shape, dtype=torch.float32, device=real.device) * 2.0 - 1.0).to(real.dtype)
        interps.append(ops.first_order_only(ops.dragan_interpolates(real, noise.contiguous(), a).requires_grad_(True)))
      else:
        interps.append(ops.first_order_only(ops.sample_lerp(real, prime, a).requires_grad_(True)))             # image_generation.py:420-424
  with ops.second_order():
    pis = pggan.discriminator_pair(P, interps[0], interps[1], cfg, streams=streams)
  ones = [ops.fill(pi.shape, 1.0, pi.dtype, pi.device) for pi in pis]
  with ops.no_param_grads():        # only d pred / d interp is needed here; parameters get theirs via the double backward
    gis = torch.autograd.grad(list(pis), interps, grad_outputs=ones, create_graph=True)  # tf.gradients(pred, interp)
  for i, ((d, *_), gi) in enumerate(zip(doms, gis)):
    with streams.domain(i):
      terms['discriminator_gradient_penalty_prime_' + d] = ops.gradient_penalty(gi.contiguous(), cfg.gradient_penalty_lambda)


def _d_domain_gp(P, cfg, terms, d, top, real, prime, a, noise=None, name=None):
  """Gradient penalty of one domain: WGAN-GP on real..fake interpolates (image_generation.py:414-439) or DRAGAN
  on real..perturbed-real ones (:441-476).  ``name``: the loss term's name (default: TwinGAN's per-domain one)."""
  if cfg.loss_architecture == 'dragan':
    if noise is None:
      noise = (torch.rand(real.shape, dtype=torch.float32, device=real.device) * 2.0 - 1.0).to(real.dtype)
    interp = ops.first_order_only(ops.dragan_interpolates(real, noise.contiguous(), a).requires_grad_(True))
  else:
    interp = ops.first_order_only(ops.sample_lerp(real, prime, a).requires_grad_(True))             # image_generation.py:420-424
  with ops.second_order():
    pi, _ = pggan.discriminator(P, interp, cfg, top, block_end_points=False)
  ones = ops.fill(pi.shape, 1.0, pi.dtype, pi.device)
  with ops.no_param_grads():        # only d pred / d interp is needed here; parameters get theirs via the double backward
    gi, = torch.autograd.grad(pi, interp, grad_outputs=ones, create_graph=True)  # tf.gradients(pred, interp)
  terms[name or 'discriminator_gradient_penalty_prime_' + d] = ops.gradient_penalty(gi.contiguous(), cfg.gradient_penalty_lambda)


BATCH_RENORM_BOUNDARIES = (10000, 20000, 30000)                      # nets/pggan_utils.py:43-47
BATCH_RENORM_RMAX, BATCH_RENORM_RMIN, BATCH_RENORM_DMAX = (1.1, 1.5, 2.0, 4.0), (0.9, 0.66, 0.5, 0.25), (0.1, 0.3, 0.5, 1.0)


def renorm_clipping(global_step):
  """tf.train.piecewise_constant over BATCH_RENORM_BOUNDARIES: value i while step <= boundary i, the last one after."""
  i = sum(1 for b in BATCH_RENORM_BOUNDARIES if global_step > b)
  return BATCH_RENORM_RMAX[i], BATCH_RENORM_RMIN[i], BATCH_RENORM_DMAX[i]


class Trainer:
  """One data-parallel clone: parameters, Adam state and the alternating step.

  ``run`` is one ``session.run(train_op)`` of the reference.  With ``use_graph=True`` the two step kinds
  (generator/encoder apply, discriminator apply) are captured once as hipGraphs over static input
  buffers and replayed: the ~4-5 k kernel launches of a step then cost one graph launch on the host
  instead of a Python/ctypes round trip each.  Everything a replay needs lives on the device: the
  shared Adam step counter and bias-corrected rate (tg_adam_tick), the WGAN-GP alphas (device RNG),
  the weight packs (rebuilt by the captured optimiser tail).
  """

  def __init__(self, cfg, device='cuda', seed=0, world_size=1, process_group=None, use_graph=False, overlap=None):
    """``overlap``: cut the backward into segments and start the clone all-reduce of each segment's finished
    gradients while the next one runs (None: whenever there is more than one clone; True forces the segmented
    schedule for a single clone too -- same results, used by the tests)."""
    if cfg.spectral_norm and cfg.domain_streams:
      # Spectral norm keeps the discriminators on one stream.  The per-run normalised kernels are computed (and their
      # packs rebuilt) on the main stream by pggan.prepare_run before anything forks, so two streams are SAFE
      # (the full GPU suite passed with them) -- but on config 4 they measured 450.0 vs 453.9
      # images/s on one stream (gpurun_out r3c): the attention kernels fill the chip on their own.
      cfg = dataclasses.replace(cfg, domain_streams=False)
    self.cfg = cfg
    self.device = torch.device(device)
    if self.device.type == 'cuda' and self.device.index is None:
      self.device = torch.device('cuda', torch.cuda.current_device())
    self.world = world_size
    if world_size > 1 and self.device.type == 'cuda':
      # every clone draws its own GP alphas / style noise (the reference's clones own their random ops,
      # deployment/model_deploy.py:224-239); weights are initialised from the shared CPU seed below
      rank = torch.distributed.get_rank(process_group) if torch.distributed.is_initialized() else 0
      with torch.cuda.device(self.device):
        torch.cuda.manual_seed(1000003 * (seed + 1) + rank)
    # the device draws of this clone (GP alphas): Philox keyed by (seed, clone), counter on the device
    rank = torch.distributed.get_rank(process_group) if (world_size > 1 and torch.distributed.is_initialized()) else 0
    self._rng_seed = 1000003 * (seed + 1) + rank
    self._rng_state = torch.zeros(2, dtype=torch.int32, device=self.device)
    self.reducer = GradReducer(world_size, process_group)
    self.store = self._declare(ParamStore(self.device), cfg).build(seed)
    self.P = self.store.P
    self.n_critic_counter = 0       # image_generation.py:622-623
    self.global_step = 0            # advanced once per n_critic cycle, see _advance_counters
    self.adam_t = 0                 # one shared optimizer: beta powers advance on every apply (:554-561)
    self._adam_step_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
    self._lr_t_dev = torch.zeros(1, dtype=torch.float32, device=self.device)
    # --use_ttur (image_generation.py:554-561) builds a second AdamOptimizer for the discriminator, but the reference
    # only uses it to COMPUTE the discriminator gradients (:603-608); they are APPLIED by the generator's optimizer
    # (:640-646 pass `optimizer`, not `d_optimizer`, to maybe_apply_gradients).  So the flag changes no update: one
    # optimizer, one learning rate, one pair of beta powers -- pinned live (tests/test_reference_live.py, ttur).
    self._group_weights = {g: [self.P[k] for k, s in self.store.specs.items() if s['group'] == g and s['kind'] == 'conv_w']
                           for g in self.store.GROUPS}
    # segmented backward (params.grad_phase) for every configuration: spectrally normalised kernels are read through
    # per-run leaves (pggan._sn_compute / sn_segment_backward), a growing stage's interpolated skip end-point is a leaf of
    # the high segment, the style encoder stays whole in segment 0
    want = (world_size > 1) if overlap is None else bool(overlap)
    self.split = bool(want and cfg.overlap_cut_hw and cfg.hw > cfg.overlap_cut_hw)
    self._ptr_phase = {self.P[k].data_ptr(): ph for k, ph in self.store.phase.items()}
    self._extras = None             # per-run dataset fields besides the images (run(..., distill_embed_s=, distill_embed_t=))
    self.use_graph = use_graph
    self.graph_fallback_reason = None
    self.capture_note = None        # set when the segmented capture had to be replaced by one graph per step kind
    self._graphs = None
    self._static = None

  def close(self):
    """Releases the captured graphs and this trainer's entries in the pack / gradient-sink registries."""
    self._graphs = self._static = None
    self.store.close()

  # ---- optimiser --------------------------------------------------------------------------------
  def _nseg(self, group):
    return len(self.store.phase_bounds[group]) if self.split else 1

  def _reduce_start(self, group, seg):
    """deployment/model_deploy.py:473-503 (tf.add_n over clones) as RCCL sum all-reduces of the flat gradient buffer:
    the range whose gradients backward segment ``seg`` completed is enqueued on the communication stream (which waits
    for everything enqueued so far) and travels while the next segment runs; _reduce_finish() waits before Adam."""
    if not self.reducer.active:
      return
    g = self.store.grad[group]
    if self._nseg(group) == 1:
      self.reducer.start(g)
    else:
      lo, hi = self.store.phase_bounds[group][seg]
      self.reducer.start(g[lo:hi], n_buckets=1)

  def set_adam_step(self, t):
    """The shared optimiser has applied ``t`` times (restoring a checkpoint: beta powers = beta^(t+1))."""
    self.adam_t = int(t)
    self._adam_step_dev.fill_(int(t))

  def _uniform(self, n):
    """n fp32 U[0, 1) draws on the device (ops.uniform: the draw counter lives on the device, so a captured step draws
    new numbers on every replay)."""
    return ops.uniform(n, self._rng_seed, self._rng_state)

  def _adam(self, group):
    """tf.train.AdamOptimizer apply (model/model_inheritor.py:537-542) on the group's flat buffers, then
    refresh the bf16 weight packs of the convs that just moved."""
    c = self.cfg
    s = self.store
    st = torch.cuda.current_stream().cuda_stream
    self.adam_t += 1
    step_dev, lr_dev, lr = self._adam_step_dev, self._lr_t_dev, c.learning_rate
    call('tg_adam_tick', step_dev.data_ptr(), lr_dev.data_ptr(), lr, c.adam_beta1, c.adam_beta2, st)
    call('tg_adam_step', s.flat[group].data_ptr(), s.grad[group].data_ptr(), s.m[group].data_ptr(),
         s.v[group].data_ptr(), None, s.flat[group].numel(), 0.0, lr_dev.data_ptr(), c.adam_beta1,
         c.adam_beta2, c.opt_epsilon, 1.0 / c.loss_scale, st,
         work=('adam:numel%d' % s.flat[group].numel(), 0, 28 * s.flat[group].numel()))
    PackCache.refresh(self._group_weights[group])

  # ---- steps ------------------------------------------------------------------------------------
  def _grad_segments(self, group, sources, targets, gp_alpha_s=None, gp_alpha_t=None):
    """Generator over the backward segments of one step: forward + segment 0 (from the loss down to the cuts), then one
    resumed backward per further segment (ops.Cuts).  Yields (segment, (loss, terms)) when the gradients of that
    segment's phase (params.grad_phase) are final in the flat buffer -- the caller starts their all-reduce and asks
    for the next segment.  A single segment (one clone, growing stages, ...) is the plain loss.backward()."""
    nseg = self._nseg(group)
    self.store.zero_grad(group)
    self._set_requires_grad(g=group == 'g', d=group == 'd')
    self.P.__dict__['sn_sink_mode'] = bool(self.cfg.spectral_norm)      # w_bar leaves with gradient sinks (pggan._sn_leaf)
    if nseg > 1:
      ops.Cuts.begin()
    try:
      if group == 'g':
        loss, terms = self._generator_loss(sources, targets)
      else:
        b = targets.shape[0]
        if gp_alpha_s is None and gp_alpha_t is None:      # tf.random_uniform([batch]) per domain (image_generation.py:420-424)
          gp_alpha_s, gp_alpha_t = self._uniform(2 * b).split(b)
        elif gp_alpha_s is None or gp_alpha_t is None:
          draw = self._uniform(b)
          gp_alpha_s, gp_alpha_t = (draw if gp_alpha_s is None else gp_alpha_s), (draw if gp_alpha_t is None else gp_alpha_t)
        loss, terms = self._discriminator_loss(sources, targets, gp_alpha_s, gp_alpha_t)
      out = (loss.detach(), {k: v.detach() for k, v in terms.items()})
      if group == 'g' and self.cfg.use_gdrop:
        self._update_gdrop(out[0])
      k = loss_scale_for_clones(self.cfg.loss_scale, self.world)       # model_deploy.py:265-268,308-313
      scaled = loss if k == 1.0 else loss * k
      ops.GradSink.pair = True
      # the slab reductions of the filter gradients that feed gradient sinks: queued, one launch per backward segment
      ops.defer_slab_reductions(True)
      for seg in range(nseg):
        if seg == 0:
          if getattr(self, '_seed', None) is None or self._seed.device != scaled.device:
            self._seed = torch.ones(1, dtype=torch.float32, device=scaled.device)      # d loss / d loss, made once
          scaled.backward(self._seed)
        else:
          roots, grads = ops.Cuts.roots(seg)
          torch.autograd.backward(roots, grads)
        _DomainStreams.join_all(self.device)
        last = seg == nseg - 1
        # filter gradients still waiting for a pair: issue those this segment completes, keep the others
        ops.GradSink.flush(None if last else (lambda ptr, seg=seg: self._ptr_phase.get(ptr, 0) <= seg))
        ops.flush_slab_reductions()      # after the join: every queued slab is written, the launch is on the main stream
        if nseg > 1 or self.cfg.spectral_norm:
          # spectrally normalised kernels whose uses are all behind us: their accumulated d loss / d w_bar (the sinks the
          # flushes above completed) through the normalisation's backward, into the master kernels' gradients
          pggan.sn_segment_backward(self.P, lambda scope, seg=seg: last or self.store.phase.get(scope + '/weights', 0) <= seg)
        if last:
          pggan.end_run(self.P)
        yield seg, out
    finally:
      self.P.__dict__['sn_sink_mode'] = False
      ops.GradSink.pair = False
      ops.defer_slab_reductions(False)
      if nseg > 1:
        ops.Cuts.end()

  # the model: TwinGAN's two loss sums (subclasses swap them: image_generation.PgganTrainer)
  def _declare(self, store, cfg):
    return declare_twingan(store, cfg)

  def _generator_loss(self, sources, targets):
    ex = self._extras or {}
    return generator_loss(self.P, sources, targets, self.cfg, distill_embed_s=ex.get('distill_embed_s'),
                          distill_embed_t=ex.get('distill_embed_t'))

  def _discriminator_loss(self, sources, targets, gp_alpha_s, gp_alpha_t):
    return discriminator_loss(self.P, sources, targets, self.cfg, gp_alpha_s, gp_alpha_t)

  def _step(self, group, sources, targets, gp_alpha_s=None, gp_alpha_t=None):
    out = None
    for seg, out in self._grad_segments(group, sources, targets, gp_alpha_s, gp_alpha_t):
      self._reduce_start(group, seg)
    self.reducer.finish()
    self._adam(group)
    return out

  def g_step(self, sources, targets):
    return self._step('g', sources, targets)

  def d_step(self, sources, targets, gp_alpha_s=None, gp_alpha_t=None):
    return self._step('d', sources, targets, gp_alpha_s, gp_alpha_t)

  # ---- hipGraph capture ---------------------------------------------------------------------------
  def _snapshot(self):
    """Everything a training run mutates: parameters, Adam moments and step, non-trainable state (BatchNorm moving /
    renorm statistics, spectral-norm u), the host counters and the device RNG."""
    s = self.store
    return dict(flat={g: s.flat[g].clone() for g in s.GROUPS}, m={g: s.m[g].clone() for g in s.GROUPS},
                v={g: s.v[g].clone() for g in s.GROUPS}, state={k: v.clone() for k, v in s.state.items()},
                step=self._adam_step_dev.clone(), lr=self._lr_t_dev.clone(), draws=self._rng_state.clone(),
                host=(self.n_critic_counter, self.global_step, self.adam_t), rng=torch.cuda.get_rng_state(self.device))

  def _restore(self, snap):
    s = self.store
    with torch.no_grad():
      for g in s.GROUPS:
        s.flat[g].copy_(snap['flat'][g])
        s.m[g].copy_(snap['m'][g])
        s.v[g].copy_(snap['v'][g])
        s.grad[g].zero_()
      for k, v in snap['state'].items():
        s.state[k].copy_(v)
      self._adam_step_dev.copy_(snap['step'])
      self._lr_t_dev.copy_(snap['lr'])
      self._rng_state.copy_(snap['draws'])
    self.n_critic_counter, self.global_step, self.adam_t = snap['host']
    torch.cuda.set_rng_state(snap['rng'], self.device)
    for g in s.GROUPS:                      # the packs follow the restored masters
      PackCache.refresh(self._group_weights[g])

  def _capture(self, sources, targets):
    """Captures, per step kind, one graph per backward segment and one for the apply: the clone all-reduces (RCCL)
    run between them, outside any capture.  The eager warm-up (one real step of each kind: allocates every weight pack
    and job table once) is undone afterwards, so graph mode and eager mode follow the same trajectory."""
    if sources is None:
      self._static = dict(s=None, t=targets.clone())
    else:      # one allocation, [s; t]: the encoder's batch is then a view of it (ops.cat_rows), not a copy per run
      both = torch.cat([sources, targets], dim=0)
      self._static = dict(s=both[:sources.shape[0]], t=both[sources.shape[0]:], st=both)
    # dataset fields besides the images (the distillation embeddings): static buffers the captured graphs read, refilled
    # before every replay; the SET of fields is part of the capture
    if self._extras:
      self._static['extras'] = {k: v.clone() for k, v in self._extras.items()}
      self._extras = self._static['extras']
    st = self._static
    snap = self._snapshot()
    side = torch.cuda.Stream(device=self.device)
    side.wait_stream(torch.cuda.current_stream())
    try:
      with torch.cuda.stream(side):
        self.g_step(st['s'], st['t'])
        self.d_step(st['s'], st['t'])
      torch.cuda.current_stream().wait_stream(side)
      torch.cuda.synchronize(self.device)
    finally:
      self._restore(snap)
    torch.cuda.synchronize(self.device)
    graphs, outs = {}, {}
    pool = None
    # with a process group alive, its watchdog thread issues event queries while we capture: only this
    # thread's unsafe calls may invalidate the capture
    mode = 'thread_local' if (self.world > 1 or torch.distributed.is_initialized()) else 'global'
    adam_t = self.adam_t
    for kind in ('g', 'd'):
      segs = []
      gen = self._grad_segments(kind, st['s'], st['t'])
      try:
        for _ in range(self._nseg(kind)):
          gr = torch.cuda.CUDAGraph()
          with torch.cuda.graph(gr, pool=pool, capture_error_mode=mode):
            _, outs[kind] = next(gen)
          pool = gr.pool()
          segs.append(gr)
      finally:
        gen.close()
      ga = torch.cuda.CUDAGraph()
      with torch.cuda.graph(ga, pool=pool, capture_error_mode=mode):
        self._adam(kind)
      graphs[kind] = (segs, ga)
    self.adam_t = adam_t                          # the captured (not executed) applies
    self._graphs, self._outs = graphs, outs

  def static_inputs(self):
    """-> (sources, targets) static input buffers of the captured graphs (None before the capture / in eager mode).  A
    loader that writes its batch INTO them and passes them to run() saves the per-run copy (run() copies only tensors
    at other addresses)."""
    if not self.use_graph or self._graphs is None:
      return None
    return self._static['s'], self._static['t']

  def _abandon_capture(self):
    """Leaves no half-issued step behind after a failed capture: held filter gradients, open cuts, forked side streams."""
    torch.cuda.synchronize(self.device)
    ops.GradSink._held.clear()
    ops.GradSink.pair = False
    ops.defer_slab_reductions(False)
    ops.Cuts.end()
    # the per-run spectrally normalised kernels of the aborted capture (detached leaves, u' tensors from the capture pool
    # that never executed): dropped WITHOUT assigning u', so the next prepare_run() runs the power iteration again
    for name in ('sn_cache', 'sn_pending', 'sn_leaves'):
      d = self.P.__dict__.get(name)
      if d:
        d.clear()
    _DomainStreams.join_all(self.device)
    self.store.zero_grad('g')
    self.store.zero_grad('d')
    self._graphs = None
    self.adam_t = int(self._adam_step_dev.item())

  def _run_graph(self, kind, sources, targets):
    if self._graphs is None:
      import warnings
      try:
        self._capture(sources, targets)
      except Exception as e:
        reason = '%s: %s' % (type(e).__name__, e)
        self._abandon_capture()
        was_split = self.split
        if self.split:
          # the segmented schedule (one graph per backward segment, captured in thread_local mode beside a live
          # communicator) could not be recorded: record ONE graph per step kind instead -- the all-reduce then starts
          # after the whole backward (no overlap) -- and say so (bench.py prints capture_note on its line)
          self.capture_note = 'segmented capture failed (%s); re-captured unsegmented: the all-reduce is not overlapped' % reason
          warnings.warn(self.capture_note)
          self.split = False
          try:
            self._capture(sources, targets)
          except Exception as e2:
            reason = '%s; unsegmented: %s: %s' % (reason, type(e2).__name__, e2)
            self._abandon_capture()
      if self._graphs is None:      # keep training: eager launches are the same kernels, only the host cost differs
        # eager steps never needed a graph to run segmented: keep the overlapped all-reduce, and do not claim a re-capture
        self.split = was_split
        self.capture_note = None if not was_split else 'segmented capture failed (%s); eager launches, still segmented' % reason
        self.graph_fallback_reason = reason
        warnings.warn('hipGraph capture failed (%s); falling back to eager launches' % reason)
        self.use_graph = False
        return self.g_step(sources, targets) if kind == 'g' else self.d_step(sources, targets)
    st = self._static
    if sources is not None and sources.data_ptr() != st['s'].data_ptr():
      st['s'].copy_(sources)
    if targets.data_ptr() != st['t'].data_ptr():
      st['t'].copy_(targets)
    have, want = st.get('extras') or {}, self._extras or {}
    if set(have) != set(want):
      raise ValueError('the captured graphs were recorded with the dataset fields %s, this run brings %s'
                       % (sorted(have), sorted(want)))
    for k, v in want.items():
      if v.data_ptr() != have[k].data_ptr():
        have[k].copy_(v)
    segs, ga = self._graphs[kind]
    for seg, gr in enumerate(segs):
      gr.replay()
      self._reduce_start(kind, seg)
    self.reducer.finish()
    ga.replay()
    self.adam_t += 1
    return self._outs[kind]

  def run(self, sources, targets, gp_alpha_s=None, gp_alpha_t=None, **extras):
    """One ``session.run(train_op)`` of the reference (image_generation.py:640-652):
    n_critic_counter % n_critic == 0 -> generator/encoder apply, else discriminator apply.
    ``extras``: further dataset fields of the batch -- distill_embed_s / distill_embed_t ([B, D] fp32, the
    'a_embedding' / 'b_embedding' of --do_encoder_distillation); under graph replay they are copied into static buffers
    (the same fields on every run)."""
    is_g = self.n_critic_counter % self.cfg.n_critic == 0
    self._extras = extras or None
    import contextlib
    # the kernels go to the current device's stream (ops._stream)
    with (torch.cuda.device(self.device) if self.device.type == 'cuda' else contextlib.nullcontext()):
      if self.cfg.generator_norm_type in ('batch_renorm', 'batch_renorm_native'):
        self._set_renorm_clipping()
      if self.cfg.use_gdrop:
        self._set_gdrop_coef()
      if self.use_graph:
        assert gp_alpha_s is None and gp_alpha_t is None, 'graph mode draws the GP alphas on the device'
        out = self._run_graph('g' if is_g else 'd', sources, targets)
      elif is_g:
        out = self.g_step(sources, targets)
      else:
        out = self.d_step(sources, targets, gp_alpha_s, gp_alpha_t)
    self._advance_counters()
    return out

  def _advance_counters(self):
    """image_generation.py:640-652: apply_gradients(global_step=n_critic_counter) adds 1 to the counter; the
    predicate of `increase_global_step` is built under control_dependencies([grad_updates]) and reads the counter
    through its (non-resource, hence live) variable value, i.e. AFTER that increment: global_step advances at the end
    of the run that completes an n_critic cycle -- with n_critic = 2, at the end of every discriminator run, so both
    runs of a G+D pair see the same global step (alpha_grow, renorm clipping)."""
    self.n_critic_counter += 1
    if self.n_critic_counter % self.cfg.n_critic == 0:
      self.global_step += 1

  def _update_gdrop(self, generator_loss):
    """GanModel._maybe_add_gdrop_update_op (image_generation.py:563-585): gdrop_strength = coef * max(clip(mean(generator
    loss), 0, 1) - gdrop_lim, 0) ** gdrop_exp, coef = gdrop_coef once global_step > 100 (the ExponentialMovingAverage the
    reference also applies feeds nothing).  On the device, so a captured step replays it.  The reference's graph evaluates
    the generator loss in EVERY run and so also refreshes the variable in discriminator runs; here it is refreshed where
    the generator loss exists -- the generator runs.  Nothing reads the variable unless cfg.do_dgrop is set."""
    c, st = self.cfg, self.store.state
    cur = generator_loss.detach().float().mean().clamp(0.0, 1.0)
    st['gdrop_strength'].copy_((st['gdrop/coef'] * (cur - c.gdrop_lim).clamp_min(0.0) ** c.gdrop_exp).reshape(1))

  def _set_gdrop_coef(self):
    coef = self.cfg.gdrop_coef if self.global_step > 100 else 0.0      # tf.cond(tf.greater(global_step, 100), ...)
    if getattr(self, '_gdrop_coef', None) != coef:
      self._gdrop_coef = coef
      self.store.state['gdrop/coef'].fill_(coef)

  def _set_renorm_clipping(self):
    """get_renorm_clipping_params (nets/pggan_utils.py:40-50,207-223): rmax / rmin / dmax are piecewise constant in the
    global step.  They live in device scalars so a captured graph picks up the current values."""
    rmax, rmin, dmax = renorm_clipping(self.global_step)
    if getattr(self, '_renorm_clip', None) != (rmax, rmin, dmax):
      self._renorm_clip = (rmax, rmin, dmax)
      st = self.store.state
      st['renorm/rmax'].fill_(rmax)
      st['renorm/rmin'].fill_(rmin)
      st['renorm/dmax'].fill_(dmax)

  def _set_requires_grad(self, g, d):
    for name, s in self.store.specs.items():
      self.P[name].requires_grad_(g if s['group'] == 'g' else d)
    for pr in self.store.pairs.values():      # the stacked views of the discriminators' twin variables
      pr.requires_grad_(d)
