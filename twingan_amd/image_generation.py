"""The plain PGGAN trainer on the same kernels -- BASELINE configs[0] ("4x4 PGGAN stage-0, batch 16").

Mirror of image_generation.GanModel (image_generation.py:194-476): the generator draws latent noise
[B, 1, 1, get_num_channels(1)] (nets/pggan.py:86-153), one discriminator (scope 'discriminator') scores the real
targets and the generated images, add_gan_loss supplies the --loss_architecture terms (:318-412) and the gradient
penalty (:414-476); _add_optimization alternates the generator and discriminator applies exactly as for TwinGAN
(:587-662), so the step machinery -- flat parameter store, device Adam, hipGraph capture, the clone all-reduce -- is
twingan.Trainer's.  No encoder, no domain postfix on the normaliser variables, scopes 'generator' / 'discriminator'.
"""
import torch

from . import ops, pggan
from .params import declare_pggan
from .twingan import LOSSES, Trainer, _d_domain_gp, _fool_loss, _real_fake_losses, _sum_terms, get_growing_image

get_noise_shape = pggan.get_noise_shape      # networks['get_noise_shape'] of GanModel._select_network (:214-226)


def generate(P, noise, cfg):
  """generator_network_fn(None, ...) with the noise supplied by the caller: [B,1,1,C] or [B,C] -> images."""
  return pggan.generator(P, noise, '', cfg, None, 'generator')[0]


def generator_loss(P, targets, cfg, noise):
  """GENERATOR_LOSSES of the plain trainer: generator_fool_loss (image_generation.py:331-344).  Returns (total, terms)."""
  assert cfg.loss_architecture in LOSSES, cfg.loss_architecture
  pggan.prepare_run(P, cfg)
  fake = generate(P, noise, cfg)
  pred, _ = pggan.discriminator(P, fake, cfg, 'discriminator', block_end_points=False)
  terms = {'generator_fool_loss': _fool_loss(pred, cfg)}
  return _sum_terms(terms), terms


def discriminator_loss(P, targets, cfg, noise, gp_alpha, dragan_noise=None):
  """DISCRIMINATOR_LOSSES (image_generation.py:348-476): real / fake terms, drift, gradient penalty."""
  assert cfg.loss_architecture in LOSSES, cfg.loss_architecture
  pggan.prepare_run(P, cfg)
  with torch.no_grad():
    if cfg.is_growing:
      targets = get_growing_image(targets, cfg.alpha_grow)      # get_growing_source_and_target (:985-1006)
    fake = generate(P, noise, cfg)
  b = targets.shape[0]
  pred, _ = pggan.discriminator(P, ops.cat_rows([targets, fake]), cfg, 'discriminator', groups=2, block_end_points=False)
  pr, pf = ops.rows(pred, [(0, b), (b, 2 * b)])
  terms = {}
  _real_fake_losses(terms, '', pf, pr, cfg)
  if cfg.wgan_drift_loss_weight and cfg.loss_architecture in ('wgan_gp', 'wgan'):
    terms['discriminator_drift_loss'] = ops.square_mean(pr, cfg.wgan_drift_loss_weight)
  if cfg.loss_architecture in ('wgan_gp', 'dragan'):
    _d_domain_gp(P, cfg, terms, '', 'discriminator', targets, fake, gp_alpha, dragan_noise,
                 name='discriminator_gradient_penalty')
  return _sum_terms(terms), terms


class PgganTrainer(Trainer):
  """One clone of the plain PGGAN trainer.  ``run(None, targets)``: the first argument (TwinGAN's source batch) is not
  used; the latent noise is drawn on the device each run (tf.random_normal, nets/pggan.py:136-137)."""

  def _declare(self, store, cfg):
    return declare_pggan(store, cfg)

  def _noise(self, b):
    shape = get_noise_shape(b, self.cfg.max_ch)
    z = torch.randn(shape, dtype=torch.float32, device=self.device)
    return z.to({'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}[self.cfg.precision])

  def _generator_loss(self, sources, targets):
    return generator_loss(self.P, targets, self.cfg, self._noise(targets.shape[0]))

  def _discriminator_loss(self, sources, targets, gp_alpha_s, gp_alpha_t):
    return discriminator_loss(self.P, targets, self.cfg, self._noise(targets.shape[0]), gp_alpha_t)
