"""TEST INFRASTRUCTURE -- executes the REFERENCE's own graph-building code (twingan.GanModel._clone_fn of
/root/reference/twingan.py:146-449, which builds encoders, generators, discriminators and every loss through
nets/pggan.py, nets/pggan_utils.py, libs/* and image_generation.py:318-439) on the eager TF stand-in of
oracle/tf_shim, and returns what a `session.run` of the two loss sums would have seen:

  * every variable the reference created (name -> value), so the oracle / the product can be loaded with them;
  * every random draw (WGAN-GP / DRAGAN alphas, style noise) in call order;
  * every loss the reference added to GENERATOR_LOSSES / DISCRIMINATOR_LOSSES, by its scope name;
  * the gradients of the two loss sums with respect to the variables (torch autograd over the same graph);
  * selected end points (generated images, predictions).

This is how the oracle is pinned (tools/make_ref_golden.py -> tests/golden/ref_*.npz).  It only works where
/root/reference exists, i.e. in the build container; nothing at test time on the GPU box needs it.
"""
import numpy as np
import torch

from .tf_shim import core, loader, tfapi

# the flags a 256x256 TwinGAN run sets (docs/training.md + BASELINE.json), on top of the reference's own defaults
BASE_FLAGS = dict(generator_network='pggan', is_growing=False, loss_architecture='wgan_gp', do_pixel_norm=True,
                  generator_norm_type='instance_norm', use_unet=True, gradient_penalty_lambda=10.0,
                  use_gdrop=False, use_conditional_labels=False, do_encoder_distillation=False)


def run(flags, sources, targets, global_step=0, seed=0, preset=None, want_grads=True):
  """flags: reference flag name -> value.  sources / targets: float arrays [B, H, W, 3].  preset: variable values
  (reference name -> array) to use instead of the reference's initialisers."""
  tf = loader.install()
  import twingan as ref      # the reference module, loaded by oracle.tf_shim.loader
  F = tf.flags.FLAGS
  if not hasattr(run, '_defaults'):
    run._defaults = F.flag_values_dict()
  for k, v in run._defaults.items():
    setattr(F, k, v)
  for k, v in dict(BASE_FLAGS, **flags).items():
    if k not in run._defaults:
      raise KeyError('the reference defines no flag %r' % k)
    setattr(F, k, v)
  core.STATE.reset(seed)
  core.STATE.preset = dict(preset or {})
  tfapi._ARG_STACK[:] = [{}]
  gs = tfapi.get_or_create_global_step()
  gs.t.fill_(int(global_step))
  S = core.Tensor(torch.tensor(np.asarray(sources, np.float64), requires_grad=True), core.float32, 'a_source')
  T = core.Tensor(torch.tensor(np.asarray(targets, np.float64), requires_grad=True), core.float32, 'b_source')
  networks = ref.GanModel._select_network(None)
  end_points = ref.GanModel._clone_fn(networks, None, None, data_batched={'a_source': S, 'b_source': T},
                                      is_training=True, global_step=gs)

  def losses(coll):
    out = {}      # op names are uniquified the way a tf.Graph does it: the 2nd 'x' becomes 'x_1' (= the t domain)
    for l in core.get_collection(coll):
      name = l.name[:-len('/value:0')]
      k, n = name, 0
      while k in out:
        n += 1
        k = '%s_%d' % (name, n)
      out[k] = l
    return out
  g_terms, d_terms = losses(ref.GENERATOR_LOSS_COLLECTION), losses(ref.DISCRIMINATOR_LOSS_COLLECTION)
  res = dict(
    variables={k: v.t.detach().numpy().copy() for k, v in core.STATE.variables.items()},
    trainable=[k for k, v in core.STATE.variables.items() if v.trainable],
    random=[(n, t.numpy().copy()) for n, t in core.STATE.random_log],
    g_terms={k: float(v.t) for k, v in g_terms.items()},
    d_terms={k: float(v.t) for k, v in d_terms.items()},
    end_points={k: v.t.detach().numpy().copy() for k, v in end_points.items()
                if isinstance(v, core.Tensor) and not k.startswith('custom_') and not k.endswith('_ph')},
  )
  res['g_loss'] = float(sum(v.t for v in g_terms.values()))
  res['d_loss'] = float(sum(v.t for v in d_terms.values()))
  if want_grads:
    names = res['trainable']
    leaves = [core.STATE.variables[k].t for k in names]
    for tag, terms in (('g_grads', g_terms), ('d_grads', d_terms)):
      total = sum(v.t for v in terms.values())
      gr = torch.autograd.grad(total, leaves, allow_unused=True, retain_graph=True)
      res[tag] = {k: g.numpy().copy() for k, g in zip(names, gr) if g is not None}
  # what the update ops of this run would leave in the non-trainable state (moving averages, spectral-norm u)
  tfapi.run_update_ops()
  res['state_after'] = {k: v.t.detach().numpy().copy() for k, v in core.STATE.variables.items() if not v.trainable}
  return res
