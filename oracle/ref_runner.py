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


def run(flags, sources, targets, global_step=0, seed=0, preset=None, want_grads=True, eager_updates=False, feed=None,
        embeddings=None):
  """flags: reference flag name -> value.  sources / targets: float arrays [B, H, W, 3].  preset: variable values
  (reference name -> array) to use instead of the reference's initialisers.  feed: placeholder name -> array for the
  inference branch (twingan.py:300-363: 'sources_ph', 'targets_ph', 'style_embed_ph'); its outputs come back under
  'custom' (custom_generated_{s,t}_style_{rand,source,target,ph}: is_training=False passes).  embeddings: (a, b)
  arrays [B, D] or None -- the dataset's 'a_embedding' / 'b_embedding' fields of --do_encoder_distillation."""
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
  core.STATE.eager_updates = bool(eager_updates)
  core.STATE.placeholder_batch = int(np.asarray(sources).shape[0])
  core.STATE.placeholder_feed = dict(feed or {})
  tfapi._ARG_STACK[:] = [{}]
  gs = tfapi.get_or_create_global_step()
  gs.t.fill_(int(global_step))
  S = core.Tensor(torch.tensor(np.asarray(sources, np.float64), requires_grad=True), core.float32, 'a_source')
  T = core.Tensor(torch.tensor(np.asarray(targets, np.float64), requires_grad=True), core.float32, 'b_source')
  networks = ref.GanModel._select_network(None)
  if F.use_style_embedding:
    # As written the reference cannot build this configuration: _clone_fn (twingan.py:203-205) forwards the
    # `self_attention_hw` entry of the shared kwargs (twingan.py:846-847) to pggan.encoder, whose signature
    # (nets/pggan.py:509-521) has no such parameter -> TypeError.  Pinned with that one keyword dropped, i.e. with
    # the style encoder's attention resolution at encoder_before_classification's default.
    style_fn = networks['encoder_style_network_fn']
    networks['encoder_style_network_fn'] = lambda *a, **k: style_fn(
      *a, **{kk: v for kk, v in k.items() if kk != 'self_attention_hw'})
  data = {'a_source': S, 'b_source': T}
  for key, emb in zip(('a_embedding', 'b_embedding'), embeddings or (None, None)):
    if emb is not None:
      data[key] = core.Tensor(torch.tensor(np.asarray(emb, np.float64)), core.float32, key)
  end_points = ref.GanModel._clone_fn(networks, None, None, data_batched=data, is_training=True, global_step=gs)

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
    # slim.get_model_variables(): what a stage's warm start restores (model/model_inheritor.py:612-614)
    model_variables=[k for k, v in core.STATE.variables.items()
                     if any(v is m for m in core.get_collection('model_variables'))],
    random=[(n, t.numpy().copy()) for n, t in core.STATE.random_log],
    g_terms={k: float(v.t) for k, v in g_terms.items()},
    d_terms={k: float(v.t) for k, v in d_terms.items()},
    end_points={k: v.t.detach().numpy().copy() for k, v in end_points.items()
                if isinstance(v, core.Tensor) and not k.startswith('custom_') and not k.endswith('_ph')},
  )
  res['g_loss'] = float(sum(v.t for v in g_terms.values()))
  res['d_loss'] = float(sum(v.t for v in d_terms.values()))
  res['custom'] = {k: v.t.detach().numpy().copy() for k, v in end_points.items()
                   if isinstance(v, core.Tensor) and k.startswith('custom_generated')}
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


# ---------------------------------------------------------------------------------------------------------------
# oracle Config <-> reference flags / names
# ---------------------------------------------------------------------------------------------------------------
GROW_STEPS = 1000      # max_number_of_steps used to express alpha_grow as a global step (grow_start_number_of_steps=0)


def flags_of(cfg):
  """The reference flags (twingan.py:40-92, image_generation.py:45-130, nets/pggan.py:24-60) an oracle Config sets."""
  return dict(
    train_image_size=cfg.hw, pggan_max_num_channels=cfg.max_ch, generator_norm_type=cfg.norm,
    do_pixel_norm=cfg.do_pixel_norm, use_unet=cfg.use_unet, is_growing=cfg.is_growing,
    max_number_of_steps=GROW_STEPS, grow_start_number_of_steps=0, loss_architecture=cfg.loss,
    gradient_penalty_lambda=cfg.gp_lambda, gan_weight=cfg.gan_weight, wgan_drift_loss_weight=cfg.drift,
    l_cyc_weight=cfg.l_cyc, l_content_weight=cfg.l_content, do_l_cyc_gan=cfg.do_l_cyc_gan,
    spectral_norm=cfg.spectral_norm, do_self_attention=cfg.do_self_attention, self_attention_hw=cfg.self_attention_hw,
    use_style_embedding=cfg.use_style_embedding, style_embed_size=cfg.style_embed_size,
    equalized_learning_rate=cfg.equalized, use_res_block=cfg.res_block,
    pggan_unet_max_concat_hw=getattr(cfg, 'unet_max_concat_hw', None),
    spectral_norm_in_non_discriminator=getattr(cfg, 'sn_non_disc', False),
    pggan_max_num_channels_dis=getattr(cfg, 'max_ch_dis', None),
    use_larger_filter_at_rgb_layer=getattr(cfg, 'larger_rgb', False),
    do_encoder_distillation=getattr(cfg, 'do_encoder_distillation', False),
    distillation_weight=getattr(cfg, 'distillation_weight', 1.0),
    distillation_start_hw=getattr(cfg, 'distillation_start_hw', 16))


def global_step_of(cfg):
  """alpha_grow = (global_step - grow_start) / (max_steps - grow_start)  (twingan.py:834-836)."""
  return int(round(cfg.alpha_grow * GROW_STEPS)) if cfg.is_growing else int(cfg.global_step)


def term_name(ref_name):
  """Reference loss-op name -> the oracle's / product's term name.  The reference builds the s-domain losses first
  (twingan.py:453), so within one graph 'x' is the s term and the uniquified 'x_1' the t term."""
  fixed = {'l_source_content_before_classification': 'l_content_s', 'l_target_content_before_classification': 'l_content_t',
           'l_source_style_prediction': 'l_style_s', 'l_target_style_prediction': 'l_style_t'}
  if ref_name.endswith('_distillation'):
    return ref_name
  if ref_name in fixed:
    return fixed[ref_name]
  if ref_name.startswith('l_cyc_'):
    return ref_name
  return ref_name[:-2] + '_t' if ref_name.endswith('_1') else ref_name + '_s'


def run_stage_driver(start_hw, max_hw, hw_to_batch_size, num_images_per_resolution, train_dir='run'):
  """Executes pggan_runner.main (/root/reference/pggan_runner.py:82-160) with an empty train_dir and a recording
  stand-in for the per-stage program, and returns the flags it sets for every stage, in order."""
  import os
  tf = loader.install()
  import pggan_runner as ref
  F = tf.flags.FLAGS
  F.start_hw, F.max_hw, F.hw_to_batch_size = start_hw, max_hw, repr(dict(hw_to_batch_size))
  F.num_images_per_resolution, F.train_dir, F.is_training, F.do_export = num_images_per_resolution, train_dir, True, False
  stages = []

  class Recorder(object):
    def main(self):
      stages.append(dict(name=os.path.relpath(F.train_dir, train_dir), hw=int(F.train_image_size),
                         is_growing=bool(F.is_growing), batch_size=int(F.batch_size),
                         max_number_of_steps=int(F.max_number_of_steps),
                         ignore_missing_vars=bool(F.ignore_missing_vars),
                         checkpoint_path=(os.path.relpath(F.checkpoint_path, train_dir)
                                          if 'checkpoint_path' in F and F.checkpoint_path else None)))
  saved = ref.select_program
  ref.select_program = lambda name: Recorder()
  try:
    F.checkpoint_path = None
    ref.main(None)
  finally:
    ref.select_program = saved
  return stages


def run_clones(flags, batches, global_step=0, seed=0, preset=None):
  """Data-parallel clones as the reference builds them (model/model_inheritor.py:1006-1045 -> deployment/
  model_deploy.py): create_clones() calls GanModel._clone_fn once per clone (variables shared, one dequeued batch
  each), optimize_clones() divides every clone's loss by num_clones and sums the per-variable gradients with
  tf.add_n.  batches: [(sources, targets)] per clone.  Returns per-clone loss terms and random draws, the two total
  losses and the summed gradients."""
  tf = loader.install()
  import twingan as ref
  from deployment import model_deploy
  F = tf.flags.FLAGS
  if not hasattr(run, '_defaults'):
    run._defaults = F.flag_values_dict()
  for k, v in run._defaults.items():
    setattr(F, k, v)
  for k, v in dict(BASE_FLAGS, **flags).items():
    setattr(F, k, v)
  core.STATE.reset(seed)
  core.STATE.preset = dict(preset or {})
  tfapi._ARG_STACK[:] = [{}]
  gs = tfapi.get_or_create_global_step()
  gs.t.fill_(int(global_step))

  class Queue(object):      # the prefetch queue of model_inheritor.py:1033-1034: one batch per dequeue()
    def __init__(self):
      self.i = 0

    def dequeue(self):
      s, t = batches[self.i]
      self.i += 1
      return [core.Tensor(torch.tensor(np.asarray(a, np.float64), requires_grad=True), core.float32) for a in (s, t)]

  config = model_deploy.DeploymentConfig(num_clones=len(batches))
  networks = ref.GanModel._select_network(None)
  marks = [0]

  def model_fn(*a, **k):
    out = ref.GanModel._clone_fn(*a, **k)
    marks.append(len(core.STATE.random_log))
    return out
  clones = model_deploy.create_clones(config, model_fn, args=[networks, Queue(), ['a_source', 'b_source']],
                                      kwargs=dict(is_training=True, global_step=gs))
  names = [k for k, v in core.STATE.variables.items() if v.trainable]
  model = ref.GanModel.__new__(ref.GanModel)      # the variable selection of image_generation.py:487-501
  var_lists = dict(g=model._get_generator_variables_to_train(), d=model._get_discriminator_variables_to_train())
  model._check_trainable_vars(var_lists['g'], var_lists['d'])
  res = dict(trainable=names, variables={k: v.t.detach().numpy().copy() for k, v in core.STATE.variables.items()},
             clones=[])
  for i, clone in enumerate(clones):
    terms = {}
    for grp, coll in (('g', ref.GENERATOR_LOSS_COLLECTION), ('d', ref.DISCRIMINATOR_LOSS_COLLECTION)):
      out = {}
      for l in core.get_collection(coll, clone.scope):
        name = l.name[len(clone.scope):-len('/value:0')]
        k, n = name, 0
        while k in out:
          n += 1
          k = '%s_%d' % (name, n)
        out[k] = float(l.t)
      terms[grp] = out
    res['clones'].append(dict(scope=clone.scope, g_terms=terms['g'], d_terms=terms['d'],
                              random=[(n, t.numpy().copy()) for n, t in
                                      core.STATE.random_log[marks[i]:marks[i + 1]]]))
  opt = tfapi.Optimizer()
  for grp, coll in (('g', ref.GENERATOR_LOSS_COLLECTION), ('d', ref.DISCRIMINATOR_LOSS_COLLECTION)):
    total, gv = model_deploy.optimize_clones(clones, opt, gradient_scale=1.0, loss_collection=coll,
                                             var_list=var_lists[grp])
    res[grp + '_loss'] = float(total.t)
    res[grp + '_grads'] = {v.op.name: g.t.detach().numpy().copy() for g, v in gv}
  return res


def run_training(flags, runs, seed=0, preset=None, global_step=0):
  """`len(runs)` consecutive session.run(train_op) calls of the reference's training graph: clones
  (model_deploy.create_clones around GanModel._clone_fn), learning rate and optimizer (model_inheritor.py:471-542),
  and GanModel._add_optimization (image_generation.py:587-662: generator / discriminator gradients through
  optimize_clones, the n_critic alternation under tf.cond, apply_gradients, the global-step increment).  The stand-in
  is eager, so the graph is rebuilt for every run with all variables (incl. the optimizer's) reused, and every state
  update executes where it is created -- the order the reference's control dependencies prescribe.
  runs: [(sources, targets)].  Returns per-run losses / counters / random draws and the final variable values."""
  tf = loader.install()
  import twingan as ref
  from deployment import model_deploy
  F = tf.flags.FLAGS
  if not hasattr(run, '_defaults'):
    run._defaults = F.flag_values_dict()
  for k, v in run._defaults.items():
    setattr(F, k, v)
  for k, v in dict(BASE_FLAGS, **flags).items():
    setattr(F, k, v)
  core.STATE.reset(seed)
  core.STATE.preset = dict(preset or {})
  core.STATE.eager_updates = True
  gs = tfapi.get_or_create_global_step()
  gs.t.fill_(int(global_step))
  model = ref.GanModel.__new__(ref.GanModel)
  networks = ref.GanModel._select_network(None)
  config = model_deploy.DeploymentConfig(num_clones=1)
  history = []
  for i, (s, t) in enumerate(runs):
    tfapi._ARG_STACK[:] = [{}]
    core.STATE.scope_count = {}
    core.STATE.opt_count = 0                                       # optimizer objects are numbered per graph build
    core.STATE.scope_stack[0].reuse = True if i > 0 else None      # the same graph, built again
    for k in list(core.STATE.collections):                       # per-run collections (losses, update ops ...)
      if k not in ('variables', 'trainable_variables', 'model_variables'):
        del core.STATE.collections[k]
    mark = len(core.STATE.random_log)

    class Queue(object):
      def dequeue(self):
        return [core.Tensor(torch.tensor(np.asarray(a, np.float64), requires_grad=True), core.float32) for a in (s, t)]
    clones = model_deploy.create_clones(config, ref.GanModel._clone_fn, args=[networks, Queue(), ['a_source', 'b_source']],
                                        kwargs=dict(is_training=True, global_step=gs))
    update_ops = core.get_collection(core.GraphKeys.UPDATE_OPS, config.clone_scope(0))
    before = dict(n_critic_counter=int(core.STATE.variables['n_critic_counter'].t) if i else 0, global_step=int(gs.t))
    lr = model._configure_learning_rate(1000, gs)
    optimizer = model._configure_optimizer(lr)
    train = model._add_optimization(clones, optimizer, set(), update_ops, gs)
    history.append(dict(
      before, train_tensor=float(train.t), learning_rate=float(core.raw(lr)),
      random=[(n, v.numpy().copy()) for n, v in core.STATE.random_log[mark:]],
      n_critic_counter_after=int(core.STATE.variables['n_critic_counter'].t), global_step_after=int(gs.t)))
  return dict(history=history,
              variables={k: v.t.detach().numpy().copy() for k, v in core.STATE.variables.items()})


def run_pggan(flags, targets, global_step=0, seed=0, preset=None, want_grads=True):
  """BASELINE configs[0]: the plain PGGAN trainer -- image_generation.GanModel._clone_fn
  (/root/reference/image_generation.py:194-316) with no generator input, i.e. pggan.generator drawing its own
  tf.random_normal noise [B,1,1,C] (nets/pggan.py:86-153), one discriminator, add_gan_loss.  Returns the same kind of
  record as run(): variables, random draws (the noise first), loss terms, gradients, end points."""
  tf = loader.install()
  import image_generation as ref
  F = tf.flags.FLAGS
  if not hasattr(run, '_defaults'):
    run._defaults = F.flag_values_dict()
  for k, v in run._defaults.items():
    setattr(F, k, v)
  base = {k: v for k, v in BASE_FLAGS.items() if k in run._defaults}      # twingan.py's own flags may not be defined
  for k, v in dict(base, **flags).items():
    if k not in run._defaults:
      raise KeyError('the reference defines no flag %r' % k)
    setattr(F, k, v)
  core.STATE.reset(seed)
  core.STATE.preset = dict(preset or {})
  core.STATE.eager_updates = False
  core.STATE.placeholder_batch = int(np.asarray(targets).shape[0])
  tfapi._ARG_STACK[:] = [{}]
  gs = tfapi.get_or_create_global_step()
  gs.t.fill_(int(global_step))
  T = core.Tensor(torch.tensor(np.asarray(targets, np.float64), requires_grad=True), core.float32, 'target')
  networks = ref.GanModel._select_network(None)
  # the image post-processing of the `custom_generated_targets` inference end point lives in the (stubbed, out of
  # scope) preprocessing package: identity here -- it does not feed any loss
  keep = ref.GanModel._post_process_image
  ref.GanModel._post_process_image = staticmethod(lambda image: image)
  try:
    end_points = ref.GanModel._clone_fn(networks, None, None, data_batched={'target': T}, is_training=True, global_step=gs)
  finally:
    ref.GanModel._post_process_image = keep

  def losses(coll):
    return {l.name[:-len('/value:0')]: l for l in core.get_collection(coll)}
  g_terms, d_terms = losses(ref.GENERATOR_LOSS_COLLECTION), losses(ref.DISCRIMINATOR_LOSS_COLLECTION)
  res = dict(
    variables={k: v.t.detach().numpy().copy() for k, v in core.STATE.variables.items()},
    trainable=[k for k, v in core.STATE.variables.items() if v.trainable],
    random=[(n, t.numpy().copy()) for n, t in core.STATE.random_log],
    g_terms={k: float(v.t) for k, v in g_terms.items()}, d_terms={k: float(v.t) for k, v in d_terms.items()},
    end_points={k: v.t.detach().numpy().copy() for k, v in end_points.items()
                if isinstance(v, core.Tensor) and not k.startswith('custom_') and not k.endswith('_ph')})
  res['g_loss'] = float(sum(v.t for v in g_terms.values()))
  res['d_loss'] = float(sum(v.t for v in d_terms.values()))
  if want_grads:
    names = res['trainable']
    leaves = [core.STATE.variables[k].t for k in names]
    for tag, terms in (('g_grads', g_terms), ('d_grads', d_terms)):
      total = sum(v.t for v in terms.values())
      gr = torch.autograd.grad(total, leaves, allow_unused=True, retain_graph=True)
      res[tag] = {k: g.numpy().copy() for k, g in zip(names, gr) if g is not None}
  return res


def run_discriminator(flags, images, scope='discriminator_s', seed=0, preset=None, **kwargs):
  """One call of the reference's pggan.discriminator (/root/reference/nets/pggan.py:337-376) under variable_scope(scope),
  with the keyword arguments the caller gives (do_dgrop, gdrop_strength, is_training ...: what no flag of the trainers
  reaches).  Returns the prediction, the random draws in call order (the gdrop noises) and the variables."""
  tf = loader.install()
  import twingan as ref      # defines the flags the network reads
  from nets import pggan
  F = tf.flags.FLAGS
  if not hasattr(run, '_defaults'):
    run._defaults = F.flag_values_dict()
  for k, v in run._defaults.items():
    setattr(F, k, v)
  for k, v in dict(BASE_FLAGS, **flags).items():
    setattr(F, k, v)
  core.STATE.reset(seed)
  core.STATE.preset = dict(preset or {})
  core.STATE.eager_updates = False
  tfapi._ARG_STACK[:] = [{}]
  x = core.Tensor(torch.tensor(np.asarray(images, np.float64), requires_grad=True), core.float32, 'images')
  with tf.variable_scope(scope):
    pred, end_points = pggan.discriminator(x, **kwargs)
  return dict(prediction=pred.t.detach().numpy().copy(),
              random=[(n, t.numpy().copy()) for n, t in core.STATE.random_log],
              variables={k: v.t.detach().numpy().copy() for k, v in core.STATE.variables.items()})


def run_preprocess(image_u8, hw, resize_mode='PAD', is_training=True, seed=0, do_random_cropping=False, color_space='rgb',
                   initial_crop_hw=None):
  """The reference's OWN preprocessing/danbooru_preprocessing.preprocess_image (the TwinGAN trainer's image
  preprocessing, model/model_inheritor.py:403-457) executed on the TF stand-in for one uint8 image [h, w, 3].
  Returns (output [hw, hw, 3] float64, draws) with draws = dict(flip_uniform, sel, applied=[(kind, value), ...]) --
  the random values the code drew for the branch that was live, so that a restatement can be fed the same ones."""
  import importlib
  tf = loader.install()
  from .tf_shim import core
  importlib.import_module('preprocessing.preprocessing_util')
  pre = importlib.import_module('preprocessing.danbooru_preprocessing')
  tf.flags.FLAGS.random_crop_and_reshape_initial_crop_hw = initial_crop_hw      # preprocessing_util.py:26-27
  core.STATE.random_log.clear()
  core.STATE.aug_log.clear()
  core.STATE.gen.manual_seed(seed)
  img = tf.Tensor(torch.tensor(np.asarray(image_u8), dtype=torch.float64), tf.uint8, 'image')
  out = pre.preprocess_image(img, hw, hw, dtype=tf.float32, resize_mode=resize_mode, is_training=is_training,
                             add_image_summaries=False, do_random_cropping=do_random_cropping, color_space=color_space)
  # draw order of a training call: [crop height, crop width, crop offsets] (only with do_random_cropping), the flip
  # uniform, the ordering selector (not for 'gray'), then the distortions of the live branch
  log = [(n, v) for n, v in core.STATE.random_log]
  k = 3 if (is_training and do_random_cropping) else 0
  applied = [a for a in core.STATE.aug_log if a[0] != 'crop']
  crops = [a[1] for a in core.STATE.aug_log if a[0] == 'crop']
  mode_crop = None
  if resize_mode in ('RANDOM_CROP', 'RANDOM_CROP_AND_RESHAPE'):      # _random_crop_to_hw draws its offsets first, in any mode
    mode_crop, crops, k = crops[0], crops[1:], k + 1
  draws = dict(flip_uniform=float(log[k][1]) if is_training else None,
               sel=int(log[k + 1][1]) if (is_training and color_space != 'gray') else None,
               applied=applied, crop=crops[0] if crops else None, mode_crop=mode_crop)
  return out.t.detach().numpy().copy(), draws
