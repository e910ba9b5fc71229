"""Progressive-growing stage driver: the MI355X counterpart of pggan_runner.main (pggan_runner.py:82-160).

The reference rewrites tf.flags per stage, builds a fresh TF graph and warm-starts from the previous stage's
checkpoint with ``ignore_missing_vars=is_growing`` (the new resolution's layers initialise fresh,
pggan_runner.py:136-146; model/model_inheritor.py:576-644).  Here a stage is a ``Config`` + a ``Trainer``; the
warm start copies every variable that exists in both stages (same TF names).
"""
import math
from dataclasses import replace

DEFAULT_HW_TO_BATCH_SIZE = {4: 16, 8: 16, 16: 16, 32: 16, 64: 12, 128: 12, 256: 12, 512: 6}      # pggan_runner.py:52-57
TWINGAN_HW_TO_BATCH_SIZE = {4: 8, 8: 8, 16: 8, 32: 8, 64: 8, 128: 4, 256: 3, 512: 2}              # docs/training.md:34
LAST_STAGE_STEPS = 10000000                                                                        # pggan_runner.py:103-104


def stage_schedule(start_hw=4, max_hw=256, hw_to_batch_size=None, num_images_per_resolution=300000):
  """[(train_dir_name, hw, is_growing, batch_size, max_number_of_steps)] in training order
  (pggan_runner.py:90-109): for every resolution a growing stage ('{hw/2}to{hw}') then a stable one ('{hw}');
  the first resolution has no growing stage; the last stable stage trains "indefinitely"."""
  hw_to_batch_size = hw_to_batch_size or DEFAULT_HW_TO_BATCH_SIZE
  resolutions = [2 ** i for i in range(int(math.log(start_hw, 2)), int(math.log(max_hw, 2)) + 1)]
  out = []
  for res in resolutions:
    batch_size = hw_to_batch_size[res]
    steps = int(num_images_per_resolution / batch_size)
    for is_growing in (True, False):
      if is_growing and res == resolutions[0]:
        continue
      n = LAST_STAGE_STEPS if (res == resolutions[-1] and not is_growing) else steps
      name = '%dto%d' % (res // 2, res) if is_growing else '%d' % res
      out.append((name, res, is_growing, batch_size, n))
  return out


def alpha_grow(global_step, max_number_of_steps, grow_start_number_of_steps=0):
  """twingan.py:833-835 / image_generation.py:1015-1017: linear fade-in over the stage."""
  return float(global_step - grow_start_number_of_steps) / float(max_number_of_steps - grow_start_number_of_steps)


def warm_start(trainer, previous_state):
  """Loads every variable of ``previous_state`` (a ParamStore.state_dict()) that this stage also has and whose
  shape matches; the rest keep their fresh initialisation -- slim's assign_from_checkpoint_fn with
  ignore_missing_vars (model/model_inheritor.py:576-644).  Returns the names that were loaded."""
  from .params import is_model_variable
  specs, state = trainer.store.specs, trainer.store.state
  usable = {k: v for k, v in previous_state.items()
            if is_model_variable(k) and      # sa_gamma is not a slim model variable: fresh (0) every stage
            ((k in specs and tuple(v.shape) == specs[k]['shape']) or
             (k in trainer.store.state_specs and tuple(v.shape) == tuple(state[k].shape)))}
  trainer.store.load_state_dict(usable, strict=False)
  return sorted(usable)


def run_progressive(base_cfg, batch_fn, start_hw=4, max_hw=256, hw_to_batch_size=None, num_images_per_resolution=300000,
                    device='cuda', seed=0, max_steps_per_stage=None, use_graph=False, on_stage_end=None, train_dir=None,
                    save_interval_secs=600, save_interval_steps=None, max_to_keep=5):
  """Trains stage after stage.  ``batch_fn(hw, batch_size)`` -> (sources, targets) device tensors, or
  (sources, targets, gp_alpha_s, gp_alpha_t) to fix the gradient-penalty interpolation draws (tests); a trailing dict
  holds further dataset fields for Trainer.run (distill_embed_s / distill_embed_t of --do_encoder_distillation).
  One reference "step" (global_step) = one generator apply = ``n_critic`` runs (image_generation.py:640-652).
  Growing stages re-create ``alpha_grow`` every step, so they launch eagerly (a captured graph bakes alpha in).
  ``train_dir``: the reference's directory protocol (pggan_runner.py:100-160) over TF-format checkpoints
  (checkpoint.py) -- stage ``name`` trains in <train_dir>/<name>, is skipped when that directory already holds a
  checkpoint of >= its number of steps, resumes from a checkpoint it finds there, otherwise warm-starts from the
  previous stage's directory with ignore_missing_vars = is_growing, and saves model.ckpt-<global_step> at its end and,
  like slim.learning.train's Saver (--save_interval_secs, model_inheritor.py:75,1126: 600 s), every
  ``save_interval_secs`` seconds (or every ``save_interval_steps`` steps) inside a stage, keeping ``max_to_keep`` files:
  the final stage runs for 10 M steps, an interruption must not lose it."""
  import os
  import time
  from . import checkpoint as ckpt
  from .twingan import Trainer
  state = None
  history = []
  last_dir = None
  for name, hw, growing, bsz, steps in stage_schedule(start_hw, max_hw, hw_to_batch_size, num_images_per_resolution):
    if max_steps_per_stage is not None:
      steps = min(steps, max_steps_per_stage)
    cur_dir = os.path.join(train_dir, name) if train_dir else None
    found = ckpt.latest_checkpoint(cur_dir) if cur_dir else None
    if found is not None and int(found.rsplit('-', 1)[1]) >= steps:      # 'Skipping already trained model'
      history.append(dict(stage=name, hw=hw, is_growing=growing, batch_size=bsz, steps=0, warm_started=0, skipped=True))
      last_dir = cur_dir
      continue
    cfg = replace(base_cfg, hw=hw, is_growing=growing, alpha_grow=0.0)
    tr = Trainer(cfg, device=device, seed=seed, use_graph=use_graph and not growing)
    first = 0
    if found is not None:
      first = ckpt.restore(tr, found)
      loaded = list(tr.store.specs)
    elif train_dir:
      loaded = ckpt.init_from_checkpoint(tr, last_dir, ignore_missing_vars=growing, train_dir=cur_dir) if last_dir else []
    else:
      loaded = warm_start(tr, state) if state is not None else []
    last_save = time.time()
    for step in range(first, steps):
      if growing:
        tr.cfg.alpha_grow = alpha_grow(step, steps)
      for _ in range(cfg.n_critic):
        batch = batch_fn(hw, bsz)
        if isinstance(batch[-1], dict):      # further dataset fields (data.TwoDomainBatches: distill_embed_s / _t)
          tr.run(*batch[:-1], **batch[-1])
        else:
          tr.run(*batch)
      if cur_dir and step + 1 < steps and (
          (save_interval_steps and (step + 1) % save_interval_steps == 0) or
          (save_interval_secs and time.time() - last_save >= save_interval_secs)):
        ckpt.save(tr, cur_dir, global_step=step + 1, max_to_keep=max_to_keep)
        last_save = time.time()
    state = tr.store.state_dict(include_state=True)
    if cur_dir:
      ckpt.save(tr, cur_dir, global_step=steps, max_to_keep=max_to_keep)
      last_dir = cur_dir
    history.append(dict(stage=name, hw=hw, is_growing=growing, batch_size=bsz, steps=steps, warm_started=len(loaded)))
    if on_stage_end is not None:
      on_stage_end(name, tr)
    tr.close()
  return state, history
