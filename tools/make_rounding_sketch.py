"""Storage-rounding sensitivity of the HEADLINE graph at full size -> tests/golden/full_hw256_c256_rounding.json.

tests/golden/full_hw256_c256.json holds what the reference's own code computed for the 256 x 256 / 256-channel
configuration (losses, image probes, the NORM of every gradient: 71 MB of gradients are not stored).  A 16-bit
implementation cannot be held to those gradients by a fixed number: bf16 storage rounding alone moves the float64
gradients of this random-weight graph by tens of percent (oracle/rounding.py).  This script measures by HOW MUCH, on the
fixture's own weights and inputs, and stores what the GPU test needs to turn that into an assertion:

  * ``rounded_rel_l2[group]``: aggregate rel-L2 between the float64 oracle's gradients and the gradients of the same
    oracle with bf16 rounding inserted at the kernels' storage points (inputs, conv inputs / outputs, layer outputs,
    pooled tensors, weight packs; forward values and the gradients flowing back), per optimiser group;
  * ``exact_sketch[name]``: K random +-1 projections of the float64 gradient of every variable (seeded per variable):
    E[(r . d)^2] = |d|^2, so the test estimates |g_hip - g_float64|^2 of a whole group from K x (number of variables)
    projections of the kernels' gradients without the 71 MB.

tests/test_gpu_model.py::test_full_width_stage_hits_the_reference[bf16] then asserts, per group,
  rel-L2(kernels vs float64) <= 1.5 x rounded_rel_l2 + 0.02.

Oracle only (oracle/torch_ref.py, pinned to the reference's code at 1e-13 on this configuration); ~30 min and ~25 GB on 8
cores:  python tools/make_rounding_sketch.py
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import rounding, torch_ref as R      # noqa: E402

K = 16
SKETCH_SEED = 7001


def sketch_vectors(index, numel, device='cpu'):
  """The K +-1 vectors of variable number ``index`` (in sorted-name order): [K, numel] float32."""
  g = torch.Generator().manual_seed(SKETCH_SEED + index)
  return (torch.randint(0, 2, (K, numel), generator=g, dtype=torch.int8).to(device).float() * 2.0 - 1.0)


def main():
  hw_arg = int(sys.argv[sys.argv.index('--hw') + 1]) if '--hw' in sys.argv else 256      # --hw 64 | 128: configs[1] / configs[2]
  base = 'full_hw%d_c256' % hw_arg
  if '--fixture' in sys.argv:      # --fixture full_hw256_c256_sn_att --dtype fp16: BASELINE configs[4] (its storage type is fp16)
    base = sys.argv[sys.argv.index('--fixture') + 1]
  sdt_name = sys.argv[sys.argv.index('--dtype') + 1] if '--dtype' in sys.argv else 'bf16'
  sdt = {'bf16': torch.bfloat16, 'fp16': torch.float16}[sdt_name]
  with open(os.path.join(ROOT, 'tests', 'golden', base + '.json')) as fh:
    fix = json.load(fh)
  hw, batch = fix['config']['hw'], fix['batch']
  cfg = R.Config(**fix['config'])
  P = {k: v.float().double() for k, v in R.init_params(cfg, seed=fix['param_seed'], dtype=torch.float64, std='he').items()}
  g = torch.Generator().manual_seed(fix['input_seed'])
  s = torch.rand(batch, hw, hw, 3, generator=g).double()
  t = torch.rand(batch, hw, hw, 3, generator=g).double()
  a_s = torch.tensor(fix['gp_alpha_s'], dtype=torch.float64).reshape(-1, 1, 1, 1)
  a_t = torch.tensor(fix['gp_alpha_t'], dtype=torch.float64).reshape(-1, 1, 1, 1)
  sr, tr_ = s.to(sdt).double(), t.to(sdt).double()      # what the 16-bit test feeds the kernels
  sn_state = {k: v.float().double() for k, v in R.init_sn_state(P, seed=fix['param_seed'] + 1).items()} if cfg.spectral_norm else None
  groups = {'g': R.generator_var_names(P), 'd': R.discriminator_var_names(P)}
  order = sorted(P)
  out = dict(K=K, sketch_seed=SKETCH_SEED, dtype=sdt_name, order=order, exact_sketch={}, rounded_rel_l2={}, exact_norm_check={})

  def grads(group, rounded):
    Q = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    x, y = (sr, tr_) if rounded else (s, t)
    if sn_state is not None:      # every run starts from the fixture's u (the runs do not assign it: no end_run)
      cfg.sn_state, cfg.sn_cache = {k: v.clone() for k, v in sn_state.items()}, {}
    t0 = time.time()
    if group == 'g':
      loss, _ = R.generator_loss(Q, x, y, cfg)
    else:
      loss, _ = R.discriminator_loss(Q, x, y, cfg, a_s, a_t)
    gr = R.grads_of(loss, Q, groups[group])
    print('  %s %s: loss %.6f, %.0f s' % (group, 'rounded' if rounded else 'exact', float(loss), time.time() - t0), flush=True)
    return {k: v.detach() for k, v in gr.items()}

  for group in ('g', 'd'):
    exact = grads(group, False)
    worst = max(abs(float(exact[k].norm()) - fix['grad_norm'][k]) / max(fix['grad_norm'][k], 1e-12) for k in exact
                if fix['grad_norm'][k] > 1e-6 * max(fix['grad_norm'].values()))
    out['exact_norm_check'][group] = worst      # the oracle reproduces the reference's gradient norms of the fixture
    print('  %s: worst gradient-norm deviation from the reference fixture %.2e' % (group, worst), flush=True)
    assert worst < 1e-6, worst
    for k, v in exact.items():
      out['exact_sketch'][k] = (sketch_vectors(order.index(k), v.numel()).double() @ v.reshape(-1)).tolist()
    with rounding.storage_rounding(sdt):
      rnd = grads(group, True)
    num = sum(float(((rnd[k] - exact[k]) ** 2).sum()) for k in exact)
    den = sum(float((exact[k] ** 2).sum()) for k in exact)
    out['rounded_rel_l2'][group] = (num / den) ** 0.5
    # how well K projections per variable estimate that number (the estimator the GPU test uses on the kernels' gradients)
    est_num = sum(float(((sketch_vectors(order.index(k), v.numel()).double() @ (rnd[k] - exact[k]).reshape(-1)) ** 2).sum())
                  for k, v in exact.items())
    est_den = sum(float((torch.tensor(out['exact_sketch'][k]) ** 2).sum()) for k in exact)
    out.setdefault('rounded_rel_l2_from_sketch', {})[group] = (est_num / est_den) ** 0.5
    print('  %s: storage rounding moves the gradients by rel-L2 %.4f (sketch estimate %.4f)'
          % (group, out['rounded_rel_l2'][group], out['rounded_rel_l2_from_sketch'][group]), flush=True)
    del exact, rnd
  with open(os.path.join(ROOT, 'tests', 'golden', base + '_rounding.json'), 'w') as fh:
    json.dump(out, fh)
  print('written')


if __name__ == '__main__':
  main()
