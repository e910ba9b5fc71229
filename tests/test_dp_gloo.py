"""world_size-2 data-parallel checks on CPU (gloo): the clone reduction of
deployment/model_deploy.py:242-315,473-503 as implemented by twingan_amd/dp.py over the flat gradient
buffers of twingan_amd/params.py.  The oracle supplies each clone's gradients (checker only); the expected result
is the REFERENCE's: tests/golden/clones2_hw16_c8.npz holds what model_deploy.create_clones / optimize_clones produced
for two clones of GanModel._clone_fn (oracle/ref_runner.run_clones, tools/make_golden.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'clones2_hw16_c8.npz')


def _clone_grads(rank, world, hw=16, max_ch=8):
  """fp64 oracle gradients of clone ``rank``'s D loss on the fixture's batch for that clone, scaled by the factor
  the product applies to a clone's loss (loss_scale_for_clones)."""
  import numpy as np
  from oracle import torch_ref as R
  from twingan_amd.dp import loss_scale_for_clones
  g = np.load(GOLD)
  rcfg = R.Config(hw=hw, max_ch=max_ch)
  P = {k[len('param/'):]: torch.from_numpy(g[k]) for k in g.files if k.startswith('param/')}
  s, t = torch.from_numpy(g['clone%d/sources' % rank]), torch.from_numpy(g['clone%d/targets' % rank])
  a_s = torch.from_numpy(g['clone%d/gp_alpha_s' % rank]).reshape(-1, 1, 1, 1)
  a_t = torch.from_numpy(g['clone%d/gp_alpha_t' % rank]).reshape(-1, 1, 1, 1)
  names = R.discriminator_var_names(P)
  for k in names:
    P[k].requires_grad_(True)
  loss, _ = R.discriminator_loss(P, s, t, rcfg, a_s, a_t)
  grads = torch.autograd.grad(loss * loss_scale_for_clones(1.0, world), [P[k] for k in names])
  return dict(zip(names, grads))


def _worker(rank, world, port, q):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from twingan_amd import Config
    from twingan_amd.dp import GradReducer, loss_scale_for_clones
    from twingan_amd.params import ParamStore, declare_twingan
    cfg = Config(hw=16, max_ch=8, precision='fp32')
    store = declare_twingan(ParamStore('cpu'), cfg).build(seed=0)
    assert loss_scale_for_clones(1.0, world) == 1.0 / world
    mine = _clone_grads(rank, world)
    store.zero_grad('d')
    for k, gk in mine.items():
      s = store.specs[k]
      store._logical(store.P[k].grad, s).copy_(gk.float())
    red = GradReducer(world, None, n_buckets=3)
    nb = red.start(store.grad['d'])
    assert nb >= 2
    red.finish()
    out = {k: v.double() for k, v in store.grad_dict().items() if store.specs[k]['group'] == 'd'}
    if rank == 0:
      q.put({k: v.numpy() for k, v in out.items()})
    dist.barrier()
  finally:
    dist.destroy_process_group()


def test_grad_reducer_sums_clone_gradients_world2():
  world = 2
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
  for p in procs:
    p.start()
  got = q.get(timeout=300)
  for p in procs:
    p.join(timeout=300)
    assert p.exitcode == 0
  # expected: what the reference's optimize_clones returned for these two clones (tf.add_n of grad(loss_r / 2))
  import numpy as np
  ref = np.load(GOLD)
  want = {k[len('grad_d/'):]: ref[k] for k in ref.files if k.startswith('grad_d/')}
  assert set(got) == set(want)
  for k in want:
    err = abs(got[k] - want[k]).max()
    assert err <= 1e-6 * max(1.0, abs(want[k]).max()), (k, err)      # the product's buffers are fp32


def test_bucket_bounds_cover_buffer():
  from twingan_amd.dp import bucket_bounds
  for numel in (64, 1000, 4096, 1 << 20):
    for nb in (1, 2, 3, 7):
      b = bucket_bounds(numel, nb)
      assert b[0][0] == 0 and b[-1][1] == numel and len(b) <= nb
      for (lo, hi), (lo2, _) in zip(b, b[1:]):
        assert hi == lo2 and lo % 64 == 0 and hi > lo


def test_single_clone_reducer_is_noop():
  from twingan_amd.dp import GradReducer
  g = torch.arange(128, dtype=torch.float32)
  r = GradReducer(1)
  assert r.start(g) == 0
  r.finish()
  assert torch.equal(g, torch.arange(128, dtype=torch.float32))
