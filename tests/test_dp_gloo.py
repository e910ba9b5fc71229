"""world_size-2 data-parallel checks on CPU (gloo): the clone reduction of
deployment/model_deploy.py:242-315,473-503 as implemented by twingan_amd/dp.py over the flat gradient
buffers of twingan_amd/params.py.  The oracle supplies each clone's gradients (checker only)."""
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


def _clone_grads(rank, world, hw=16, max_ch=8, batch=2):
  """fp64 oracle gradients of clone ``rank``'s D loss, already scaled by 1/world."""
  from oracle import torch_ref as R
  rcfg = R.Config(hw=hw, max_ch=max_ch)
  P = R.init_params(rcfg, seed=0, dtype=torch.float64, std='he')
  g = torch.Generator().manual_seed(100 + rank)
  s = torch.rand(batch, hw, hw, 3, generator=g, dtype=torch.float64)
  t = torch.rand(batch, hw, hw, 3, generator=g, dtype=torch.float64)
  a = torch.rand(batch, 1, 1, 1, generator=g, dtype=torch.float64)
  names = R.discriminator_var_names(P)
  for k in names:
    P[k].requires_grad_(True)
  loss, _ = R.discriminator_loss(P, s, t, rcfg, a, a)
  grads = torch.autograd.grad(loss / world, [P[k] for k in names])
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
  # expected: sum over clones of grad(loss_r / world) == grad of the mean clone loss
  g0, g1 = _clone_grads(0, world), _clone_grads(1, world)
  assert set(got) == set(g0)
  for k in g0:
    want = (g0[k] + g1[k]).numpy()
    err = abs(got[k] - want).max()
    assert err <= 1e-6 * max(1.0, abs(want).max()), (k, err)


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
