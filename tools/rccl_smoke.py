"""One-GPU RCCL smoke test of the data-parallel schedule: a single-rank "nccl" (= RCCL) process group, the segmented
backward captured as hipGraphs, one RCCL all-reduce per segment range enqueued between the segment graphs and waited
for before the apply graph.  A one-rank sum is the identity, so the trajectory must equal the plain single-clone one.
(The multi-GPU numbers are the driver's; this only proves the RCCL calls, streams and captures compose.)"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
  os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
  os.environ.setdefault('MASTER_PORT', '29517')
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  torch.cuda.set_device(0)
  dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
  from twingan_amd import Config
  from twingan_amd.twingan import Trainer
  hw, mc, b = int(os.environ.get('HW', 64)), int(os.environ.get('MC', 32)), int(os.environ.get('B', 4))
  cfg = Config(hw=hw, max_ch=mc, precision='bf16', loss_architecture='wgan', overlap_cut_hw=min(32, hw // 2))
  g = torch.Generator().manual_seed(3)
  s = torch.rand(b, hw, hw, 3, generator=g).cuda().to(torch.bfloat16)
  t = torch.rand(b, hw, hw, 3, generator=g).cuda().to(torch.bfloat16)
  ref = Trainer(cfg, device='cuda:0', seed=1, use_graph=True)
  tr = Trainer(cfg, device='cuda:0', seed=1, use_graph=True, overlap=True)
  tr.reducer.always = True
  assert tr._nseg('g') == 3 and tr._nseg('d') == 2
  for _ in range(6):
    ref.run(s, t)
    tr.run(s, t)
  torch.cuda.synchronize()
  assert tr.use_graph, tr.graph_fallback_reason
  for grp in ('g', 'd'):
    a, c = ref.store.flat[grp], tr.store.flat[grp].new_empty(0)
    # same names, different layout order is impossible here (same cfg): compare name by name
  sa, sb = ref.store.state_dict(), tr.store.state_dict()
  num = sum(float(((sa[k] - sb[k]).double() ** 2).sum()) for k in sa)
  den = sum(float((sa[k].double() ** 2).sum()) for k in sa)
  rel = (num / den) ** 0.5
  t0 = time.perf_counter()
  for _ in range(10):
    tr.run(s, t)
  torch.cuda.synchronize()
  ms = 1e3 * (time.perf_counter() - t0) / 10
  print('rccl_smoke ok: world %d backend %s, segments g=%d d=%d, params vs single clone rel-L2 %.2e, %.2f ms/run'
        % (dist.get_world_size(), dist.get_backend(), tr._nseg('g'), tr._nseg('d'), rel, ms))
  assert rel < 2e-2, rel
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
