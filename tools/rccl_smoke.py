"""One-GPU check of the data-parallel schedule the first real N > 1 run will take: a single-rank "nccl" (= RCCL) process
group alive in the process, the segmented backward captured as per-segment hipGraphs (thread_local capture mode next to the
communicator's watchdog), one RCCL all-reduce per segment range enqueued between the segment graphs and waited for before the
apply graph -- replacing deployment/model_deploy.py:265-268,473-503.  A one-rank sum is the identity, so the trajectory
must equal the unsegmented single-graph trainer's: bit for bit in fp32 (the exact path sums in a fixed order), to 16-bit
run-to-run noise otherwise.  Prints one JSON line; tests/test_gpu_model.py::test_rccl_segmented_capture_one_rank runs it in a
child process (the process group must not outlive the check)."""
import argparse
import json
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--precision', default='bf16', choices=['fp32', 'bf16', 'fp16'])
  ap.add_argument('--hw', type=int, default=int(os.environ.get('HW', 64)))
  ap.add_argument('--max-ch', type=int, default=int(os.environ.get('MC', 32)))
  ap.add_argument('--batch', type=int, default=int(os.environ.get('B', 4)))
  ap.add_argument('--steps', type=int, default=3, help='G+D steps (two runs each)')
  ap.add_argument('--loss', default='wgan_gp')
  ap.add_argument('--time', type=int, default=0, help='also time this many runs of the segmented trainer')
  a = ap.parse_args()
  os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
  if 'MASTER_PORT' not in os.environ:
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
    sk.close()
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  torch.cuda.set_device(0)
  dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
  from twingan_amd import Config
  from twingan_amd.twingan import Trainer
  cfg = Config(hw=a.hw, max_ch=a.max_ch, precision=a.precision, loss_architecture=a.loss, overlap_cut_hw=min(32, a.hw // 2))
  dt = {'fp32': torch.float32, 'bf16': torch.bfloat16, 'fp16': torch.float16}[a.precision]
  g = torch.Generator().manual_seed(3)
  s = torch.rand(a.batch, a.hw, a.hw, 3, generator=g).cuda().to(dt)
  t = torch.rand(a.batch, a.hw, a.hw, 3, generator=g).cuda().to(dt)
  ref = Trainer(cfg, device='cuda:0', seed=1, use_graph=True)                      # one graph per step kind, no collective
  tr = Trainer(cfg, device='cuda:0', seed=1, use_graph=True, world_size=1, overlap=True)
  tr.reducer.always = True                                                         # issue the one-rank all-reduces anyway
  assert tr.reducer.active and tr._nseg('g') == 3 and tr._nseg('d') == 2, (tr._nseg('g'), tr._nseg('d'))
  for _ in range(2 * a.steps):
    ref.run(s, t)
    tr.run(s, t)
  torch.cuda.synchronize()
  assert tr.use_graph, tr.graph_fallback_reason
  st = tr.reducer.stats()
  sa, sb = ref.store.state_dict(), tr.store.state_dict()
  unequal = [k for k in sa if not torch.equal(sa[k], sb[k])]
  num = sum(float(((sa[k] - sb[k]).double() ** 2).sum()) for k in sa)
  den = sum(float((sa[k].double() ** 2).sum()) for k in sa)
  out = dict(world=dist.get_world_size(), backend=dist.get_backend(), segments={'g': tr._nseg('g'), 'd': tr._nseg('d')},
             capture_note=tr.capture_note, use_graph=bool(tr.use_graph), collectives=st['collectives'],
             allreduce_bytes=st['allreduce_bytes'], finishes=st['finishes'], exposed_max_ms=round(st['exposed_max_ms'], 4),
             tensors=len(sa), tensors_not_bit_equal=len(unequal), params_rel_l2=(num / den) ** 0.5)
  if a.time:
    t0 = time.perf_counter()
    for _ in range(a.time):
      tr.run(s, t)
    torch.cuda.synchronize()
    out['ms_per_run'] = round(1e3 * (time.perf_counter() - t0) / a.time, 3)
  print(json.dumps(out))
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
