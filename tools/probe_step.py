"""Where does the step's wall time go: CPU time inside graph replay vs GPU completion, with / without domain streams."""
import sys, time
sys.path.insert(0, '.')
import torch
from twingan_amd import Config
from twingan_amd.twingan import Trainer
dev = 'cuda:0'
for streams in (True, False):
  cfg = Config(hw=256, max_ch=256, domain_streams=streams)
  tr = Trainer(cfg, device=dev, seed=0, use_graph=True)
  g = torch.Generator().manual_seed(1)
  s = torch.rand(16, 256, 256, 3, generator=g).to(dev).bfloat16()
  t = torch.rand(16, 256, 256, 3, generator=g).to(dev).bfloat16()
  for _ in range(8):
    tr.run(s, t)
  torch.cuda.synchronize()
  cpu = []; wall = []
  for _ in range(8):
    t0 = time.perf_counter()
    tr.run(s, t)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    cpu.append((t1 - t0) * 1e3); wall.append((t2 - t0) * 1e3)
  print('domain_streams=%s: per run (D then G alternating) cpu-in-run ms %s | wall ms %s' % (
      streams, ['%.2f' % c for c in cpu], ['%.2f' % w for w in wall]))
  t0 = time.perf_counter()
  for _ in range(12):
    tr.run(s, t)
  torch.cuda.synchronize()
  print('  pipelined: %.2f ms per G+D step' % ((time.perf_counter() - t0) / 6 * 1e3))
  del tr
