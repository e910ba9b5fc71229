"""Where does the step's wall time go: D run / G run, graph vs eager, with / without domain streams."""
import sys, time
sys.path.insert(0, '.')
import torch
from twingan_amd import Config
from twingan_amd.twingan import Trainer
dev = 'cuda:0'
for graph in (True, False):
  for streams in (True, False):
    cfg = Config(hw=256, max_ch=256, domain_streams=streams)
    tr = Trainer(cfg, device=dev, seed=0, use_graph=graph)
    g = torch.Generator().manual_seed(1)
    s = torch.rand(16, 256, 256, 3, generator=g).to(dev).bfloat16()
    t = torch.rand(16, 256, 256, 3, generator=g).to(dev).bfloat16()
    for _ in range(8):
      tr.run(s, t)
    torch.cuda.synchronize()
    wall = {True: [], False: []}
    for i in range(8):
      is_g = tr.n_critic_counter % 2 == 0
      t0 = time.perf_counter()
      tr.run(s, t)
      torch.cuda.synchronize()
      wall[is_g].append((time.perf_counter() - t0) * 1e3)
    print('graph=%s streams=%s: G run %.2f ms, D run %.2f ms' % (graph, streams, min(wall[True]), min(wall[False])), flush=True)
    del tr
