"""In-process A/B of launch-heuristic settings: one Trainer (= one set of captured hipGraphs) per TG_TUNE_* setting,
timed alternately on the same box.  usage: python tools/ab_env.py "" "TG_TUNE_X=1" "TG_TUNE_X=2,TG_TUNE_Y=1" ..."""
import os, sys, time
sys.path.insert(0, '.')
import torch
from twingan_amd import Config
from twingan_amd.twingan import Trainer

dev = 'cuda:0'
variants = sys.argv[1:] or ['']
g = torch.Generator().manual_seed(1)
s = torch.rand(16, 256, 256, 3, generator=g).to(dev).bfloat16()
t = torch.rand(16, 256, 256, 3, generator=g).to(dev).bfloat16()
trainers = []
for v in variants:
  keys = []
  for kv in filter(None, v.split(',')):
    k, val = kv.split('=')
    os.environ[k] = val
    keys.append(k)
  tr = Trainer(Config(hw=256, max_ch=256), device=dev, seed=0, use_graph=True)
  for _ in range(6):
    tr.run(s, t)
  torch.cuda.synchronize()
  for k in keys:
    del os.environ[k]
  trainers.append(tr)
best = [1e9] * len(variants)
tot = [0.0] * len(variants)
ROUNDS, STEPS = int(os.environ.get("AB_ROUNDS", 5)), 10
for r in range(ROUNDS):
  for i, tr in enumerate(trainers):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2 * STEPS):      # G run + D run = one step
      tr.run(s, t)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / STEPS
    best[i] = min(best[i], ms)
    tot[i] += ms
for i, v in enumerate(variants):
  print('%-50s best %.3f ms  mean %.3f ms' % (v or '(default)', best[i], tot[i] / ROUNDS), flush=True)
