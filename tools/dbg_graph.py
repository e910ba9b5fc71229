import sys, torch
sys.path.insert(0, '.')
from twingan_amd import Config
from twingan_amd.twingan import Trainer
cfg = Config(hw=32, max_ch=16, precision='fp32', loss_architecture='wgan')
g = torch.Generator().manual_seed(9)
s = torch.rand(2, 32, 32, 3, generator=g).to('cuda:0'); t = torch.rand(2, 32, 32, 3, generator=g).to('cuda:0')
a = Trainer(cfg, device='cuda:0', seed=4)
b = Trainer(cfg, device='cuda:0', seed=4, use_graph=True)
for i in range(8):
  la, ta = a.run(s, t)
for i in range(4):
  lb, tb = b.run(s, t)
  print('graph', i + 4, float(lb), {k: round(float(v), 5) for k, v in tb.items()})
mode = sys.argv[1] if len(sys.argv) > 1 else 'eager'
if mode == 'eager':
  la, ta = a.run(s, t); print('eager a', float(la))
elif mode == 'alloc':
  xs = [torch.empty(1 << 20, device='cuda:0').normal_() for _ in range(50)]; del xs
elif mode == 'sync':
  torch.cuda.synchronize()
elif mode == 'statedict':
  sd = b.store.state_dict(); print(len(sd))
for i in range(4):
  lb, tb = b.run(s, t)
  print('graph', i + 8, float(lb), {k: float(v) for k, v in tb.items()})
