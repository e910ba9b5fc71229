import sys, numpy as np, torch
sys.path.insert(0, '.')
from twingan_amd import Config
from twingan_amd import twingan as T
g = dict(np.load('tests/golden/twingan_hw64_c8.npz'))
cfg = Config(precision='fp32', hw=64, max_ch=8)
for rep in range(40):
  tr = T.Trainer(cfg, device='cuda:0', seed=0)
  tr.store.load_state_dict({k[len('param/'):]: torch.from_numpy(v).float() for k, v in g.items() if k.startswith('param/')})
  dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().cuda().contiguous()
  s, t = dev(g['in/sources']), dev(g['in/targets'])
  tr.store.zero_grad('g'); tr._set_requires_grad(g=True, d=False)
  loss, terms = T.generator_loss(tr.P, s, t, cfg)
  loss.backward()
  gd = tr.store.grad_dict()
  errs = []
  num = den = 0
  for k in tr.store.names('g'):
    a = gd[k].double().cpu().numpy(); b = g['grad/' + k]
    num += ((a - b) ** 2).sum(); den += (b ** 2).sum()
    errs.append((np.linalg.norm(a - b), np.linalg.norm(b), k))
  print('rep', rep, 'total rel', (num / den) ** .5, 'loss', loss.item(), float(g['loss/g_total']))
  errs.sort(reverse=True); errs = errs if (num / den) ** .5 > 5e-3 else []
  for e in errs[:6]:
    print('   abs err %.4f  norm %.4f  %s' % e)
