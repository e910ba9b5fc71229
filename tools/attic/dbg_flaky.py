import sys, numpy as np, torch
sys.path.insert(0, '.')
from twingan_amd import Config
from twingan_amd import twingan as T
g = dict(np.load('tests/golden/twingan_hw64_c8.npz'))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().cuda().contiguous()
s, t = dev(g['in/sources']), dev(g['in/targets'])
for name, kw in (('base', {}), ('no_content', dict(l_content_weight=0.0)), ('no_unet', dict(use_unet=False)),
                 ('no_cycgan', dict(do_l_cyc_gan=False)), ('no_gan', dict(gan_weight=0.0))):
  cfg = Config(precision='fp32', hw=64, max_ch=8, **kw)
  tr = T.Trainer(cfg, device='cuda:0', seed=0)
  sd = {k[len('param/'):]: torch.from_numpy(v).float() for k, v in g.items() if k.startswith('param/')}
  if not cfg.use_unet:
    sd = {k: v for k, v in sd.items() if k in tr.store.specs and tuple(v.shape) == tr.store.specs[k]['shape']}
  tr.store.load_state_dict(sd, strict=False)
  ref = None; devs = []
  for rep in range(30):
    tr.store.zero_grad('g'); tr._set_requires_grad(g=True, d=False)
    loss, terms = T.generator_loss(tr.P, s, t, cfg)
    loss.backward()
    gr = tr.store.grad['g'].clone()
    if ref is None: ref = gr
    devs.append(float((gr - ref).norm() / ref.norm()))
  print(name, 'outliers>5e-3:', sum(d > 5e-3 for d in devs), 'max %.4f' % max(devs), ['%.4f' % d for d in devs if d > 5e-3][:4], flush=True)
