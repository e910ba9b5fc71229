"""Measures how far the fp64 oracle's G gradients move when bf16 / fp16 storage rounding is inserted at the
product's storage points (conv outputs, layer outputs, pooled tensors, weights).  Test/diagnostic tool only."""
import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import torch_ref as R
MODE = {'f': torch.bfloat16, 'b': torch.bfloat16, 'w': torch.bfloat16}
class Rnd(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x): return x.to(MODE['f']).to(x.dtype) if MODE['f'] else x
  @staticmethod
  def backward(ctx, g): return g.to(MODE['b']).to(g.dtype) if MODE['b'] else g
rnd = Rnd.apply
def wr(w):
  return w + (w.detach().to(MODE['w']).to(w.dtype) - w.detach()) if MODE['w'] else w
_conv, _in, _pn, _lr, _pool = R.conv2d, R.instance_norm, R.pixel_norm, R.leaky_relu, R.avg_pool2
R.conv2d = lambda x, w, padding: rnd(_conv(rnd(x), wr(w), padding))
def ge_conv(P, scope, x, domain, cfg, k=3, padding='SAME', act=True, pixnorm=True):
  y = R.conv2d(x, P[scope + '/weights'], padding)
  y = _in(y, P[scope + '/InstanceNorm/gamma_' + domain], P[scope + '/InstanceNorm/beta_' + domain], cfg.in_eps)
  if act: y = _lr(y, cfg.lrelu)
  if pixnorm and cfg.do_pixel_norm: y = _pn(y, cfg.pn_eps)
  return rnd(y)
R.ge_conv = ge_conv
def d_conv(P, scope, x, cfg, k=3, padding='SAME'):
  y = _conv(rnd(x), wr(P[scope + '/weights']), padding) + P[scope + '/biases']
  return rnd(_lr(y, cfg.lrelu))
R.d_conv = d_conv
R.avg_pool2 = lambda x: rnd(_pool(x))

def grads(hw, max_ch, seed=2, batch=2):
  cfg = R.Config(hw=hw, max_ch=max_ch)
  P = R.init_params(cfg, seed=seed, dtype=torch.float64, std='he')
  P = {k: v.float().double() for k,v in P.items()}
  g = torch.Generator().manual_seed(1234+seed)
  s = torch.rand(batch,hw,hw,3,generator=g).to(torch.bfloat16).double(); t = torch.rand(batch,hw,hw,3,generator=g).to(torch.bfloat16).double()
  for v in P.values(): v.requires_grad_(True)
  gl,terms = R.generator_loss(P, s, t, cfg)
  return R.grads_of(gl, P, R.generator_var_names(P)), {k: float(v) for k,v in terms.items()}
def cmp(a,b):
  num = sum(float(((a[k]-b[k])**2).sum()) for k in b); den = sum(float((b[k]**2).sum()) for k in b)
  return (num/den)**.5
for hw,mc in ((32,32),):
  MODE.update(f=None,b=None,w=None); ref,_ = grads(hw,mc)
  for name, m in (('all bf16', dict(f=torch.bfloat16,b=torch.bfloat16,w=torch.bfloat16)),
                  ('fwd only', dict(f=torch.bfloat16,b=None,w=None)),
                  ('bwd only', dict(f=None,b=torch.bfloat16,w=None)),
                  ('w only', dict(f=None,b=None,w=torch.bfloat16)),
                  ('all fp16', dict(f=torch.float16,b=torch.float16,w=torch.float16))):
    MODE.update(m); g,_ = grads(hw,mc); print(hw,mc,name,'%.3e'%cmp(g,ref))
