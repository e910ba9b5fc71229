"""Times the flash-attention forward / backward entry points (incl. their packing kernels) at config 4's shape."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from twingan_amd import ops

n, ln, dk, dv = (int(a) for a in (sys.argv[1:5] if len(sys.argv) > 4 else (32, 4096, 8, 64)))
dt = torch.float16
q = torch.tanh(torch.randn(n, ln, dk, device='cuda')).to(dt).requires_grad_(True)
k = torch.tanh(torch.randn(n, ln, dk, device='cuda')).to(dt).requires_grad_(True)
v = torch.randn(n, ln, dv, device='cuda').to(dt).requires_grad_(True)
go = torch.randn(n, ln, dv, device='cuda').to(dt)


def timed(fn, reps=10):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(reps):
    fn()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) / reps * 1e3


fwd = timed(lambda: ops.flash_attention_fwd_raw(q.detach(), k.detach(), v.detach()))
o = ops.flash_attention(q, k, v)
bwd = timed(lambda: torch.autograd.grad(o, (q, k, v), go, retain_graph=True))
fl = 2.0 * n * ln * ln
print('flash n%d len%d dk%d dv%d: fwd %.1f us (%.0f TFLOP/s)  bwd %.1f us (%.0f TFLOP/s)' % (
    n, ln, dk, dv, fwd, fl * (dk + dv) / fwd * 1e-6, bwd, fl * 3 * (dk + dv) / bwd * 1e-6))
