"""Times the elementwise backward entry points whose tail is a burst of same-address atomics (raw C-ABI calls, HIP events)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from twingan_amd import ops


def timed(fn, reps=20):
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


dt = torch.bfloat16
tag = 'lrelu_blocks=%s' % os.environ.get('TG_LRELU_BWD_BLOCKS', 'default')
for n, c, hw in ((16, 32, 256), (48, 32, 256), (16, 64, 128), (48, 64, 128), (16, 128, 64), (48, 256, 32), (16, 256, 16)):
  z = torch.randn(n, hw, hw, c, device='cuda').to(dt)
  gzp = torch.randn(n, hw // 2, hw // 2, c, device='cuda').to(dt)
  bias = torch.zeros(c, device='cuda')
  us = timed(lambda: ops.lrelu_pool_bwd(None, gzp, z, 0.2, bias, True))
  print('%s lrelu_pool_bwd n%d c%d hw%d: %.1f us (%.2f TB/s)' % (tag, n, c, hw, us, 2.25 * z.numel() * 2 / us * 1e-6))
for ca, cb, px in ((3, 16, 1 << 20), (3, 16, 2 << 20), (3, 16, 3 << 20), (16, 3, 4 << 20)):
  a = torch.randn(px, ca, device='cuda').to(dt)
  b = torch.randn(px, cb, device='cuda').to(dt)
  us = timed(lambda: ops.PointwiseWgradFn.apply(a, b))
  print('pw_wgrad c%d>%d px%d: %.1f us (%.2f TB/s)' % (ca, cb, px, us, (a.numel() + b.numel()) * 2 / us * 1e-6))
