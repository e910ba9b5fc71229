"""Streaming-kernel microbenchmark: the element-wise / reduction passes of the step at their real shapes.
usage: python tools/ebench.py [filter]   -> us per launch and algorithmic GB/s (same byte model as bench.py)"""
import sys, time
sys.path.insert(0, '.')
import torch
from twingan_amd import ops
from twingan_amd import _lib

dev = 'cuda:0'
flt = sys.argv[1] if len(sys.argv) > 1 else ''
SHAPES = [(48, 256, 32), (48, 256, 16), (64, 256, 16), (32, 256, 16), (16, 256, 16), (48, 128, 64), (64, 128, 32), (32, 64, 128), (32, 16, 256)]


def timeit(f, iters=20):
  f(); f()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters):
    f()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e3


def report(name, shape, us, nbytes):
  print('%-18s n%-3d hw%-4d c%-4d | %8.1f us %8.0f GB/s' % (name, shape[0], shape[1], shape[2], us, nbytes / us * 1e-3), flush=True)


with torch.no_grad():
  for (n, hw, c) in SHAPES:
    z = torch.randn(n, hw, hw, c, device=dev).bfloat16()
    gz = torch.randn(n, hw, hw, c, device=dev).bfloat16()
    gzp = torch.randn(n, hw // 2, hw // 2, c, device=dev).bfloat16()
    bias = torch.zeros(c, device=dev)
    gamma = torch.ones(c, device=dev); beta = torch.zeros(c, device=dev)
    nb = z.numel() * 2
    if 'lrelu_pool_bwd'.startswith(flt) or flt in 'lrelu_pool_bwd':
      sink = torch.zeros(c, device=dev)
      ops.GradSink.register(bias, sink)
      report('lrelu_pool_bwd', (n, hw, c), timeit(lambda: ops.lrelu_pool_bwd(gz, gzp, z, 0.2, bias, True)), int(3.25 * nb))
      report('lrelu_bwd_bias', (n, hw, c), timeit(lambda: ops.lrelu_pool_bwd(gz, None, z, 0.2, bias, True)), 3 * nb)
      ops.GradSink.clear()
    if flt in 'lrelu_bwd':
      report('lrelu_bwd', (n, hw, c), timeit(lambda: ops.lrelu_bwd_raw(gz, z, 0.2)), 3 * nb)
    if flt in 'in_stats' or flt in 'norm_act_fwd' or flt in 'norm_act_bwd':
      class Ctx:      # stand-in for the autograd ctx of NormActFn
        def save_for_backward(self, *a): self.saved_tensors = a
      ctx = Ctx()
      flags = ops.NF_LRELU | ops.NF_PIXNORM
      f_fwd = lambda: ops._norm_act_forward(ctx, z, gamma, beta, None, None, n, flags, 1e-6, 1e-6, 0.2)
      report('stats+norm_act_fwd', (n, hw, c), timeit(f_fwd), 3 * nb)
      f_fwd()
      ctx.split, ctx.flags, ctx.alpha = n, flags, 0.2
      with ops.no_param_grads():
        report('norm_act_bwd', (n, hw, c), timeit(lambda: ops._norm_act_backward(ctx, gz)), 6 * nb)
    if flt in 'pool_fwd':
      report('pool_fwd', (n, hw, c), timeit(lambda: ops.Pool2Fn.apply(z, 0.25)), int(1.25 * nb))
    del z, gz, gzp
