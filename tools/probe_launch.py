"""Per-node cost of tiny kernels: eager launches vs hipGraph replay (single stream, dependent chain)."""
import ctypes, time, sys
sys.path.insert(0, '.')
import torch
from twingan_amd import _lib
from twingan_amd._lib import call
lib = _lib.load()
dev = 'cuda:0'
x = torch.zeros(1 << 20, dtype=torch.float32, device=dev)
y = torch.zeros(1 << 20, dtype=torch.float32, device=dev)
def body(n, numel):
  s = torch.cuda.current_stream().cuda_stream
  for i in range(n):
    call('tg_axpby', x.data_ptr(), 0, y.data_ptr(), numel, 1.0, 0.0, 0, s)
for numel in (64, 1 << 16, 1 << 20):
  N = 1000
  body(10, numel); torch.cuda.synchronize()
  t0 = time.perf_counter(); body(N, numel); torch.cuda.synchronize(); te = (time.perf_counter() - t0) / N * 1e6
  g = torch.cuda.CUDAGraph()
  st = torch.cuda.Stream()
  with torch.cuda.stream(st):
    with torch.cuda.graph(g, stream=st):
      body(N, numel)
  g.replay(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(5): g.replay()
  torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / (5 * N) * 1e6
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record(); body(N, numel); e1.record(); torch.cuda.synchronize()
  print('numel %8d: eager %.2f us/launch (wall), %.2f us (events); graph replay %.2f us/node' % (numel, te, e0.elapsed_time(e1) / N * 1e3, tg))
