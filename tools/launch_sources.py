"""Which Python line launches each kernel of a training step?

One EAGER G+D step of the headline configuration under torch.profiler (with_stack): every device kernel is attributed to
the operator that launched it (torch's own, or the autograd Function whose forward / backward called into
libtwingan_hip.so) and to the innermost frame of that operator's Python stack that lies in twingan_amd/.

  python tools/launch_sources.py [steps=2] [glue]      ->  stdout table: launches per step, total us, kernel, op, call site
  (glue: only torch's own kernels)
"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from twingan_amd import Config      # noqa: E402
from twingan_amd.twingan import Trainer      # noqa: E402


def main():
  steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
  dev = 'cuda:0'
  cfg = Config(hw=256, max_ch=256)
  tr = Trainer(cfg, device=dev, seed=0, use_graph=False)
  g = torch.Generator().manual_seed(1)
  s = torch.rand(16, 256, 256, 3, generator=g).to(dev).bfloat16()
  t = torch.rand(16, 256, 256, 3, generator=g).to(dev).bfloat16()
  for _ in range(4):
    tr.run(s, t)
  torch.cuda.synchronize()
  from torch.profiler import profile, ProfilerActivity
  with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
               experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    for _ in range(2 * steps):      # a "step" of the bench = one D run + one G run
      tr.run(s, t)
    torch.cuda.synchronize()
  agg = collections.defaultdict(lambda: [0, 0.0])
  for e in prof.events():
    if not e.kernels:
      continue
    # the event a kernel hangs on is the runtime launch call; its nearest ancestor with a Python stack names the site
    p, op = e, None
    while p is not None and not p.stack:
      p = p.cpu_parent
    frames = [str(f) for f in (p.stack if p is not None else [])]
    q = e
    while q is not None and (q.name.startswith('hip') or q.name.startswith('cuda')):
      q = q.cpu_parent
    op = q.name if q is not None else '?'
    site = '?'
    for f in frames:
      if 'twingan_amd/' in f and '_lib.py' not in f:
        site = f.split('twingan_amd/')[-1]
        break
    else:
      site = frames[0][-60:] if frames else '?'
    for k in e.kernels:
      key = (k.name.split('(')[0][-60:], op, site)
      agg[key][0] += 1
      agg[key][1] += k.duration
  rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
  only_glue = len(sys.argv) > 2 and sys.argv[2] == 'glue'
  print('launches/step   us/step  kernel | op | site')
  for (k, op, site), (n, us) in rows:
    if only_glue and not ('at::native' in k or 'rocclr' in k or 'elementwise' in k):
      continue
    print('%8.1f %10.1f  %s | %s | %s' % (n / steps, us / steps, k, op, site))


if __name__ == '__main__':
  main()
