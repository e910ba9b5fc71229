"""Is the fp32 (exact-parity) trajectory bit-reproducible?  N pairs of trainers on the same data (the test's setup); prints
which tensors / optimiser slots differ per pair."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from twingan_amd import Config                    # noqa: E402
from twingan_amd.twingan import Trainer           # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
norm = sys.argv[2] if len(sys.argv) > 2 else 'instance_norm'
cfg = Config(hw=32, max_ch=16, precision='fp32', generator_norm_type=norm)
g = torch.Generator().manual_seed(3)
data = [(torch.rand(3, 32, 32, 3, generator=g), torch.rand(3, 32, 32, 3, generator=g), torch.rand(3, generator=g),
         torch.rand(3, generator=g)) for _ in range(4)]
ends = []
for rep in range(reps):
  tr = Trainer(cfg, device='cuda:0', seed=4)
  for s, t, a_s, a_t in data:
    tr.run(s.cuda(), t.cuda(), a_s.cuda(), a_t.cuda())
  torch.cuda.synchronize()
  ends.append((tr.store.state_dict(include_state=True), {k: (m, v) for k, (m, v) in tr.store.adam_dict().items()}))
  tr.close()
tag = ' '.join('%s=%s' % (k, v) for k, v in sorted(os.environ.items()) if k.startswith('TG_'))
for i in range(1, reps):
  (pa, sa), (pb, sb) = ends[0], ends[i]
  badp = [k for k in pa if not torch.equal(pa[k], pb[k])]
  bads = [k for k in sa if not (torch.equal(sa[k][0], sb[k][0]) and torch.equal(sa[k][1], sb[k][1]))]
  print('[%s] run %d vs 0: params differing %s, slots differing %s' % (tag or 'default', i, badp[:4], bads[:4]))
