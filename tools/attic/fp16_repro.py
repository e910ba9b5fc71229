"""Root cause of the round-2 red test (test_fp16_training_steps_with_loss_scale_and_graph, e_graph 3.26e-3 > 1e-6):
is a 16-bit trajectory reproducible at all?  For each (precision, deterministic mode) it trains N eager and N graph
trainers on the same data from the same seed (4 runs each, as the test does) and prints, against eager #0, the
update rel-L2 and the number of parameter tensors that are not bit-identical.

  python tools/fp16_repro.py [--reps 6] [--prec fp16,bf16] > gpurun_out/fp16_repro.log
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from twingan_amd import Config, _lib      # noqa: E402
from twingan_amd.twingan import Trainer   # noqa: E402


def trajectory(prec, graph, hw=32, ch=32, steps=4, **kw):
  g = torch.Generator().manual_seed(8)
  s, t = torch.rand(4, hw, hw, 3, generator=g), torch.rand(4, hw, hw, 3, generator=g)
  dt = dict(fp16=torch.float16, bf16=torch.bfloat16, fp32=torch.float32)[prec]
  tr = Trainer(Config(hw=hw, max_ch=ch, precision=prec, loss_scale=128.0 if prec == 'fp16' else 1.0, **kw), device='cuda:0',
               seed=3, use_graph=graph)
  p0 = {k: v.clone() for k, v in tr.store.state_dict().items()}
  torch.manual_seed(11)
  for _ in range(steps):
    tr.run(s.cuda().to(dt), t.cuda().to(dt))
  torch.cuda.synchronize()
  assert not graph or tr.graph_fallback_reason is None, tr.graph_fallback_reason
  sd = {k: v.clone() for k, v in tr.store.state_dict().items()}
  tr.close()
  return p0, sd


def err(a, b, p0):
  num = sum(float((((a[k] - p0[k]) - (b[k] - p0[k])).double() ** 2).sum()) for k in a)
  den = sum(float(((b[k] - p0[k]).double() ** 2).sum()) for k in a)
  return (num / den) ** 0.5, sum(not torch.equal(a[k], b[k]) for k in a)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--reps', type=int, default=6)
  ap.add_argument('--prec', default='fp16,bf16')
  ap.add_argument('--variants', default='plain')
  a = ap.parse_args()
  lib = _lib.load()
  variants = dict(plain={}, sn_att=dict(spectral_norm=True, do_self_attention=True, self_attention_hw=16),
                  bn=dict(generator_norm_type='batch_norm'))
  for vname in a.variants.split(','):
    for prec in a.prec.split(','):
      for det in (0, 1):
        lib.tg_set_deterministic(det)
        p0, base = trajectory(prec, False, **variants[vname])
        rows = []
        for graph in (False, True):
          for r in range(a.reps):
            if not graph and r == 0:
              continue
            _, sd = trajectory(prec, graph, **variants[vname])
            e, nbad = err(sd, base, p0)
            rows.append((graph, e, nbad))
        ee = [e for g, e, _ in rows if not g]
        ge = [e for g, e, _ in rows if g]
        print('%-7s %s det=%d  eager-vs-eager0: max %.3e (tensors differing: %s)   graph-vs-eager0: max %.3e (tensors differing: %s)' % (
            vname, prec, det, max(ee), [n for g, _, n in rows if not g], max(ge), [n for g, _, n in rows if g]), flush=True)
  lib.tg_set_deterministic(0)


if __name__ == '__main__':
  main()
