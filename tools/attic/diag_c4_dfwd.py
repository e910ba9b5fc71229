"""Forward diagnostic for tools/diag_c4_dgroup.py: per end-point error of both discriminators (fp16 kernels vs float64 oracle)
on the SAME (fp16-rounded) real / prime inputs, plus the gradient at the tail when only one image group feeds the loss."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from oracle import torch_ref as R                  # noqa: E402
from test_gpu_model import make                    # noqa: E402
from twingan_amd import pggan                      # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else 'fp16'
kw = dict(hw=32, max_ch=64, spectral_norm=True, do_self_attention=True, self_attention_hw=16, loss_architecture='wgan_gp', loss_scale=128.0)
cfg, rcfg, tr, Pref, dev, ref = make(kw, prec, seed=6, batch=2)
rcfg.sn_state = R.init_sn_state(Pref, seed=3)
sn0 = {k: v.float() for k, v in rcfg.sn_state.items()}
adt = dev['s'].dtype


def reset():
  if rcfg.sn_cache:
    rcfg.sn_cache.clear()
  for k, v in sn0.items():
    rcfg.sn_state[k] = v.double()
    tr.store.state[k].copy_(v)


def rel(a, b):
  return float((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30))


with torch.no_grad():
  reset()
  o = R.forward_generators(Pref, ref['s'], ref['t'], rcfg)
  for d, real, prime in (('s', ref['s'], o['s_prime']), ('t', ref['t'], o['t_prime'])):
    top = 'discriminator_' + d
    prime16 = prime.to(adt).double()
    for nm, x in (('real', real), ('prime', prime16)):
      reset()
      pr, ep_r = R.discriminator(Pref, x, rcfg, top)
      reset()
      ph, ep_h = pggan.discriminator(tr.P, x.to('cuda:0').to(adt).contiguous(), cfg, top)
      pggan.end_run(tr.P)
      keys = [k for k in ep_r if k in ep_h and torch.is_tensor(ep_r[k]) and tuple(ep_r[k].shape[:3]) == tuple(ep_h[k].shape[:3])]
      print('D_%s(%s): prediction %s vs %s' % (d, nm, ph.flatten().tolist(), pr.flatten().tolist()))
      for k in keys:
        c = ep_r[k].shape[-1]
        print('    %-40s rel %.2e   |ref| %.3e' % (k, rel(ep_h[k][..., :c], ep_r[k]), float(ep_r[k].norm())))
