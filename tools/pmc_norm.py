"""Driver for the PMC passes of the normalisation backward (tools/pmc_norm.sh): runs tg_norm_act_bwd (= the launch pair
norm_act_bwd1_kernel + norm_act_bwd2_part_kernel) a few times at one of the bench step's shapes, bf16, instance norm +
LeakyReLU + pixel norm over two domains -- the dominant kernel family of the round-3 bench line."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import twingan_amd.ops as O      # noqa: E402


def main():
  c, hw, n, reps = (int(v) for v in sys.argv[1:5])
  g = torch.Generator(device='cuda').manual_seed(1)
  y = (torch.randn((n, hw, hw, c), generator=g, device='cuda') * 0.7 + 0.3).to(torch.bfloat16).requires_grad_(True)
  par = [torch.ones(c, device='cuda').requires_grad_(True), torch.zeros(c, device='cuda').requires_grad_(True),
         torch.ones(c, device='cuda').requires_grad_(True), torch.zeros(c, device='cuda').requires_grad_(True)]
  gz = torch.randn((n, hw, hw, c), generator=g, device='cuda').to(torch.bfloat16)
  for _ in range(reps):
    z = O.norm_act(y, par[0], par[1], lrelu=True, pixel_norm=True, gamma2=par[2], beta2=par[3], split=n // 2)
    z.backward(gz)
    y.grad = None
  torch.cuda.synchronize()


if __name__ == '__main__':
  main()
