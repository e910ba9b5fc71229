import sys, numpy as np, torch
sys.path.insert(0, '.')
import twingan_amd.ops as O
from twingan_amd import ops
rng = np.random.RandomState(2)
n,h,w,cin,cout,k = 16,4,4,256,256,4
x = torch.from_numpy(rng.randn(n,h,w,cin)).float().cuda().bfloat16()
wt = torch.from_numpy(rng.randn(k,k,cin,cout)/np.sqrt(k*k*cin)).float().cuda()
g = torch.from_numpy(np.random.RandomState(3).randn(n,1,1,cout)).float().cuda().bfloat16()
spec = O.ConvSpec(k, 'VALID')
res = {}
for algo in ('mfma','direct'):
  saved = O._mfma_ok
  if algo == 'direct': O._mfma_ok = lambda *a: False
  res[algo] = O.conv_bwd_weight_raw(x, g, spec).clone()
  res[algo+'y'] = O.conv_fwd_raw(x, wt, None, spec, 0).float()
  O._mfma_ok = saved
a, b = res['mfma'], res['direct']
print('gw rel', float((a-b).norm()/b.norm()), 'y rel', float((res['mfmay']-res['directy']).norm()/res['directy'].norm()))
d = (a-b).abs().reshape(16, cin, cout)
print('per tap max err', d.amax(dim=(1,2)).cpu().numpy())
print('per ci-block(32) err', d.reshape(16, 8, 32, cout).amax(dim=(0,2,3)).cpu().numpy())
