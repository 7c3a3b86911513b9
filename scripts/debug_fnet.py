import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
from conftest import seeded_tensor
from oracle import rcot_oracle as O
from rcot_amd import params as P
from rcot_amd.net_restormer import F_net
ps = 128
prm = {k: torch.from_numpy(v) for k, v in P.seeded_params(P.fnet_param_shapes(ps), 21, "F").items()}
net = F_net(patch_size=ps); net.load_state_dict(prm)
x = seeded_tensor(602, (2, 3, ps, ps), lo=0.0, hi=1.0)
po = {k: v.clone().double().requires_grad_(True) for k, v in prm.items()}
(-O.fnet_forward(po, x.double()).mean()).backward()
p32 = {k: v.clone().requires_grad_(True) for k, v in prm.items()}
(-O.fnet_forward(p32, x).mean()).backward()
net.zero_grad(); net.forward(x.cuda(), save=True); net.backward(torch.full((2,), -0.5, device="cuda"), wgrad=True)
for k in po:
    g = net.store.g[k].cpu().double(); r = po[k].grad; r32 = p32[k].grad.double()
    print(f"{k:22s} hip-vs-fp64 max {float((g-r).abs().max()/r.abs().max()):.2e} norm {abs(float(g.norm()-r.norm()))/float(r.norm()):.2e} | torch32-vs-fp64 max {float((r32-r).abs().max()/r.abs().max()):.2e} norm {abs(float(r32.norm()-r.norm()))/float(r.norm()):.2e}")
