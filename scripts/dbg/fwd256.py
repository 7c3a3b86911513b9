"""T_net forward at 2 x 256x256 with fixed weights: dump the output (RCOT_NT_OLD=1 selects the round-1 Gram kernel gate)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from rcot_amd.net_restormer import T_net
from rcot_amd.ops import HipBackend
net = T_net(decoder=True, backend=HipBackend(), seed=31)
g = torch.Generator().manual_seed(5)
x = torch.rand(2, 3, 256, 256, generator=g).cuda()
y = net(x)
torch.cuda.synchronize()
torch.save(y.cpu(), sys.argv[1])
print("saved", float(y.abs().max()))
