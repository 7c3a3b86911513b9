"""Which Python lines of one minimax iteration launch PyTorch's own kernels (fill_, copy_, ...) instead of ours: torch.profiler
with stacks over one eager iteration at BASELINE configs[1]; prints, per aten op that launched a device kernel, the call sites."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from rcot_amd import lib
from rcot_amd.net_restormer import F_net, T_net
from rcot_amd.ops import default_backend
from rcot_amd.synth import make_batch
from rcot_amd.trainer import FlatOptimizer, MinimaxStep

B, P = 8, 128
be = default_backend()
be.prec = lib.PREC_BF16X3
Tn, Fn = T_net(decoder=True, seed=1234), F_net(patch_size=P, seed=1235)
st = MinimaxStep(Tn, Fn, FlatOptimizer(Tn, "RMSprop", 5e-5), FlatOptimizer(Fn, "RMSprop", 1e-4), 1.0, 10000.0)
de = [0] * B
st.set_de_ids(de)
de_dev = torch.tensor(de, dtype=torch.int32, device="cuda")
_, x, y = make_batch(1, B, P, de, unpaired=False)
x, y = x.cuda(), y.cuda()
al = torch.rand(B).cuda()
for _ in range(2):
    st.iteration(x, y, de_dev, al, True)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    st.iteration(x, y, de_dev, al, True)
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and (ev.cpu_parent is None or not ev.cpu_parent.name.startswith("aten::")):
        fr = [s for s in ev.stack if "rcot_amd" in s or "bench" in s]
        sites[(ev.name, fr[0] if fr else (ev.stack[0] if ev.stack else "?"))] += 1
for (name, where), n in sites.most_common(40):
    print(f"{n:5d}  {name:28s} {where}")
