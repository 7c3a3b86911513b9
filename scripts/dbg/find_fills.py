"""Where do at::native fill / copy kernels inside a PLANNED iteration come from?  torch.profiler over three plan replays."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import ProfilerActivity, profile
from rcot_amd import lib
from rcot_amd.net_restormer import F_net, T_net
from rcot_amd.ops import default_backend
from rcot_amd.synth import make_batch
from rcot_amd.trainer import FlatOptimizer, MinimaxStep

B, P = 8, 128
be = default_backend()
Tn, Fn = T_net(decoder=True, seed=1234), F_net(patch_size=P, seed=1235)
st = MinimaxStep(Tn, Fn, FlatOptimizer(Tn, "RMSprop", 5e-5), FlatOptimizer(Fn, "RMSprop", 1e-4), 1.0, 10000.0)
de = [0] * B
st.set_de_ids(de)
de_dev = torch.tensor(de, dtype=torch.int32, device="cuda")
_, x, y = make_batch(1, B, P, de, unpaired=False)
x, y = x.cuda(), y.cuda()
al = torch.rand(B).cuda()
for _ in range(3):
    st.run(x, y, de_dev, al, True)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(3):
        st.run(x, y, de_dev, al, True)
    torch.cuda.synchronize()
kn = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        kn[ev.name[:90]] += 1
for name, n in kn.most_common(200):
    if "at::" in name or "rocclr" in name or "Memset" in name or "Memcpy" in name:
        print(f"{n:5d}  {name}")
sites = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name not in ("aten::empty", "aten::view", "aten::unsqueeze", "aten::select", "aten::expand", "aten::slice", "aten::alias"):
        fr = [s for s in ev.stack if "rcot_amd" in s or "bench" in s or "scripts" in s]
        sites[(ev.name, fr[0] if fr else (ev.stack[0] if ev.stack else "?"))] += 1
for (name, where), n in sites.most_common(30):
    print(f"{n:5d}  {name:28s} {where}")
