"""Which host call sites launch torch fill kernels during one minimax iteration?"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rcot_amd import lib
from rcot_amd.net_restormer import T_net, F_net
from rcot_amd.ops import HipBackend
from rcot_amd.trainer import FlatOptimizer, MinimaxStep
be = HipBackend(); be.prec = lib.PREC_BF16X3
Tn, Fn = T_net(decoder=True, backend=be, seed=1), F_net(patch_size=128, backend=be, seed=2)
st = MinimaxStep(Tn, Fn, FlatOptimizer(Tn, "RMSprop", 5e-5), FlatOptimizer(Fn, "RMSprop", 1e-4), 1.0, 10000.0)
B = 8
x, y = torch.rand(B, 3, 128, 128).cuda(), torch.rand(B, 3, 128, 128).cuda()
de = [2] * B
st.set_de_ids(de)
ded, al = torch.tensor(de, dtype=torch.int32).cuda(), torch.rand(B).cuda()
st.iteration(x, y, ded, al, True)
sites = collections.Counter()
for name in ("zero_", "fill_"):
    orig = getattr(torch.Tensor, name)
    def wrap(self, *a, _o=orig, _n=name, **k):
        fr = traceback.extract_stack(limit=4)[:-1]
        sites[(_n, tuple(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-2:]))] += 1
        return _o(self, *a, **k)
    setattr(torch.Tensor, name, wrap)
for fn in ("zeros", "zeros_like", "full", "ones"):
    orig = getattr(torch, fn)
    def wrapf(*a, _o=orig, _n=fn, **k):
        fr = traceback.extract_stack(limit=4)[:-1]
        sites[(_n, tuple(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-2:]))] += 1
        return _o(*a, **k)
    setattr(torch, fn, wrapf)
st.iteration(x, y, ded, al, True)
torch.cuda.synchronize()
for k, v in sites.most_common(20):
    print(v, k)
