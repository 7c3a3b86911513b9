import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rcot_amd import lib
from rcot_amd.net_restormer import F_net, T_net
from rcot_amd.ops import default_backend
from rcot_amd.synth import make_batch
from rcot_amd.trainer import FlatOptimizer, MinimaxStep
B, P = int(os.environ.get("RB", "2")), int(os.environ.get("RP", "64"))
be = default_backend()
be.prec = lib.PREC_BF16X3
Tn, Fn = T_net(decoder=True, seed=1), F_net(patch_size=P, seed=2)
st = MinimaxStep(Tn, Fn, FlatOptimizer(Tn, "RMSprop", 5e-5), FlatOptimizer(Fn, "RMSprop", 1e-4), 1.0, 10000.0)
de = [2] * B
st.set_de_ids(de)
d = torch.tensor(de, dtype=torch.int32).cuda()
_, x, y = make_batch(5, B, P, de)
x, y = x.cuda(), y.cuda()
al = torch.full((B,), 0.5).cuda()
for i in range(4):
    st.run(x, y, d, al, True)
    torch.cuda.synchronize(); print("x3 plan step", i, st.scalars(), flush=True)
