import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rcot_amd import lib
from rcot_amd.net_restormer import F_net, T_net
from rcot_amd.ops import default_backend
from rcot_amd.plan import LaunchPlan
from rcot_amd.synth import make_batch
from rcot_amd.trainer import FlatOptimizer, MinimaxStep
B, P = int(os.environ.get("RB", "2")), int(os.environ.get("RP", "64"))
be = default_backend()
PREC = {"fp32": lib.PREC_FP32, "bf16x3": lib.PREC_BF16X3}
be.prec = PREC["fp32"]
Tn, Fn = T_net(decoder=True, seed=1), F_net(patch_size=P, seed=2)
st = MinimaxStep(Tn, Fn, FlatOptimizer(Tn, "RMSprop", 5e-5), FlatOptimizer(Fn, "RMSprop", 1e-4), 1.0, 10000.0)
de = [2] * B
st.set_de_ids(de)
d = torch.tensor(de, dtype=torch.int32).cuda()
_, x, y = make_batch(5, B, P, de)
x, y = x.cuda(), y.cuda()
al = torch.full((B,), 0.5).cuda()
def steps(n, tag):
    for i in range(n):
        st.run(x, y, d, al, True)
    torch.cuda.synchronize(); print(tag, st.scalars(), flush=True)
def unit(tag):
    r = torch.randn_like(x)
    def u():
        Tn.zero_grad(); Tn.forward(x, save=True); Tn.backward(r)
    u(); torch.cuda.synchronize()
    if os.environ.get("RGRAPH", "1") == "1":
        pg, ps = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        with torch.cuda.graph(pg, stream=ps):
            u()
        pg.replay(); torch.cuda.synchronize(); del pg
        print(tag, "graph ok", flush=True)
    if os.environ.get("RPLAN", "1") == "1":
        pl = LaunchPlan(be).record(u)
        pl.replay(); torch.cuda.synchronize(); del pl
    print(tag, "unit ok", flush=True)
steps(3, "fp32")
st.iteration(x, y, d, al, True); torch.cuda.synchronize(); print("eager fp32 ok", flush=True)
if os.environ.get("RUNIT", "1") == "1":
    unit("fp32")
if os.environ.get("RWARM", "0") == "1":
    be.prec = PREC["bf16x3"]
    st.iteration(x, y, d, al, True); torch.cuda.synchronize(); print("eager x3 warm ok", flush=True)
be.prec = PREC["bf16x3"]
steps(3, "x3")
st.iteration(x, y, d, al, True); torch.cuda.synchronize(); print("eager x3 ok", flush=True)
unit("x3")
be.prec = PREC["fp32"]
steps(2, "fp32 again")
