"""debug: one minimax iteration, MPRNet on HIP (MinimaxStep) vs the stock-ops loop, component by component"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import relerr, seeded_tensor
from rcot_amd import mprnet as MP
from rcot_amd.mprnet_hip import MPRNetHip
from rcot_amd.net_restormer import F_net
from rcot_amd.ops import default_backend
from rcot_amd.synth import make_batch
from rcot_amd.trainer import FlatOptimizer, MinimaxStep

be = default_backend()
B, ps, de = 2, 64, [7, 7]
Tn, Fn = MPRNetHip(backend=be, seed=5), F_net(ps, backend=be, seed=6)
Tr, Fr = MP.MPRNetT(seed=5), MP.FNetTorch(ps, seed=6)
Tr.load_state_dict({k: v.cpu() for k, v in Tn.state_dict().items()})
Fr.load_state_dict({k: v.cpu() for k, v in Fn.state_dict().items()})
lr = 1e-4
st = MinimaxStep(Tn, Fn, FlatOptimizer(Tn, "RMSprop", lr / 2), FlatOptimizer(Fn, "RMSprop", lr), 1.0, 10000.0)
st.set_de_ids(de)
To, Fo = torch.optim.RMSprop(Tr.parameters(), lr=lr / 2), torch.optim.RMSprop(Fr.parameters(), lr=lr)
_, x, y = make_batch(77, B, ps, de)
alpha = seeded_tensor(78, (B,), lo=0.0, hi=1.0)
grads = {}
def probe(tag):
    torch.cuda.synchronize()
    net = Tn if tag == "T_gen" else Fn
    grads[tag] = {n: net.store.g[n].detach().cpu().clone() for n, _ in net.store.shapes}
st.grad_probe = probe
st.iteration(x.cuda(), y.cuda(), torch.tensor(de, dtype=torch.int32).cuda(), alpha.cuda(), False)
torch.cuda.synchronize()
s = st.scalars()
L = st.logs
print("hip  ", s, "fo", L["fo"].cpu().tolist(), "scal", L["scal"].cpu().tolist())
# stock loop, instrumented
with torch.no_grad():
    fake = Tr(x)
Fr.zero_grad()
loss_f = -Fr(y).mean() + Fr(fake).mean(); loss_f.backward()
gF1 = {n: (t.grad.clone() if t.grad is not None else None) for n, t in Fr.p.items()}
Fo.step(); Fr.zero_grad()
a = alpha.view(B, 1, 1, 1)
interp = (a * y + (1 - a) * fake).requires_grad_(True)
(g,) = torch.autograd.grad(Fr(interp), interp, torch.ones(B), create_graph=True)
gp = 10.0 * ((g.flatten(1).pow(2).sum(1).sqrt() - 1) ** 2).mean(); gp.backward()
gF2 = {n: (t.grad.clone() if t.grad is not None else None) for n, t in Fr.p.items()}
Fo.step(); Fr.zero_grad(); Tr.zero_grad()
out = Tr(x); res = x - out
fo = Fr(out)
rmse = res.pow(2).mean().sqrt(); fr = torch.fft.fft2(res)
pen = sum(fr[i].abs().mean() for i in range(B))
print("stock", dict(Loss_F=float(loss_f), Loss_T=float(-fo.mean() + rmse + pen), Loss_mse=float(rmse), gp=float(gp)), "fo", fo.tolist(), "pen", float(pen))
for tag, gg in (("F_critic", gF1), ("F_gp", gF2)):
    worst = []
    for n, t in gg.items():
        if t is None:
            continue
        worst.append((relerr(grads[tag][n], t), n, float(t.abs().max())))
    worst.sort(reverse=True)
    print(tag, worst[:5])
for n in ("features.0.weight", "fc.weight", "fc2.weight", "features.0.bias", "fc.bias"):
    print(n, "param diff after 2 steps", float((Fn.store.p[n].cpu() - Fr.p[n].detach()).abs().max()), "max", float(Fr.p[n].abs().max()))
