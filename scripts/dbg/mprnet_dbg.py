"""debug: MPRNet on HIP vs the stock-ops form on the same GPU box (forward, spectrum penalty, per-tensor gradient errors)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import relerr, seeded_tensor
from rcot_amd import mprnet as MP
from rcot_amd.mprnet_hip import MPRNetHip
from rcot_amd.ops import default_backend

be = default_backend()
for seed, shape in ((5, (2, 3, 64, 64)), (5, (2, 3, 32, 48)), (None, (2, 3, 32, 48))):
    net = MPRNetHip(backend=be, seed=seed or 0)
    ref = MP.MPRNetT(seed=seed or 0)
    if seed is None:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_mprnet_gpu import _params
        net.load_state_dict(_params()); ref.load_state_dict(_params())
    else:
        ref.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
    x, r = seeded_tensor(31, shape, lo=0.0, hi=1.0), seeded_tensor(32, shape)
    yr = ref(x)
    (yr * r).sum().backward()
    net.zero_grad()
    y = net.forward(x.cuda(), save=True)
    net.backward(r.cuda())
    torch.cuda.synchronize()
    print("seed", seed, shape, "forward", relerr(y, yr))
    pen = lambda o: float(torch.fft.fft2(x - o.cpu()).abs().mean((1, 2, 3)).sum())
    print("  pen hip", pen(y), "stock", pen(yr.detach()))
    errs = []
    seen = set()
    for n, _ in MP.mprnet_param_shapes():
        t = ref.p[n]
        if id(t) in seen or t.grad is None:
            continue
        seen.add(id(t))
        key = net.slope_name if n.endswith("body.1.weight") else n
        g = net.store.g[key]
        errs.append((relerr(g, t.grad), n, float(t.grad.abs().max())))
    errs.sort(reverse=True)
    for e in errs[:8]:
        print("  ", e)
    print("   median", errs[len(errs) // 2][0])
