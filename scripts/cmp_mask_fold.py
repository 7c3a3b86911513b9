"""The LeakyReLU backward folded into the store of the critic's data gradients / linearised forward sweep (rcot_conv2d_* mask
argument) against the separate rcot_lrelu_bwd launch: every parameter gradient of the critic-loss backward and of the gradient
penalty, and the input gradient, must be BIT-identical (torch.equal)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd.net_restormer import F_net
from rcot_amd.ops import HipBackend


def run(unfused):
    be = HipBackend()
    if unfused:
        dg, fw = be.conv2d_dgrad, be.conv2d_fwd

        def dgrad(dY, Wt, dX, stride, pad, beta=0.0, mask=None, mslope=1.0):
            dg(dY, Wt, dX, stride, pad, beta)
            if mask is not None:
                be.lrelu_bwd(dX, mask, dX, mslope)

        def fwd(X, Wt, bias, Y, stride, pad, lrelu=1.0, cmap=0, R=None, mask=None, mslope=1.0):
            fw(X, Wt, bias, Y, stride, pad, lrelu, cmap, R)
            if mask is not None:
                be.lrelu_bwd(Y, mask, Y, mslope)
        be.conv2d_dgrad, be.conv2d_fwd = dgrad, fwd
    out = []
    for ps, B in ((128, 16), (128, 8), (256, 2), (64, 4)):
        Fn = F_net(patch_size=ps, backend=be, seed=7)
        g = torch.Generator(device="cuda").manual_seed(ps + B)
        x = torch.rand(B, 3, ps, ps, device="cuda", generator=g)
        w = torch.randn(B, device="cuda", generator=g)
        Fn.zero_grad()
        Fn.forward(x, save=True)
        dx = Fn.backward(w, wgrad=True, need_dx=True)
        out += [dx.clone(), Fn.store.grad.clone()]
        Fn.zero_grad()
        gp = be.empty(1)
        Fn.gradient_penalty_backward(x, 1.0 / B, gp)
        out += [gp.clone(), Fn.store.grad.clone()]
    torch.cuda.synchronize()
    return out


a, b = run(True), run(False)
bad = [i for i, (u, v) in enumerate(zip(a, b)) if not torch.equal(u, v)]
print("tensors compared:", len(a), " not bit-identical:", bad)
