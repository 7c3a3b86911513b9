import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from rcot_amd import params as P
from rcot_amd.ops import HipBackend
be = HipBackend()
out = {}
torch.manual_seed(0)
for B in (16, 8, 2):
    H = 128
    for li, (ci, co, k, s, pad, bias) in enumerate(P.FNET_CONVS):
        Ho = H // s
        g = torch.Generator(device="cuda").manual_seed(100 + li)
        X = torch.randn(B, ci, H, H, device="cuda", generator=g); Wt = torch.randn(co, ci, k, k, device="cuda", generator=g) * 0.02
        bv = torch.randn(co, device="cuda", generator=g) if bias else None
        Y = torch.empty(B, co, Ho, Ho, device="cuda"); dZ = torch.randn(B, co, Ho, Ho, device="cuda", generator=g); dX = torch.empty_like(X)
        dW = torch.zeros_like(Wt)
        be.conv2d_fwd(X, Wt, bv, Y, s, pad, 0.2, 0, None)
        be.conv2d_dgrad(dZ, Wt, dX, s, pad, 0.0)
        be.conv2d_wgrad(dZ, X, dW, s, pad, 1.0)
        out[(B, li)] = (Y.cpu(), dX.cpu(), dW.cpu())
        H = Ho
torch.save(out, sys.argv[1])
