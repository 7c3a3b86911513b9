"""Few launches of the two GEMM kernels on level-1 shapes, for a rocprofv3 --pmc pass (keep it tiny: counters serialise)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd.ops import HipBackend
be = HipBackend()
B, N = 8, 16384
for (Co, Ci, ln) in ((510, 96, True), (96, 510, False), (96, 96, False)):
    W = torch.randn(Co, Ci, device="cuda") * 0.1
    X = torch.randn(B, Ci, 128, 128, device="cuda"); Y = torch.empty(B, Co, 128, 128, device="cuda")
    dY = torch.randn(B, Co, 128, 128, device="cuda"); dW = torch.zeros(Co, Ci, device="cuda")
    st, sp = be.pack_shapes(Co, Ci)
    WT, WP = torch.zeros(*st, device="cuda"), torch.zeros(*sp, device="cuda")
    be.pack_weight(W, WT, WP)
    mu, rs = torch.zeros(B, N, device="cuda"), torch.ones(B, N, device="cuda")
    lw, lb = torch.ones(Ci, device="cuda"), torch.zeros(Ci, device="cuda")
    for _ in range(3):
        be.conv1x1_fwd(W, X, Y, ln=(mu, rs, lw, lb) if ln else None, packed=(WT, WP))
        be.conv1x1_wgrad(dY, X, dW, ln=(mu, rs, lw, lb) if ln else None, beta=1.0)
torch.cuda.synchronize()
