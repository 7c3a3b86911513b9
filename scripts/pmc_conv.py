"""Few launches of the implicit-GEMM conv entry points, for a rocprofv3 --pmc pass (tiny on purpose)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd.ops import HipBackend
be = HipBackend()
for (B, Ci, Co, H, k, s, p) in ((8, 96, 48, 64, 3, 1, 1), (16, 64, 128, 64, 3, 1, 1), (16, 128, 128, 64, 4, 2, 1)):
    X = torch.randn(B, Ci, H, H, device="cuda"); Wt = torch.randn(Co, Ci, k, k, device="cuda") * 0.1
    OH = (H + 2 * p - k) // s + 1
    Y = torch.empty(B, Co, OH, OH, device="cuda"); dY = torch.randn_like(Y); dX = torch.empty_like(X); dW = torch.zeros_like(Wt)
    for _ in range(3):
        be.conv2d_fwd(X, Wt, None, Y, s, p)
        be.conv2d_dgrad(dY, Wt, dX, s, p)
        be.conv2d_wgrad(dY, X, dW, s, p, beta=1.0)
torch.cuda.synchronize()
