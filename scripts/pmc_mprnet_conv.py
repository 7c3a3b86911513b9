"""Counter-pass script (scripts/rocprof_traffic.sh): the 80-channel 3x3 products of the MPRNet transport map at 4 x 128 x 128 on the 16-row
forms of the convolution engine — forward, the data gradient as a forward product, weight gradient — three cold launches each (fresh
operands every time), so that FETCH_SIZE / WRITE_SIZE give the HBM bytes of one launch against its algorithmic bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd.ops import HipBackend
be = HipBackend()
B, C, H = 4, 80, 128
Wt = torch.randn(C, C, 3, 3, device="cuda") * 0.05
for _ in range(3):
    X = torch.randn(B, C, H, H, device="cuda"); Y = torch.empty(B, C, H, H, device="cuda")
    be.conv2d_fwd(X, Wt, None, Y, 1, 1)
    dY = torch.randn(B, C, H, H, device="cuda"); dW = torch.zeros_like(Wt)
    be.conv2d_wgrad(dY, X, dW, 1, 1, beta=1.0)
torch.cuda.synchronize()
print("algorithmic bytes per launch: forward", (2 * B * C * H * H + C * C * 9) * 4, " weight gradient", (2 * B * C * H * H + C * C * 9) * 4)
