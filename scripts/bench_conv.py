"""In-isolation timing of the dense conv entry points at the thin (RGB) shapes and a few regular ones (B=8, 128x128)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd.ops import HipBackend
be = HipBackend()
def tm(f, reps=20):
    for _ in range(3): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for (B, Ci, Co, H, k, p) in ((8, 96, 3, 128, 3, 1), (8, 3, 64, 128, 5, 2), (8, 3, 48, 128, 3, 1), (8, 48, 96, 64, 3, 1)):
    X = torch.randn(B, Ci, H, H, device="cuda"); Wt = torch.randn(Co, Ci, k, k, device="cuda") * 0.1
    Y = torch.empty(B, Co, H, H, device="cuda"); dY = torch.randn_like(Y); dX = torch.empty_like(X); dW = torch.zeros_like(Wt)
    fl = 2.0 * B * H * H * Ci * Co * k * k
    t = tm(lambda: be.conv2d_fwd(X, Wt, None, Y, 1, p)); print(f"fwd   {Ci:3d}->{Co:3d} k{k} {H}: {t:7.1f} us {fl/t/1e6:6.1f} TF/s")
    t = tm(lambda: be.conv2d_dgrad(dY, Wt, dX, 1, p)); print(f"dgrad {Ci:3d}->{Co:3d} k{k} {H}: {t:7.1f} us {fl/t/1e6:6.1f} TF/s")
    t = tm(lambda: be.conv2d_wgrad(dY, X, dW, 1, p, beta=1.0)); print(f"wgrad {Ci:3d}->{Co:3d} k{k} {H}: {t:7.1f} us {fl/t/1e6:6.1f} TF/s")
