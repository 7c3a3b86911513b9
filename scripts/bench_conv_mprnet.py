"""The 3x3 convolutions of the MPRNet transport map (80 / 128 / 176 channels at 128 / 64 / 32 pixels, B = 4) on the convolution engine,
forward / data gradient / weight gradient, hot operands, back to back (SHAPES="B,C,H;.." for others)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd.ops import HipBackend
be = HipBackend()
def tm(f, reps=30):
    for _ in range(3): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
shapes = os.environ.get("SHAPES", "4,80,128;4,128,64;4,176,32;4,96,128;4,64,128")
for sh in shapes.split(";"):
    B, C, H = (int(v) for v in sh.split(","))
    X = torch.randn(B, C, H, H, device="cuda"); Wt = torch.randn(C, C, 3, 3, device="cuda") * 0.05
    Y = torch.empty(B, C, H, H, device="cuda"); dY = torch.randn_like(Y); dX = torch.empty_like(X); dW = torch.zeros_like(Wt)
    fl = 2.0 * B * H * H * C * C * 9
    a = tm(lambda: be.conv2d_fwd(X, Wt, None, Y, 1, 1)); b = tm(lambda: be.conv2d_dgrad(dY, Wt, dX, 1, 1)); c = tm(lambda: be.conv2d_wgrad(dY, X, dW, 1, 1, beta=1.0))
    print(f"C={C:3d} {H:3d}x{H:<3d} B={B}: {fl/1e9:5.2f} GF  fwd {a:6.1f} us {fl/a/1e6:5.1f} TF/s | dgrad {b:6.1f} us {fl/b/1e6:5.1f} | wgrad {c:6.1f} us {fl/c/1e6:5.1f}")
