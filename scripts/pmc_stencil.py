"""A few launches of the depthwise-stencil kernels at the level-1 shapes, for rocprofv3 --pmc passes (tiny on purpose)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd.ops import HipBackend
be = HipBackend()
B, H, hid, C = 8, 128, 255, 96
p_ = torch.randn(B, 2 * hid, H, H, device="cuda"); w = torch.randn(2 * hid, 9, device="cuda") * 0.3
dg = torch.randn(B, hid, H, H, device="cuda"); dp = torch.empty_like(p_); dw = torch.zeros(2 * hid, 9, device="cuda")
g = torch.empty(B, hid, H, H, device="cuda")
t = torch.randn(B, 3 * C, H, H, device="cuda"); u = torch.empty_like(t); wq = torch.randn(3 * C, 9, device="cuda") * 0.3
du = torch.randn_like(t); dt = torch.empty_like(t); dwq = torch.zeros(3 * C, 9, device="cuda")
x = torch.randn(B, C, H, H, device="cuda"); gl = torch.randn_like(x); dx = torch.empty_like(x)
mu, rs = torch.zeros(B, H * H, device="cuda"), torch.ones(B, H * H, device="cuda")
lw = torch.ones(C, device="cuda"); dlw, dlb = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
for _ in range(3):
    be.gdfn_bwd(p_, w, dg, dp, dw)
    be.gdfn_gate_fwd(p_, w, g)
    be.dwconv3x3(t, wq, u)
    be.dwconv3x3_bwd(du, t, wq, dt, dwq)
    be.ln_stats(x, mu, rs)
    be.ln_bwd(gl, x, mu, rs, lw, x, dx, dlw, dlb)
torch.cuda.synchronize()
