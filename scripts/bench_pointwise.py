"""In-isolation timing of the HBM-bound kernels at the shapes of the four Restormer levels (B=8, 128x128 patches)."""
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
B = 8
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for (C, H) in ((48, 128), (96, 128), (96, 64), (192, 32), (384, 16), (192, 64), (384, 32)):
    N = H * H
    g, x = torch.randn(B, C, H, H, device="cuda"), torch.randn(B, C, H, H, device="cuda")
    mu, rs = torch.zeros(B, N, device="cuda"), torch.ones(B, N, device="cuda")
    w = torch.ones(C, device="cuda"); dres = torch.randn_like(g); dx = torch.empty_like(g)
    dw, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    be.ln_stats(x, mu, rs)
    t = tm(lambda: be.ln_bwd(g, x, mu, rs, w, dres, dx, dw, db))
    byt = 4.0 * g.numel() * 4
    print(f"ln_bwd  C={C:4d} {H:3d}x{H:<3d}: {t:7.1f} us  {byt / t / 1e3:7.0f} GB/s")
    t = tm(lambda: be.ln_stats(x, mu, rs))
    print(f"ln_stat C={C:4d} {H:3d}x{H:<3d}: {t:7.1f} us  {4.0 * g.numel() / t / 1e3:7.0f} GB/s")

for (hid, H) in ((255, 128), (255, 64), (510, 32), (1021, 16)):
    p_ = torch.randn(B, 2 * hid, H, H, device="cuda"); w = torch.randn(2 * hid, 9, device="cuda") * 0.3
    dg = torch.randn(B, hid, H, H, device="cuda"); dd = torch.empty_like(p_); dw = torch.zeros(2 * hid, 9, device="cuda")
    byt = 4.0 * (p_.numel() * 2 + dg.numel())
    t0 = tm(lambda: be.gdfn_gate_bwd(p_, w, dg, dd))
    t1 = tm(lambda: be.dwconv3x3_wgrad(dd, p_, dw))
    t2 = tm(lambda: be.gdfn_gate_bwd(p_, w, dg, dd, dw=dw))
    dp = torch.empty_like(p_)
    t3 = tm(lambda: be.dwconv3x3(dd, w, dp, flip=True))
    t4 = tm(lambda: be.gdfn_bwd(p_, w, dg, dp, dw))
    print(f"gate_bwd hid={hid:4d} {H:3d}x{H:<3d}: plain {t0:6.1f} + wgrad {t1:6.1f} = {t0+t1:6.1f} ; gate+wgrad fused {t2:6.1f} ; + flip dwconv {t3:6.1f} = {t2+t3:6.1f} ; one pass {t4:6.1f} us")
