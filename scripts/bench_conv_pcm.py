"""The critic's inner convolutions (Net_Restormer.py:447-487) at the critic-step batch (2 x 8 images, 128x128 patches): implicit-GEMM
engine (exact fp32, rcot_conv2d_*) vs the padded-plane bf16x3 product (rcot_conv_pcm_*, prep + product [+ merge]), forward and data
gradient, hot operands, HIP-event timing of back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd import params as P
from rcot_amd.ops import HipBackend
be = HipBackend()


def tm(f, reps=20):
    for _ in range(3):
        f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


B, H = int(os.environ.get("B", "16")), 128
tot = [0.0, 0.0, 0.0, 0.0]
for li, (ci, co, k, s, pad, bias) in enumerate(P.FNET_CONVS):
    Ho = H // s
    if li > 0:
        X = torch.randn(B, ci, H, H, device="cuda"); Wt = torch.randn(co, ci, k, k, device="cuda") * 0.02
        bv = torch.randn(co, device="cuda") if bias else None
        Y = torch.empty(B, co, Ho, Ho, device="cuda"); dZ = torch.randn_like(Y); dX = torch.empty_like(X)
        pf, pd = be.conv_pcm_pack(Wt, "fwd"), be.conv_pcm_pack(Wt, "dgrad")
        t0 = tm(lambda: be.conv2d_fwd(X, Wt, bv, Y, s, pad, 0.2, 0, None))
        t1 = tm(lambda: be.conv_pcm_fwd(X, pf, bv, Y, k, 0.2))
        t2 = tm(lambda: be.conv2d_dgrad(dZ, Wt, dX, s, pad, 0.0))
        t3 = tm(lambda: be.conv_pcm_dgrad(dZ, pd, dX, k))
        fl = 2.0 * B * co * ci * k * k * Ho * Ho
        for i, t in enumerate((t0, t1, t2, t3)):
            tot[i] += t
        print(f"layer {li}: {ci:3d}->{co:3d} k{k}s{s} {H:3d}->{Ho:3d}  fwd {t0:6.1f} -> {t1:6.1f} us ({fl / t1 / 1e6:5.0f} TF/s)   dgrad {t2:6.1f} -> {t3:6.1f} us ({fl / t3 / 1e6:5.0f} TF/s)")
    H = Ho
print(f"sum over layers 1..9: fwd {tot[0]:.0f} -> {tot[1]:.0f} us, dgrad {tot[2]:.0f} -> {tot[3]:.0f} us")
