"""Can an exact-fp32 projection (MFMA- / power-bound) and a stencil (HBM-bound) of the level-1 forward share the chip? (round 6)
Half-batch (B = 4) launches of project_in 510 <- 96 + LayerNorm and of the gate (depthwise 3x3 + GELU gate) at 128x128, cold operands:
  serial     : N x (projection, gate) on one stream
  concurrent : N projections on stream A, N gates on stream B
  full batch : N x (projection, gate) at B = 8 on one stream, halved — what the schedule does today
A gain of concurrent over serial is what a two-chain (half-batch) forward could get out of its level-1 stages."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rcot_amd import lib
from rcot_amd.ops import HipBackend
be = HipBackend()
be.prec = lib.PREC_FP32
C, hid, N, H = 96, 255, 16384, 128
W = torch.randn(2 * hid, C, device="cuda") * 0.1
st, sp = be.pack_shapes(2 * hid, C)
WT, WP = torch.zeros(*st, device="cuda"), torch.zeros(*sp, device="cuda")
lw, lb = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
be.pack_weight(W, WT, WP)
Wd = torch.randn(2 * hid, 9, device="cuda") * 0.1
def sets(B, n):
    out = []
    for _ in range(n):
        x = torch.randn(B, C, H, H, device="cuda"); p = torch.randn(B, 2 * hid, H, H, device="cuda"); p2 = torch.randn(B, 2 * hid, H, H, device="cuda")
        g = torch.empty(B, hid, H, H, device="cuda"); mu = torch.zeros(B, N, device="cuda"); rs = torch.ones(B, N, device="cuda")
        out.append((x, p, p2, g, mu, rs))
    return out
A, Bs = torch.cuda.Stream(), torch.cuda.Stream()
def proj(s_):
    x, p, p2, g, mu, rs = s_
    be.conv1x1_fwd(W, x, p, ln=(mu, rs, lw, lb), packed=(WT, WP))
def gate(s_):
    x, p, p2, g, mu, rs = s_
    be.gdfn_gate_fwd(p2, Wd, g)
def run(mode, S, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        s_ = S[r % len(S)]
        if mode == "serial":
            with torch.cuda.stream(A):
                proj(s_); gate(s_)
        else:
            with torch.cuda.stream(A):
                proj(s_)
            with torch.cuda.stream(Bs):
                gate(S[(r + 1) % len(S)])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
S4, S8 = sets(4, 6), sets(8, 3)
for _ in range(2):
    a = run("serial", S4, 60); b = run("concurrent", S4, 60); c = run("serial", S8, 30)
    print(f"B=4 halves: serial {a:6.1f} us per (projection + gate), concurrent {b:6.1f} us;   B=8 serial {c:6.1f} us = {c / 2:6.1f} per half", flush=True)
