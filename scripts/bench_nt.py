"""Cold-buffer timing of the pixel-reduction GEMMs (1x1 weight gradients, rcot_conv1x1_wgrad) in both arithmetic modes on
the shapes of the transport map, each call timed inside a replayed HIP graph (scripts/bench_x3.py).  Prints us and
algorithmic GB/s (both operands read once)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd import lib
from rcot_amd.ops import HipBackend
be = HipBackend()
SHAPES = [(8, 16384, 510, 96, True), (8, 16384, 288, 96, True), (8, 16384, 96, 255, False), (8, 16384, 96, 96, False),
          (8, 16384, 144, 48, True), (8, 16384, 48, 127, False), (8, 4096, 510, 96, True), (8, 4096, 96, 255, False),
          (8, 1024, 1020, 192, True), (8, 1024, 192, 510, False), (8, 256, 2042, 384, True), (8, 256, 384, 1021, False)]
if os.environ.get("NT_SHAPES"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["NT_SHAPES"].split(",")]
if os.environ.get("NT_NOLN"):                     # what the on-the-fly LayerNorm of X costs: the same shapes without it
    SHAPES = [(b, n, co, ci, False) for (b, n, co, ci, _) in SHAPES]
PRECS = (lib.PREC_BF16X3,) if os.environ.get("X3_ONLY") else (lib.PREC_FP32, lib.PREC_BF16X3)
def tm(fs, reps=24):
    for f in fs: f()
    torch.cuda.synchronize()
    g, st = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.graph(g, stream=st):
        for i in range(reps): fs[i % len(fs)]()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for (B, N, Co, Ci, ln) in SHAPES:
    byt = 4.0 * B * N * (Ci + Co)
    nbuf = max(2, int(600e6 // byt) + 1)
    lw, lb = torch.ones(Ci, device="cuda"), torch.zeros(Ci, device="cuda")
    sets = []
    for _ in range(nbuf):
        X = torch.randn(B, Ci, N, device="cuda"); dY = torch.randn(B, Co, N, device="cuda")
        mu, rs = torch.zeros(B, N, device="cuda"), torch.ones(B, N, device="cuda")
        sets.append((X, dY, mu, rs))
    dW = torch.zeros(Co, Ci, device="cuda")
    out = []
    for prec in PRECS:
        be.prec = prec
        fs = [(lambda X=X, dY=dY, mu=mu, rs=rs: be.conv1x1_wgrad(dY, X, dW, ln=(mu, rs, lw, lb) if ln else None, beta=0.0)) for (X, dY, mu, rs) in sets]
        ms = tm(fs)
        out.append(f"{'fp32' if prec == 0 else 'x3'}: {ms*1e3:7.1f} us {byt/ms/1e6:6.0f} GB/s")
    print(f"B={B} N={N:5d} dW {Co:4d}x{Ci:4d} ln={int(ln)}:  " + "   ".join(out), flush=True)
    del sets
