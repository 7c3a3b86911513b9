"""Cold-buffer timing (round 6): the data gradient of a projection behind a LayerNorm + that LayerNorm's backward as TWO launches
(rcot_gemm_kmajor, rcot_ln_bwd) against ONE (rcot_conv1x1_dgrad_ln_bwd), exact fp32, B = 8, the transport map's C <= 96 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd import lib
from rcot_amd.ops import HipBackend
be = HipBackend()
be.prec = lib.PREC_FP32
SHAPES = [(8, 16384, 510, 96), (8, 16384, 288, 96), (8, 16384, 254, 48), (8, 16384, 144, 48), (8, 4096, 510, 96), (8, 4096, 288, 96)]
def tm(fs, reps=24):
    for f in fs: f()
    torch.cuda.synchronize()
    g, st = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.graph(g, stream=st):
        for i in range(reps): fs[i % len(fs)]()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for (B, N, Co, Ci) in SHAPES:
    byt = 4.0 * B * N * (Co + 3 * Ci)
    nbuf = max(2, int(600e6 // byt) + 1)
    W = torch.randn(Co, Ci, device="cuda") * 0.1
    st, sp = be.pack_shapes(Co, Ci)
    WT, WP = torch.zeros(*st, device="cuda"), torch.zeros(*sp, device="cuda")
    be.pack_weight(W, WT, WP)
    packed = (WT, WP)
    lw = torch.ones(Ci, device="cuda")
    sets = []
    for _ in range(nbuf):
        x = torch.randn(B, Ci, N, device="cuda"); dY = torch.randn(B, Co, N, device="cuda"); g = torch.empty(B, Ci, N, device="cuda")
        dres = torch.randn(B, Ci, N, device="cuda"); dx = torch.empty(B, Ci, N, device="cuda")
        mu, rs = torch.zeros(B, N, device="cuda"), torch.ones(B, N, device="cuda")
        sets.append((x, dY, g, dres, dx, mu, rs))
    d = tm([(lambda dY=dY, g=g: be.conv1x1_dgrad(W, dY, g, packed=packed)) for (x, dY, g, dres, dx, mu, rs) in sets])
    l = tm([(lambda x=x, g=g, dres=dres, dx=dx, mu=mu, rs=rs: be.ln_bwd(g, x, mu, rs, lw, dres, dx, None, None, slot=0)) for (x, dY, g, dres, dx, mu, rs) in sets])
    f = tm([(lambda x=x, dY=dY, dres=dres, dx=dx, mu=mu, rs=rs: be.conv1x1_dgrad_ln_bwd(dY, x, mu, rs, lw, dres, dx, packed, 0)) for (x, dY, g, dres, dx, mu, rs) in sets])
    print(f"B={B} N={N:5d} Co={Co:4d} Ci={Ci:3d}: dgrad {d:6.1f} us + ln_bwd {l:6.1f} us = {d + l:6.1f}   fused {f:6.1f} us   ({byt / f / 1e3:5.0f} GB/s algorithmic of the fused form)", flush=True)
    del sets
