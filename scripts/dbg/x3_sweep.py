import os, sys
sys.path.insert(0, '.')
import torch
from rcot_amd import lib
from rcot_amd.ops import HipBackend
be = HipBackend(); be.prec = lib.PREC_BF16X3
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
for (B, N, Co) in ((8, 4096, 96), (8, 1024, 192), (8, 256, 384), (8, 16384, 96)):
    for Ci in (32, 64, 128, 256, 512, 1024):
        W = torch.randn(Co, Ci, device="cuda") * 0.1
        st_, sp = be.pack_shapes(Co, Ci)
        WT, WP = torch.zeros(*st_, device="cuda"), torch.zeros(*sp, device="cuda")
        be.pack_weight(W, WT, WP)
        byt = 4.0 * B * N * (Ci + Co)
        nbuf = max(2, int(600e6 // byt) + 1)
        sets = [(torch.randn(B, Ci, N, device="cuda"), torch.empty(B, Co, N, device="cuda")) for _ in range(nbuf)]
        fs = [(lambda X=X, Y=Y: be.conv1x1_fwd(W, X, Y, packed=(WT, WP))) for X, Y in sets]
        t = tm(fs)
        print(f"N={N:5d} M={Co:4d} K={Ci:5d}: {t:7.1f} us  {byt/t/1e3:6.0f} GB/s  {2.0*Co*Ci*B*N/t/1e6:6.1f} TF", flush=True)
        del sets
