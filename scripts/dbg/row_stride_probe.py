"""Does the row stride of the operands matter to the pixel-reduction kernel?  At 128 x 128 every channel plane starts 64 KiB after
the last, so the 224 row pieces of one slab (64 bytes each) share their low 16 address bits: if the memory channels are selected from
those bits, a slab's requests all queue at one channel.  rcot_bmm_nt takes row strides: the same product (per image 510 x 96, K = 16384,
8 images) on operands whose rows are 16384, 16384 + 16, + 64, + 1040 floats apart; cold operands, timed as replayed HIP graphs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rcot_amd import lib
from rcot_amd.ops import HipBackend
be = HipBackend()
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
B, N = 8, 16384
for (M, Nn) in ((510, 96), (96, 96), (288, 96)):
    for prec, name in ((lib.PREC_FP32, "fp32"), (lib.PREC_BF16X3, "x3")):
        be.prec = prec
        row = []
        for pad in (0, 16, 64, 1040):
            ld = N + pad
            sets = []
            for _ in range(3):
                A = torch.randn(B, 1, M, ld, device="cuda")[..., :N]
                Bm = torch.randn(B, 1, Nn, ld, device="cuda")[..., :N]
                sets.append((A, Bm))
            C = torch.zeros(B, 1, M, Nn, device="cuda")
            row.append(tm([(lambda A=A, Bm=Bm: be.bmm_nt(A, Bm, C)) for (A, Bm) in sets]))
            del sets
        print(f"{M:3d} x {Nn:3d} K=16384 x 8 images {name:4s}: row stride 16384 {row[0]:6.1f} us | +16 {row[1]:6.1f} | +64 {row[2]:6.1f} | +1040 {row[3]:6.1f}", flush=True)
