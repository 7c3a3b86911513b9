"""Do the library's GEMM kernels run faster on ZERO operands than on random ones (power: operand bits that toggle cost energy, the clocks
follow)?  510 <- 96 at 8 x 128 x 128: forward (K-major kernel), data gradient, weight gradient (pixel-reduction kernel), three arithmetics,
cold operands, replayed HIP graphs; "zeros" = every activation operand zero (weights random)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rcot_amd import lib
from rcot_amd.ops import HipBackend
be = HipBackend()
be.x6_packs = True
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
B, N, Co, Ci = 8, 16384, 510, 96
W = torch.randn(Co, Ci, device="cuda") * 0.1
st, sp = be.pack_shapes(Co, Ci)
WT, WP = torch.zeros(*st, device="cuda"), torch.zeros(*sp, device="cuda")
lw, lb = torch.ones(Ci, device="cuda"), torch.zeros(Ci, device="cuda")
WTf, c12 = (torch.zeros(*s_, device="cuda") for s_ in be.fold_shapes(Co, Ci))
sp3 = tuple(torch.zeros(*be.split_shapes(Co, Ci)[i], device="cuda") for i in (0, 1, 0))
sp6 = tuple(torch.zeros(*be.split6_shapes(Co, Ci)[i], device="cuda") for i in (0, 1, 0))
be.pack_weight(W, WT, WP, (lw, lb, WTf, c12), sp3, sp6)
packed = (WT, WP, (WTf, c12), sp3, sp6)
for kind in ("random", "zeros"):
    sets = []
    for _ in range(3):
        mk = torch.randn if kind == "random" else torch.zeros
        X, Y, dX = mk(B, Ci, N, device="cuda"), mk(B, Co, N, device="cuda"), torch.empty(B, Ci, N, device="cuda")
        mu, rs = torch.zeros(B, N, device="cuda"), torch.ones(B, N, device="cuda")
        sets.append((X, Y, dX, mu, rs))
    dW = torch.zeros(Co, Ci, device="cuda")
    for prec, name in ((lib.PREC_FP32, "fp32"), (lib.PREC_BF16X6, "x6"), (lib.PREC_BF16X3, "x3")):
        be.prec = prec
        f = tm([(lambda X=X, Y=Y, mu=mu, rs=rs: be.conv1x1_fwd(W, X, Y, ln=(mu, rs, lw, lb), packed=packed)) for (X, Y, dX, mu, rs) in sets])
        d = tm([(lambda dX=dX, Y=Y: be.conv1x1_dgrad(W, Y, dX, packed=packed)) for (X, Y, dX, mu, rs) in sets])
        w = tm([(lambda X=X, Y=Y, mu=mu, rs=rs: be.conv1x1_wgrad(Y, X, dW, ln=(mu, rs, lw, lb), beta=0.0)) for (X, Y, dX, mu, rs) in sets])
        print(f"{kind:6s} operands {name:4s}: fwd {f:6.1f} us  dgrad {d:6.1f} us  wgrad {w:6.1f} us", flush=True)
    del sets
