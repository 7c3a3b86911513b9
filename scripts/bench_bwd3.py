"""Cold-buffer timing of the three products of one 1x1 projection — forward, data gradient, weight gradient (with the LayerNorm
recomputed in the loop where the layer has one) — in the three arithmetics, on the transport map's shapes at B = 8.  Each call
works on the next of NBUF operand sets (nothing served from the Infinity Cache); GPU-side time per call from a captured graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd import lib
from rcot_amd.ops import HipBackend
be = HipBackend()
be.x6_packs = True
SHAPES = [(8, 16384, 510, 96, True), (8, 16384, 288, 96, True), (8, 16384, 96, 255, False), (8, 16384, 96, 96, False),
          (8, 4096, 510, 96, True), (8, 4096, 96, 255, False), (8, 1024, 1020, 192, True), (8, 1024, 192, 510, False),
          (8, 256, 2042, 384, True), (8, 256, 384, 1021, False), (8, 256, 1152, 384, True)]
if os.environ.get("X3_SHAPES"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["X3_SHAPES"].split(",")]
if os.environ.get("BWD3_SHAPE_LIST"):      # "B,N,Co,Ci,ln;..." : any shapes (K sweeps)
    SHAPES = [tuple(int(v) for v in t.split(",")[:4]) + (bool(int(t.split(",")[4])),) for t in os.environ["BWD3_SHAPE_LIST"].split(";")]
PRECS = (lib.PREC_FP32, lib.PREC_BF16X6, lib.PREC_BF16X3)
if os.environ.get("BWD3_PRECS"):
    PRECS = tuple({"fp32": lib.PREC_FP32, "x6": lib.PREC_BF16X6, "x3": lib.PREC_BF16X3}[n] for n in os.environ["BWD3_PRECS"].split(","))
NAMES = {lib.PREC_FP32: "fp32", lib.PREC_BF16X3: "x3", lib.PREC_BF16X6: "x6"}
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
for (B, N, Co, Ci, ln) in SHAPES:
    byt = 4.0 * B * N * (Ci + Co)
    nbuf = max(2, int(600e6 // byt) + 1)
    W = torch.randn(Co, Ci, device="cuda") * 0.1
    st, sp = be.pack_shapes(Co, Ci)
    WT, WP = torch.zeros(*st, device="cuda"), torch.zeros(*sp, device="cuda")
    lw, lb = torch.ones(Ci, device="cuda"), torch.zeros(Ci, device="cuda")
    WTf, c12 = (torch.zeros(*s_, device="cuda") for s_ in be.fold_shapes(Co, Ci))
    sp3 = tuple(torch.zeros(*be.split_shapes(Co, Ci)[i], device="cuda") for i in (0, 1, 0))
    sp6 = tuple(torch.zeros(*be.split6_shapes(Co, Ci)[i], device="cuda") for i in (0, 1, 0))
    be.pack_weight(W, WT, WP, (lw, lb, WTf, c12), sp3, sp6)
    packed = (WT, WP, (WTf, c12), sp3, sp6)
    sets = []
    for _ in range(nbuf):
        X = torch.randn(B, Ci, N, device="cuda"); Y = torch.randn(B, Co, N, device="cuda"); dX = torch.empty(B, Ci, N, device="cuda")
        mu, rs = torch.zeros(B, N, device="cuda"), torch.ones(B, N, device="cuda")
        sets.append((X, Y, dX, mu, rs))
    dW = torch.zeros(Co, Ci, device="cuda")
    print(f"B={B} N={N:5d} Co={Co:4d} Ci={Ci:4d} ln={int(ln)} ({byt/1e6:.0f} MB, {2.0*Co*Ci*B*N/1e9:.1f} GF):", flush=True)
    for prec in PRECS:
        be.prec = prec
        lnf = lambda mu, rs: (mu, rs, lw, lb) if ln else None
        f = tm([(lambda X=X, Y=Y, mu=mu, rs=rs: be.conv1x1_fwd(W, X, Y, ln=lnf(mu, rs), packed=packed)) for (X, Y, dX, mu, rs) in sets])
        d = tm([(lambda dX=dX, Y=Y: be.conv1x1_dgrad(W, Y, dX, packed=packed)) for (X, Y, dX, mu, rs) in sets])
        w = tm([(lambda X=X, Y=Y, mu=mu, rs=rs: be.conv1x1_wgrad(Y, X, dW, ln=lnf(mu, rs), beta=0.0)) for (X, Y, dX, mu, rs) in sets])
        print(f"   {NAMES[prec]:5s} fwd {f:7.1f} us  dgrad {d:7.1f} us  wgrad {w:7.1f} us   ({byt/f/1e3:5.0f} / {byt/d/1e3:5.0f} / {byt/w/1e3:5.0f} GB/s)", flush=True)
    del sets
