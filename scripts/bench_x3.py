"""Cold-buffer timing of the K-major projection GEMM (rcot_gemm_kmajor) in both arithmetic modes on the level-1/2 shapes of
the transport map: each call works on the next of NBUF operand sets so that nothing is served from the 256 MB Infinity
Cache.  Prints us, fp32-equivalent TFLOP/s and algorithmic GB/s.  X3_ONLY=1: only the bf16x3 mode (library A/B runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd import lib
from rcot_amd.ops import HipBackend
be = HipBackend()
SHAPES = [(8, 16384, 510, 96, True, False), (8, 16384, 288, 96, True, False), (8, 16384, 96, 255, False, True),
          (8, 16384, 96, 96, False, True), (8, 16384, 96, 510, False, False), (8, 16384, 96, 288, False, False),
          (8, 16384, 144, 48, True, False), (8, 16384, 48, 127, False, True),
          (8, 4096, 510, 96, True, False), (8, 4096, 96, 255, False, True), (8, 1024, 1020, 192, True, False),
          (8, 1024, 192, 510, False, True),
          (8, 256, 384, 1021, False, True), (8, 256, 384, 2042, False, False), (8, 1024, 192, 1020, False, False),
          (8, 256, 2042, 384, True, False), (8, 256, 384, 1152, False, False), (8, 1024, 576, 192, True, False),
          (8, 4096, 96, 510, False, False), (8, 4096, 96, 288, False, False), (8, 1024, 192, 576, False, False)]
if os.environ.get("X3_SHAPES"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["X3_SHAPES"].split(",")]
PRECS = (lib.PREC_BF16X3,) if os.environ.get("X3_ONLY") else (lib.PREC_FP32, lib.PREC_BF16X6, lib.PREC_BF16X3)
NAMES = {lib.PREC_FP32: "fp32", lib.PREC_BF16X3: "x3", lib.PREC_BF16X6: "x6"}
def tm(fs, reps=24):
    """reps calls captured into ONE HIP graph and replayed: the GPU-side time per call (an eager loop measures the ~20 us
    of Python/ctypes per launch for anything shorter than that)"""
    for f in fs: f()
    torch.cuda.synchronize()
    g, st = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.graph(g, stream=st):
        for i in range(reps): fs[i % len(fs)]()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for (B, N, Co, Ci, ln, res) in SHAPES:
    byt = 4.0 * B * N * (Ci + Co * (2 if res else 1))
    nbuf = max(2, int(600e6 // byt) + 1)
    W = torch.randn(Co, Ci, device="cuda") * 0.1
    st, sp = be.pack_shapes(Co, Ci)
    WT, WP = torch.zeros(*st, device="cuda"), torch.zeros(*sp, device="cuda")
    lw, lb = torch.ones(Ci, device="cuda"), torch.zeros(Ci, device="cuda")
    WTf, c12 = (torch.zeros(*s_, device="cuda") for s_ in be.fold_shapes(Co, Ci))
    WTs, WPs, WTfs = (torch.zeros(*be.split_shapes(Co, Ci)[i], device="cuda") for i in (0, 1, 0))
    sp3 = None if os.environ.get("X3_NOSPLIT") else (WTs, WPs, WTfs)
    sp6 = tuple(torch.zeros(*be.split6_shapes(Co, Ci)[i], device="cuda") for i in (0, 1, 0))
    be.pack_weight(W, WT, WP, (lw, lb, WTf, c12), sp3, sp6)
    sets = []
    for _ in range(nbuf):
        X = torch.randn(B, Ci, N, device="cuda"); Y = torch.empty(B, Co, N, device="cuda")
        R = torch.randn(B, Co, N, device="cuda") if res else None
        mu, rs = torch.zeros(B, N, device="cuda"), torch.ones(B, N, device="cuda")
        sets.append((X, Y, R, mu, rs))
    out = []
    for prec in PRECS:
        be.prec = prec
        fs = [(lambda X=X, Y=Y, R=R, mu=mu, rs=rs: be.conv1x1_fwd(W, X, Y, ln=(mu, rs, lw, lb) if ln else None, R=R, packed=(WT, WP, (WTf, c12), sp3, sp6)))
              for (X, Y, R, mu, rs) in sets]
        ms = tm(fs)
        out.append(f"{NAMES[prec]}: {ms*1e3:7.1f} us {2.0*Co*Ci*B*N/ms/1e9:6.1f} TF {byt/ms/1e6:6.0f} GB/s")
    print(f"B={B} N={N:5d} M={Co:4d} K={Ci:4d} ln={int(ln)} res={int(res)} nbuf={nbuf}:  " + "   ".join(out), flush=True)
    del sets
