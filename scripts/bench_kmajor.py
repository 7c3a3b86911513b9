"""Hot-buffer timing of the K-major LDS-DMA GEMM on the level-1 projection shapes (RCOT_BW=1 adds copy/fill/sum reference
bandwidths).  NOTE: operands that fit the 256 MB Infinity Cache plus DVFS make this optimistic; decisions were taken with
the in-situ A/B (scripts/ab_env.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd.ops import HipBackend
be = HipBackend()
B, N = 8, 16384
def tm(f, reps=20):
    for _ in range(3): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
if os.environ.get("RCOT_BW"):
    for mb in (64, 256, 1024):
        n = mb * (1 << 20) // 4
        a, b, c = torch.randn(n, device="cuda"), torch.randn(n, device="cuda"), torch.empty(n, device="cuda")
        t = tm(lambda: c.copy_(a)); print(f"copy  {mb:5d} MiB: {2*n*4/t/1e9:7.1f} GB/s (r+w)")
        t = tm(lambda: torch.add(a, b, out=c)); print(f"add   {mb:5d} MiB: {3*n*4/t/1e9:7.1f} GB/s (2r+w)")
        t = tm(lambda: c.fill_(1.0)); print(f"fill  {mb:5d} MiB: {n*4/t/1e9:7.1f} GB/s (w)")
        t = tm(lambda: a.sum()); print(f"sum   {mb:5d} MiB: {n*4/t/1e9:7.1f} GB/s (r)")
for (Co, Ci, ln, res) in ((510, 96, True, False), (288, 96, True, False), (96, 255, False, True), (96, 96, False, True), (96, 510, False, False), (96, 288, False, False)):
    W = torch.randn(Co, Ci, device="cuda") * 0.1
    X = torch.randn(B, Ci, N, device="cuda")
    Y = torch.empty(B, Co, N, device="cuda")
    R = torch.randn(B, Co, N, device="cuda")
    st, sp = be.pack_shapes(Co, Ci)
    WT, WP = torch.zeros(*st, device="cuda"), torch.zeros(*sp, device="cuda")
    be.pack_weight(W, WT, WP)
    mu, rs = torch.zeros(B, N, device="cuda"), torch.ones(B, N, device="cuda")
    lw, lb = torch.ones(Ci, device="cuda"), torch.zeros(Ci, device="cuda")
    f = lambda: be.conv1x1_fwd(W, X, Y, ln=(mu, rs, lw, lb) if ln else None, R=R if res else None, packed=(WT, WP))
    ms = tm(f)
    byt = 4.0 * B * N * (Ci + Co * (2 if res else 1))
    print(f"M={Co:4d} K={Ci:4d} ln={int(ln)}: {ms*1e3:7.1f} us  {2.0*Co*Ci*B*N/ms/1e9:6.1f} TF/s  {byt/ms/1e6:6.0f} GB/s")
