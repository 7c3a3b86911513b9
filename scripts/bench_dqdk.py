"""dQ = Eq K + Dq.Q and dK = Eq^T Q + Dk.K of MDTA's backward: the two per-head products of today against ONE
product with the dense symmetric 2C x 2C operand E2 = [[diag Dq, Eq], [Eq^T, diag Dk]] acting on the stacked [Q; K] rows of u.
Graph-replayed timing on every level's shape (B = 8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd import lib
from rcot_amd.ops import HipBackend
be = HipBackend()
be.prec = lib.PREC_BF16X3


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


for (B, heads, c, N) in [(8, 1, 48, 16384), (8, 1, 96, 16384), (8, 2, 48, 4096), (8, 4, 48, 1024), (8, 8, 48, 256)]:
    C = heads * c
    nb = max(2, int(400e6 // (4.0 * B * 3 * C * N)) + 1)
    sets = []
    for _ in range(nb):
        u, du = torch.randn(B, 3 * C, N, device="cuda"), torch.empty(B, 3 * C, N, device="cuda")
        Eq = torch.randn(B, heads, c, c, device="cuda") * 0.1
        EqT = Eq.transpose(-1, -2).contiguous()
        Dq, Dk = torch.randn(B, C, device="cuda"), torch.randn(B, C, device="cuda")
        E2 = torch.zeros(B, 2 * C, 2 * C, device="cuda")
        for h in range(heads):
            r = slice(h * c, (h + 1) * c)
            E2[:, r, C + h * c:C + (h + 1) * c] = Eq[:, h]
            E2[:, C + h * c:C + (h + 1) * c, r] = EqT[:, h]
        E2[:, torch.arange(C), torch.arange(C)] = Dq
        E2[:, C + torch.arange(C), C + torch.arange(C)] = Dk
        sets.append((u, du, Eq, EqT, Dq, Dk, E2))

    def two(u, du, Eq, EqT, Dq, Dk, E2):
        uu, dd = u.view(B, 3, heads, c, N), du.view(B, 3, heads, c, N)
        be.gemm_kmajor(EqT, uu[:, 1], dd[:, 0], c, c, R=uu[:, 0], rowscale=Dq.view(B, heads, c))
        be.gemm_kmajor(Eq, uu[:, 0], dd[:, 1], c, c, R=uu[:, 1], rowscale=Dk.view(B, heads, c))

    def one(u, du, Eq, EqT, Dq, Dk, E2):
        be.gemm_kmajor(E2.unsqueeze(1), u[:, :2 * C].unsqueeze(1), du[:, :2 * C].unsqueeze(1), 2 * C, 2 * C)
    t2 = tm([(lambda s=s: two(*s)) for s in sets])
    t1 = tm([(lambda s=s: one(*s)) for s in sets])
    a, b = torch.empty_like(sets[0][1]), torch.empty_like(sets[0][1])
    s0 = sets[0]
    two(*s0); a.copy_(s0[1]); one(*s0); b.copy_(s0[1])
    err = float((a[:, :2 * C] - b[:, :2 * C]).abs().max() / a[:, :2 * C].abs().max())
    print(f"B={B} heads={heads} c={c} N={N:5d}: two launches {t2:6.1f} us   one dense 2C x 2C launch {t1:6.1f} us   (max diff {err:.1e})", flush=True)
