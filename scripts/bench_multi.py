"""dV / dQ / dK of one MDTA block: three rcot_gemm_kmajor launches vs one rcot_gemm_kmajor_multi launch, per level of T_net
(B = 8), re-issued from recorded launch plans (a few us of host per launch, as in the real iteration), 25 x 8 rotating buffer sets
between two HIP events.
  python scripts/bench_multi.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from rcot_amd.ops import default_backend
from rcot_amd.plan import LaunchPlan

LEVELS = [("128x128 C48 h1", 1, 48, 16384), ("128x128 C96 h1", 1, 96, 16384), ("64x64 C96 h2", 2, 48, 4096), ("32x32 C192 h4", 4, 48, 1024),
          ("16x16 C384 h8", 8, 48, 256), ("16x16 C384 h4 (noise3)", 4, 96, 256)]


def main():
    be = default_backend()
    B, reps, nset = 8, 200, 8
    for label, heads, c, N in LEVELS:
        C = heads * c
        sets = []
        for _ in range(nset):
            u, dy, du = torch.randn(B, 3 * C, N, device="cuda"), torch.randn(B, 1, C, N, device="cuda"), torch.empty(B, 3 * C, N, device="cuda")
            Mf, Eq, EqT = torch.randn(B, 1, C, C, device="cuda"), torch.randn(B, heads, c, c, device="cuda"), torch.randn(B, heads, c, c, device="cuda")
            Dq, Dk = torch.randn(B, heads, c, device="cuda"), torch.randn(B, heads, c, device="cuda")
            uu, dd = u.view(B, 3, heads, c, N), du.view(B, 3, heads, c, N)
            sets.append((Mf, dy, du.view(B, 3, C, N)[:, 2].unsqueeze(1), EqT, uu[:, 1], dd[:, 0], uu[:, 0], Dq, Eq, dd[:, 1], Dk))

        def three(s):
            Mf, dy, dV, EqT, K, dQ, Q, Dq, Eq, dK, Dk = s
            be.gemm_kmajor(Mf, dy, dV, C, C)
            be.gemm_kmajor(EqT, K, dQ, c, c, R=Q, rowscale=Dq)
            be.gemm_kmajor(Eq, Q, dK, c, c, R=K, rowscale=Dk)

        def one(s):
            Mf, dy, dV, EqT, K, dQ, Q, Dq, Eq, dK, Dk = s
            assert be.gemm_kmajor_multi([(Mf, dy, dV, C, C, None, None), (EqT, K, dQ, c, c, Q, Dq), (Eq, Q, dK, c, c, K, Dk)])
        out = []
        for fn in (three, one):
            for s in sets:
                fn(s)
            torch.cuda.synchronize()
            plan = LaunchPlan(be).record(lambda: [fn(s) for s in sets])
            for _ in range(3):
                plan.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(reps // nset):
                plan.replay()
            b.record()
            torch.cuda.synchronize()
            out.append(a.elapsed_time(b) / (reps // nset * nset) * 1e3)
        print(f"{label:26s} three launches {out[0]:7.1f} us   one launch {out[1]:7.1f} us")


if __name__ == "__main__":
    main()
