"""Per-layer time of the critic's convolutions (forward, data gradient, weight gradient) (B = 16, 128x128 input) from a recorded launch plan: us and TFLOP/s."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rcot_amd import params as P
from rcot_amd.ops import HipBackend
from rcot_amd.plan import LaunchPlan
be = HipBackend()
B = int(os.environ.get("CL_B", "16"))
H = 128
out = []
for li, (ci, co, k, s, pad, bias) in enumerate(P.FNET_CONVS):
    Ho = H // s
    X = torch.randn(B, ci, H, H, device="cuda"); Wt = torch.randn(co, ci, k, k, device="cuda") * 0.02
    bv = torch.zeros(co, device="cuda") if bias else None
    Y = torch.empty(B, co, Ho, Ho, device="cuda")
    dZ = torch.randn(B, co, Ho, Ho, device="cuda"); dX = torch.empty_like(X); dW = torch.zeros_like(Wt)
    fl = 2.0 * B * Ho * Ho * ci * co * k * k
    line = f"L{li} {ci:3d}->{co:3d} k{k}s{s} {H:3d}->{Ho:3d}:"
    for name, fn in (("fwd", lambda: be.conv2d_fwd(X, Wt, bv, Y, s, pad, 0.2, 0, None)), ("dgrad", lambda: be.conv2d_dgrad(dZ, Wt, dX, s, pad, 0.0)),
                     ("wgrad", lambda: be.conv2d_wgrad(dZ, X, dW, s, pad, 1.0))):
        fn(); torch.cuda.synchronize()
        pl = LaunchPlan(be).record(lambda: [fn() for _ in range(10)])
        pl.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter(); pl.replay(); torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 10 * 1e6
        line += f"  {name} {t:6.1f} us {fl/t/1e6:5.1f} TF/s"
    out.append(line)
    H = Ho
print("\n".join(out))
