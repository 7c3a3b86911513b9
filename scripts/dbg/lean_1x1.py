"""Is the lean convolution kernel (k = 1) faster than gemm_xx_kernel on the 1x1 projection shapes?  Cold operands, us per call."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from rcot_amd.ops import HipBackend
from rcot_amd.plan import LaunchPlan
be = HipBackend()
B, H = 8, 128
for (ci, co) in ((96, 510), (96, 288), (255, 96), (510, 96), (96, 96), (288, 96)):
    nb = 3
    Xs = [torch.randn(B, ci, H, H, device="cuda") for _ in range(nb)]
    Ys = [torch.empty(B, co, H, H, device="cuda") for _ in range(nb)]
    W = torch.randn(co, ci, device="cuda") * 0.05
    W4 = W.view(co, ci, 1, 1).contiguous()
    WT, WP = (torch.zeros(*s, device="cuda") for s in be.pack_shapes(co, ci))
    be.pack_weight(W, WT, WP)
    res = []
    for name, fn in (("gemm_xx", lambda i: be.conv1x1_fwd(W, Xs[i].view(B, ci, -1), Ys[i].view(B, co, -1), packed=(WT, WP))),
                     ("lean k1", lambda i: be.conv2d_fwd(Xs[i], W4, None, Ys[i], 1, 0, 1.0, 0, None))):
        for i in range(nb): fn(i)
        torch.cuda.synchronize()
        pl = LaunchPlan(be).record(lambda: [fn(i % nb) for i in range(12)])
        pl.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter(); pl.replay(); torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 12 * 1e6
        res.append(f"{name} {t:7.1f} us {2.0*B*H*H*ci*co/t/1e6:5.1f} TF/s")
    a = Ys[0].clone(); be.conv1x1_fwd(W, Xs[0].view(B, ci, -1), Ys[0].view(B, co, -1), packed=(WT, WP)); torch.cuda.synchronize()
    b_ = Ys[0].clone(); be.conv2d_fwd(Xs[0], W4, None, Ys[0], 1, 0, 1.0, 0, None); torch.cuda.synchronize()
    print(f"{co:4d} <- {ci:4d}: " + "   ".join(res) + f"   max diff {float((Ys[0]-b_).abs().max()):.1e}")
