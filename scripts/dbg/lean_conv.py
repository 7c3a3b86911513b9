import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from rcot_amd.ops import HipBackend
be = HipBackend()
torch.manual_seed(0)
B, ci, co, H, k, s, pad = 1, 64, 64, 16, 3, 1, 1
X = torch.randn(B, ci, H, H, device="cuda"); Wt = torch.randn(co, ci, k, k, device="cuda") * 0.05
Y = torch.zeros(B, co, H // s, H // s, device="cuda")
be.conv2d_fwd(X, Wt, None, Y, s, pad, 1.0, 0, None)
ref = torch.nn.functional.conv2d(X.double(), Wt.double(), None, s, pad).float()
torch.cuda.synchronize()
e = (Y - ref).abs()
print("max err", float(e.max()), "ref max", float(ref.abs().max()))
print("err per out channel (first 8):", e.amax(dim=(0, 2, 3))[:8].tolist())
print("err per row y:", e.amax(dim=(0, 1, 3)).tolist())
print("err per col x:", e.amax(dim=(0, 1, 2)).tolist())
# probe: impulse input / impulse weights
X2 = torch.zeros_like(X); X2[0, 5, 7, 9] = 1.0
Y2 = torch.zeros_like(Y)
be.conv2d_fwd(X2, Wt, None, Y2, s, pad, 1.0, 0, None)
ref2 = torch.nn.functional.conv2d(X2, Wt, None, s, pad)
torch.cuda.synchronize()
nz = (Y2 != 0).nonzero()[:12].tolist(); nzr = (ref2 != 0).nonzero()[:12].tolist()
print("impulse nz got", nz); print("impulse nz ref", nzr)
print("impulse max err", float((Y2 - ref2).abs().max()))
