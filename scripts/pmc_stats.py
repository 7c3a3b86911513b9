"""Counter-pass script (round 6): the attention apply / project_out of a C = 96 block at 8 x 128x128 as the plain exact-fp32 product
followed by rcot_ln_stats, and as ONE rcot_gemm_kmajor_stats launch — three cold launches each (scripts/rocprof_traffic.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd import lib
from rcot_amd.ops import HipBackend
be = HipBackend()
be.prec = lib.PREC_FP32
B, N, C, K = 8, 16384, 96, 255
W = torch.randn(C, K, device="cuda") * 0.1
st, sp = be.pack_shapes(C, K)
WT, WP = torch.zeros(*st, device="cuda"), torch.zeros(*sp, device="cuda")
be.pack_weight(W, WT, WP)
for _ in range(3):
    X = torch.randn(B, K, N, device="cuda"); Y = torch.empty(B, C, N, device="cuda"); R = torch.randn(B, C, N, device="cuda")
    mu, rs = torch.empty(B, N, device="cuda"), torch.empty(B, N, device="cuda")
    be.conv1x1_fwd(W, X, Y, R=R, packed=(WT, WP))
    be.ln_stats(Y, mu, rs)
for _ in range(3):
    X = torch.randn(B, K, N, device="cuda"); Y = torch.empty(B, C, N, device="cuda"); R = torch.randn(B, C, N, device="cuda")
    mu, rs = torch.empty(B, N, device="cuda"), torch.empty(B, N, device="cuda")
    be.conv1x1_fwd(W, X, Y, R=R, packed=(WT, WP), stats=(mu, rs))
torch.cuda.synchronize()
