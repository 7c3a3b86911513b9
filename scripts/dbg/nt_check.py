"""bmm_nt (Gram products) and conv1x1_wgrad in fp32 and bf16x3 against fp64 on long reductions (256x256 planes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rcot_amd import lib
from rcot_amd.ops import HipBackend
be = HipBackend()
torch.manual_seed(0)
for (B, hd, c, N) in [(2, 1, 48, 65536), (2, 2, 48, 16384), (2, 1, 48, 16384), (2, 4, 48, 4096), (2, 1, 96, 65536), (1, 1, 48, 65536), (2, 1, 64, 65536)]:
    Q, K = torch.randn(B, hd, c, N, device="cuda"), torch.randn(B, hd, c, N, device="cuda")
    ref = (Q.double() @ K.double().transpose(-1, -2))
    for prec in (0, 1):
        be.prec = prec
        G = torch.zeros(B, hd, c, c, device="cuda")
        be.bmm_nt(Q, K, G)
        torch.cuda.synchronize()
        err = ((G.double() - ref).abs().max() / ref.abs().max()).item()
        print(f"bmm_nt B={B} heads={hd} c={c} N={N} prec={prec}: rel err {err:.2e}", flush=True)
for (B, Co, Ci, N) in [(2, 144, 48, 65536), (2, 48, 127, 65536), (2, 48, 48, 65536), (2, 254, 48, 65536)]:
    dY, X = torch.randn(B, Co, N, device="cuda"), torch.randn(B, Ci, N, device="cuda")
    ref = torch.einsum("bon,bcn->oc", dY.double(), X.double())
    for prec in (0, 1):
        be.prec = prec
        dW = torch.zeros(Co, Ci, device="cuda")
        be.conv1x1_wgrad(dY, X, dW, beta=0.0)
        torch.cuda.synchronize()
        err = ((dW.double() - ref).abs().max() / ref.abs().max()).item()
        print(f"wgrad B={B} {Co}x{Ci} N={N} prec={prec}: rel err {err:.2e}", flush=True)
