"""The launch bench.py names as dominant at BASELINE configs[1] in round 3 — y = (W_o blockdiag(A)) V + x of one level-1 block,
gemm_kmajor((8, 1, 96, 96), (8, 1, 96, 16384), (8, 1, 96, 16384)), bf16x3 — three launches for the rocprofv3 --pmc passes
(scripts/rocprof_traffic.sh with PMC_SCRIPT=pmc_apply.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd import lib
from rcot_amd.ops import HipBackend
be = HipBackend()
be.prec = lib.PREC_BF16X3
B, C, N = 8, 96, 16384
for _ in range(3):
    MfT = torch.randn(B, C, C, device="cuda") * 0.1
    V, x, y = torch.randn(B, 1, C, N, device="cuda"), torch.randn(B, 1, C, N, device="cuda"), torch.empty(B, 1, C, N, device="cuda")
    be.gemm_kmajor(MfT.unsqueeze(1), V, y, C, C, R=x)
torch.cuda.synchronize()
