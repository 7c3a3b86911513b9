"""Three cold launches each of the kernels bench.py names as dominant SYMBOLS at BASELINE configs[1] in round 5, for rocprofv3 --pmc
passes (scripts/rocprof_traffic.sh): the 1x1 weight gradient with the LayerNorm recomputed in the loop, 510 <- 96 and 288 <- 96 at
8 x 128x128, left as split-K slabs exactly as the schedule launches it (rcot_conv1x1_wgrad_slabs):
  exact fp32  gemm_nt_kernel<1, 3, 4, 1, true, false, false, 0>     bf16x6  gemm_nt_kernel<1, 3, 4, 1, true, true, true, 1>
(the bf16x3 symbol, x3p_kernel<true, false, 2, 4, false, true, 2>, was counted in round 4: profiles/r04_pmc_traffic_dominant.txt)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd import lib
from rcot_amd.ops import HipBackend
B, N = 8, 16384
flush = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
for prec in (lib.PREC_FP32, lib.PREC_BF16X6):
    be = HipBackend()
    be.prec = prec
    be.x6_packs = prec == lib.PREC_BF16X6
    for (Co, Ci) in ((510, 96), (288, 96)):
        X = torch.randn(B, Ci, 128, 128, device="cuda")
        dY = torch.randn(B, Co, 128, 128, device="cuda")
        dW = torch.zeros(Co, Ci, device="cuda")
        mu, rs = torch.zeros(B, N, device="cuda"), torch.ones(B, N, device="cuda")
        lw, lb = torch.ones(Ci, device="cuda"), torch.zeros(Ci, device="cuda")
        for _ in range(3):
            flush.fill_(1)
            d = be.conv1x1_wgrad_slabs(dY, X, dW, ln=(mu, rs, lw, lb))
            assert d is not None
            print(Co, Ci, "slabs", d[1], flush=True)
torch.cuda.synchronize()
