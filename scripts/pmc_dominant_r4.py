"""A few launches of the kernels bench.py names as dominant SYMBOLS at BASELINE configs[1] in round 4, on their 128x128-level shapes
with operands cold in the Infinity Cache, for rocprofv3 --pmc passes (scripts/rocprof_traffic.sh):
  exact fp32   gemm_xx_kernel<96, 128, 1, 4, false>: the 1x1 data gradients 96 <- 510 and 96 <- 288 (and the MDTA apply 96 <- 96)
  bf16x3       x3p_kernel<true, false, 2, 4, false, true, 2>: LayerNorm + projection 510 <- 96 and 288 <- 96 (statistics made in-kernel)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd import lib
from rcot_amd.ops import HipBackend
be = HipBackend()
B, N = 8, 16384
flush = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
for (Co, Ci) in ((510, 96), (288, 96)):
    W = torch.randn(Co, Ci, device="cuda") * 0.1
    X = torch.randn(B, Ci, 128, 128, device="cuda"); Y = torch.empty(B, Co, 128, 128, device="cuda")
    dY = torch.randn(B, Co, 128, 128, device="cuda"); dX = torch.empty(B, Ci, 128, 128, device="cuda")
    WT, WP = (torch.zeros(*s, device="cuda") for s in be.pack_shapes(Co, Ci))
    WTf, c12 = (torch.zeros(*s, device="cuda") for s in be.fold_shapes(Co, Ci))
    (st,), (sp,) = be.split_shapes(Co, Ci)
    WTs, WPs, WTfs = torch.zeros(st, device="cuda"), torch.zeros(sp, device="cuda"), torch.zeros(st, device="cuda")
    mu, rs = torch.zeros(B, N, device="cuda"), torch.ones(B, N, device="cuda")
    lw, lb = torch.ones(Ci, device="cuda"), torch.zeros(Ci, device="cuda")
    pk = (WT, WP, (WTf, c12), (WTs, WPs, WTfs))
    be.pack_weight(W, WT, WP, (lw, lb, WTf, c12), (WTs, WPs, WTfs))
    for _ in range(3):
        be.prec = lib.PREC_FP32
        flush.fill_(1)
        be.conv1x1_dgrad(W, dY, dX, packed=pk)                               # gemm_xx_kernel<96, 128, 1, 4, false>
        be.prec = lib.PREC_BF16X3
        flush.fill_(1)
        be.conv1x1_fwd(W, X, Y, ln=(mu, rs, lw, lb), packed=pk, ln_compute=True)   # x3p_kernel<true, false, 2, 4, false, true, 2>
torch.cuda.synchronize()
