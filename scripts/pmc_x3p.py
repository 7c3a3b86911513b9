"""A few launches of the producer / consumer bf16x3 projection (x3p_kernel: pre-split weight packs) on the 128x128-level shapes,
for rocprofv3 --pmc passes (tiny on purpose): 510<-96 +LN, 288<-96 +LN, 96<-510 +residual, 96<-96."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd import lib
from rcot_amd.ops import HipBackend
be = HipBackend()
be.prec = lib.PREC_BF16X3
B, N = 8, 16384
flush = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
for (Co, Ci, ln, res) in ((510, 96, True, False), (288, 96, True, False), (96, 510, False, True), (96, 96, False, False)):
    W = torch.randn(Co, Ci, device="cuda") * 0.1
    X = torch.randn(B, Ci, 128, 128, device="cuda"); Y = torch.empty(B, Co, 128, 128, device="cuda")
    R = torch.randn(B, Co, 128, 128, device="cuda") if res else None
    WT, WP = (torch.zeros(*s, device="cuda") for s in be.pack_shapes(Co, Ci))
    WTf, c12 = (torch.zeros(*s, device="cuda") for s in be.fold_shapes(Co, Ci))
    (st,), (sp,) = be.split_shapes(Co, Ci)
    WTs, WPs, WTfs = torch.zeros(st, device="cuda"), torch.zeros(sp, device="cuda"), torch.zeros(st, device="cuda")
    mu, rs = torch.zeros(B, N, device="cuda"), torch.ones(B, N, device="cuda")
    lw, lb = torch.ones(Ci, device="cuda"), torch.zeros(Ci, device="cuda")
    be.pack_weight(W, WT, WP, (lw, lb, WTf, c12), (WTs, WPs, WTfs))
    be.ln_stats(X, mu, rs)
    for _ in range(3):
        flush.fill_(1)                                                   # operands cold in the Infinity Cache
        be.conv1x1_fwd(W, X, Y, ln=(mu, rs, lw, lb) if ln else None, R=R, packed=(WT, WP, (WTf, c12), (WTs, WPs, WTfs)))
torch.cuda.synchronize()
