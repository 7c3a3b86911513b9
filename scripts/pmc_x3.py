"""A few launches of the bf16x3 K-major projection on level-1/2 shapes, for a rocprofv3 --pmc pass (tiny on purpose)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd import lib
from rcot_amd.ops import HipBackend
be = HipBackend()
be.prec = lib.PREC_BF16X3 if os.environ.get("PREC", "x3") == "x3" else lib.PREC_FP32
B, N = 8, 16384
for (Co, Ci, ln) in ((510, 96, True), (288, 96, True), (96, 510, False)):
    W = torch.randn(Co, Ci, device="cuda") * 0.1
    X = torch.randn(B, Ci, 128, 128, device="cuda"); Y = torch.empty(B, Co, 128, 128, device="cuda")
    st, sp = be.pack_shapes(Co, Ci)
    WT, WP = torch.zeros(*st, device="cuda"), torch.zeros(*sp, device="cuda")
    mu, rs = torch.zeros(B, N, device="cuda"), torch.ones(B, N, device="cuda")
    lw, lb = torch.ones(Ci, device="cuda"), torch.zeros(Ci, device="cuda")
    WTf, c12 = (torch.zeros(*s_, device="cuda") for s_ in be.fold_shapes(Co, Ci))
    be.pack_weight(W, WT, WP, (lw, lb, WTf, c12))
    for _ in range(3):
        be.conv1x1_fwd(W, X, Y, ln=(mu, rs, lw, lb) if ln else None, packed=(WT, WP, (WTf, c12)))
torch.cuda.synchronize()
