"""A few launches of the small-level (32x32 / 16x16) projection GEMMs in both arithmetic modes, for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd import lib
from rcot_amd.ops import HipBackend
be = HipBackend()
B = 8
for prec in (lib.PREC_FP32, lib.PREC_BF16X3):
    be.prec = prec
    for (N, Co, Ci) in ((1024, 192, 510), (256, 384, 1021), (256, 2042, 384), (1024, 1020, 192)):
        W = torch.randn(Co, Ci, device="cuda") * 0.1
        X = torch.randn(B, Ci, N, device="cuda"); Y = torch.empty(B, Co, N, device="cuda")
        st, sp = be.pack_shapes(Co, Ci)
        WT, WP = torch.zeros(*st, device="cuda"), torch.zeros(*sp, device="cuda")
        be.pack_weight(W, WT, WP)
        for _ in range(3):
            be.conv1x1_fwd(W, X, Y, packed=(WT, WP))
torch.cuda.synchronize()
