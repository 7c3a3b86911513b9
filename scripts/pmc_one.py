"""One K-major projection shape (env SHAPE="B,N,Co,Ci", PREC=x3|fp32), three launches, for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd import lib
from rcot_amd.ops import HipBackend
be = HipBackend()
be.prec = lib.PREC_BF16X3 if os.environ.get("PREC", "x3") == "x3" else lib.PREC_FP32
B, N, Co, Ci = (int(v) for v in os.environ.get("SHAPE", "8,4096,96,512").split(","))
W = torch.randn(Co, Ci, device="cuda") * 0.1
st, sp = be.pack_shapes(Co, Ci)
WT, WP = torch.zeros(*st, device="cuda"), torch.zeros(*sp, device="cuda")
be.pack_weight(W, WT, WP)
for _ in range(3):
    X = torch.randn(B, Ci, N, device="cuda"); Y = torch.empty(B, Co, N, device="cuda")
    be.conv1x1_fwd(W, X, Y, packed=(WT, WP))
torch.cuda.synchronize()
