"""rcot_attn_core_fwd on the small-level shapes, many calls back to back (run under rocprofv3 --kernel-trace --stats)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rcot_amd.ops import default_backend  # noqa: E402

be = default_backend()
reps = int(os.environ.get("REPS", "50"))
SH = [(8, 8, 48, 256), (8, 4, 48, 1024), (8, 2, 48, 4096), (8, 4, 96, 256)]
if os.environ.get("SHAPE"):
    SH = [SH[int(os.environ["SHAPE"])]]
for B, heads, c, N in SH:
    C = heads * c
    u = torch.randn(B, 3 * C, 16, N // 16, device="cuda")
    temp = torch.ones(heads, device="cuda")
    WoT = torch.randn(C, C, device="cuda") * 0.1
    sq, Gn, A, MfT = be.empty(B, 2 * C), be.empty(B, heads, c, c), be.empty(B, heads, c, c), be.empty(B, C, C)
    for _ in range(reps):
        assert be.attn_core_fwd(u, temp, WoT, sq, Gn, A, MfT)
    torch.cuda.synchronize()
