"""ONE TransformerBlock forward (and backward: PMC_BWD=1) at C = 96, 128x128, B = 8, three times, for the FETCH_SIZE / WRITE_SIZE
counter passes (scripts/rocprof_traffic.sh with PMC_SCRIPT=pmc_block.py): the per-kernel table sums to the block's HBM traffic,
to set against the stage model's algorithmic bytes — forward (8C + 3h) N 4 B per image, C = 96, h = 255."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcot_amd import lib
from rcot_amd.net_restormer import T_net
from rcot_amd.ops import default_backend
be = default_backend()
be.prec = lib.PREC_BF16X3
Tn = T_net(decoder=True, seed=1234)
blk = Tn.dec1[0]
B, C, H = 8, 96, 128
x = torch.randn(B, C, H, H, device="cuda")
d = torch.randn(B, C, H, H, device="cuda")
for _ in range(3):
    y, ctx = blk.forward(x, True)
    if os.environ.get("PMC_BWD") == "1":
        blk.backward(ctx, d)
        be.side_join()
torch.cuda.synchronize()
