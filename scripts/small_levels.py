"""Wall time per TransformerBlock forward / backward at the three small levels of T_net (B=8 at 128x128 input), launched from a
recorded launch plan (no Python between the kernels): the number a kernel change at those levels has to move.
  python scripts/small_levels.py [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from rcot_amd import lib
from rcot_amd.net_restormer import T_net
from rcot_amd.ops import default_backend
from rcot_amd.plan import LaunchPlan

LEVELS = [("enc1 C48 128", "enc1", 48, 128), ("dec1 C96 128", "dec1", 96, 128), ("enc2 C96 64", "enc2", 96, 64), ("enc3 C192 32", "enc3", 192, 32),
          ("latent C384 16", "latent", 384, 16), ("noise3 C384h4 16", "noise3", 384, 16)]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    be = default_backend()
    be.prec = {"fp32": lib.PREC_FP32, "bf16x3": lib.PREC_BF16X3, "bf16x6": lib.PREC_BF16X6}[os.environ.get("RCOT_GEMM_PREC", "bf16x3")]
    be.x6_packs = True
    Tn = T_net(decoder=True, seed=1234)
    B = int(os.environ.get("BT_BATCH", "8"))
    tot = 0.0
    for label, attr, C, H in LEVELS:
        blk = getattr(Tn, attr)
        blk = blk[0] if isinstance(blk, list) else blk
        x = torch.randn(B, C, H, H, device="cuda")
        d = torch.randn(B, C, H, H, device="cuda")
        y, ctx = blk.forward(x, True)
        blk.backward(ctx, d)
        be.side_join()
        torch.cuda.synchronize()
        box = {}
        pf = LaunchPlan(be).record(lambda: box.__setitem__("c", blk.forward(x, True)[1]))
        pb = LaunchPlan(be).record(lambda: blk.backward(box["c"], d))
        out = []
        for p in (pf, pb):
            for _ in range(3):
                p.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                p.replay()
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / reps * 1e6)
        n = {"enc1": 8, "dec1": 16, "enc2": 26, "enc3": 26, "latent": 16, "noise3": 10}[attr]     # block applications per T_net pass pair (approx.)
        tot += (out[0] + out[1]) * n / 1e3
        print(f"{label:18s} fwd {out[0]:7.1f} us ({pf.n_launches:2d} launches)  bwd {out[1]:7.1f} us ({pb.n_launches:2d})   x{n}: {(out[0] + out[1]) * n / 1e3:6.2f} ms")
    print(f"sum over the unit's 94 block applications ~ {tot:.1f} ms")


if __name__ == "__main__":
    main()
