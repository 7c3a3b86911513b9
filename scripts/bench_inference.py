"""Whole-image inference of both transport maps (what rcot_amd/tester.py and trainer.evaluate() run): milliseconds and megapixels per
second by image size, from a launch plan (no Python between the kernels), plus the tester's tiled form on the largest size.
  python scripts/bench_inference.py [reps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from rcot_amd.mprnet_hip import MPRNetHip
from rcot_amd.net_restormer import T_net
from rcot_amd.ops import default_backend
from rcot_amd.plan import LaunchPlan
from rcot_amd.tester import restore


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    be = default_backend()
    nets = {"restormer (two-pass T_net)": (T_net(decoder=True, seed=1), 8), "mprnet (Net.T_net)": (MPRNetHip(backend=be, seed=1), 4)}
    out = {}
    for name, (net, mult) in nets.items():
        rows = []
        for H, W in ((256, 256), (512, 512), (720, 1280)):
            x = torch.rand(1, 3, H, W, device="cuda")
            net(x)
            torch.cuda.synchronize()
            plan = LaunchPlan(be).record(lambda: net(x))
            for _ in range(2):
                plan.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                plan.replay()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            rows.append({"image": f"{H}x{W}", "ms": round(ms, 2), "megapixels_per_s": round(H * W / ms / 1e3, 1), "launches": plan.n_launches,
                         "peak_mem_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)})
            del plan
            torch.cuda.reset_peak_memory_stats()
        x = torch.rand(1, 3, 720, 1280, device="cuda")
        restore(net, x, 512, 32, mult)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            restore(net, x, 512, 32, mult)
        torch.cuda.synchronize()
        rows.append({"image": "720x1280 as 512x512 tiles, overlap 32 (tester --tile 512; eager launches)", "ms": round((time.perf_counter() - t0) / 3 * 1e3, 2)})
        out[name] = rows
    print(json.dumps(out))


if __name__ == "__main__":
    main()
