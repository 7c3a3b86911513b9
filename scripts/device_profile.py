"""Per-kernel in-situ device time of ONE replayed minimax iteration from the library's own per-launch time stamps
(rcot_profile_begin / rcot_profile_end: hipExtLaunchKernelGGL start / stop events) — the table `rocprofv3 --kernel-trace --stats` gives,
without the tool.   python scripts/device_profile.py [fp32|bf16x6|bf16x3]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from rcot_amd import lib
from rcot_amd.net_restormer import F_net, T_net
from rcot_amd.ops import PREC_BY_NAME, default_backend
from rcot_amd.synth import make_batch
from rcot_amd.trainer import FlatOptimizer, MinimaxStep


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    be = default_backend()
    be.prec = PREC_BY_NAME[prec]
    be.x6_packs = prec == "bf16x6"
    B, P, de = 8, 128, [2] * 8
    Tn, Fn = T_net(decoder=True, seed=1234), F_net(patch_size=P, seed=1235)
    st = MinimaxStep(Tn, Fn, FlatOptimizer(Tn, "RMSprop", 5e-5), FlatOptimizer(Fn, "RMSprop", 1e-4), 1.0, 10000.0)
    st.set_de_ids(de)
    _, x, y = make_batch(1, B, P, de)
    x, y, d, a = x.cuda(), y.cuda(), torch.tensor(de, dtype=torch.int32).cuda(), torch.rand(B).cuda()
    for _ in range(3):
        st.run(x, y, d, a, True)
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 18)
    be.L.rcot_profile_begin()
    st.run(x, y, d, a, True)
    torch.cuda.synchronize()
    n = be.L.rcot_profile_end(buf, 1 << 18)
    rows = [l.rsplit("|", 2) for l in buf.value.decode(errors="replace").splitlines()]
    tot = sum(float(r[2]) for r in rows)
    print(f"# {prec}: {n} launches, {tot:.2f} ms of kernel time in one replayed iteration; kernel | launches | ms | avg us")
    for name, c, ms in rows[:70]:
        print(f"{name[:150]} | {c} | {float(ms):.3f} | {float(ms) / max(int(c), 1) * 1e3:.2f}")


if __name__ == "__main__":
    main()
