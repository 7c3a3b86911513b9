"""|mu|/sigma of every LayerNorm input of T_net (per pixel, over channels) at the seeded initialisation and after 10 minimax
iterations on synthetic denoising patches: the quantity the LN fold of the bf16x3 kernels is sensitive to
(tests/test_x3_gpu.py::test_x3_ln_fold_with_large_pixel_mean)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rcot_amd import lib  # noqa: E402
from rcot_amd.net_restormer import F_net, T_net  # noqa: E402
from rcot_amd.ops import default_backend  # noqa: E402
from rcot_amd.synth import make_batch  # noqa: E402
from rcot_amd.trainer import FlatOptimizer, MinimaxStep  # noqa: E402

be = default_backend()
be.prec = lib.PREC_BF16X3
Tn, Fn = T_net(decoder=True, seed=1234), F_net(patch_size=128, seed=1235)
st = MinimaxStep(Tn, Fn, FlatOptimizer(Tn, "RMSprop", 5e-5), FlatOptimizer(Fn, "RMSprop", 1e-4), 1.0, 10000.0)
de = [2] * 4
st.set_de_ids(de)
stats = []
orig = be.ln_stats


def probe(x, mu, rs):
    orig(x, mu, rs)
    r = (mu.abs() * rs).flatten()
    stats.append((float(r.mean()), float(r.quantile(0.999)) if r.numel() < 2 ** 24 else float(r.max()), float(r.max())))


def report(tag):
    _, x, _ = make_batch(4242, 4, 128, de)
    stats.clear()
    be.ln_stats = probe
    Tn(x.cuda())
    be.ln_stats = orig
    t = torch.tensor(stats)
    print(f"{tag}: {len(stats)} LayerNorm inputs; |mu|/sigma mean {float(t[:, 0].mean()):.2f}, worst layer mean {float(t[:, 0].max()):.2f}, "
          f"worst 99.9th percentile {float(t[:, 1].max()):.2f}, worst pixel {float(t[:, 2].max()):.2f}")


report("seeded init")
gen = torch.Generator().manual_seed(1)
for it in range(10):
    _, x, y = make_batch(100 + it, 4, 128, de)
    st.iteration(x.cuda(), y.cuda(), torch.tensor(de, dtype=torch.int32).cuda(), torch.rand(4, generator=gen).cuda(), True)
report("after 10 iterations")
