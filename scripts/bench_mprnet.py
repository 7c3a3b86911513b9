"""BASELINE configs[0] on an MI355X: the MPRNet transport map + F_net(128), B = 4, 128 x 128, de_type single (L1-spectrum
cost), unpaired, RMSprop — one minimax iteration three ways on the same box:
  hip        MPRNetHip + the HIP critic through MinimaxStep (launch-plan replay), and the transport map's forward + backward alone;
  torch_gpu  the stock-ops loop of rcot_amd/mprnet.py (torch autograd, MIOpen / rocBLAS / hipFFT) on the same GPU;
  torch_cpu  the same loop on the host cores (what BASELINE.json describes for this configuration), a bounded sample.
Prints ONE JSON line.   python scripts/bench_mprnet.py [steps] [--no-cpu | --hip-only]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from rcot_amd import mprnet as MP
from rcot_amd.mprnet_hip import MPRNetHip
from rcot_amd.net_restormer import F_net
from rcot_amd.ops import default_backend
from rcot_amd.plan import LaunchPlan
from rcot_amd.synth import make_batch
from rcot_amd.trainer import FlatOptimizer, MinimaxStep


def mprnet_fwd_flop(H, W):
    """MFMA work of one forward of Net.T_net on one H x W image (3x3 and 1x1 convolutions; the channel-attention gates are O(C^2))"""
    n1, n2, n3, hw = 80, 128, 176, H * W
    c3 = lambda ci, co, px: 2.0 * ci * co * 9 * px
    c1 = lambda ci, co, px: 2.0 * ci * co * px
    fl = 2 * c3(3, n1, hw)                                       # shallow_feat1.0, res_shallow_feat1.0
    fl += 12 * 2 * c3(n1, n1, hw)                                # 12 CAB applications at level 1 (decoder + skip_attn1 in both passes)
    fl += 10 * 2 * c3(n2, n2, hw / 4) + 8 * 2 * c3(n3, n3, hw / 16)
    fl += 2 * (c1(n1, n2, hw / 4) + c1(n2, n3, hw / 16))         # DownSample 1x1 of the two encoders
    fl += 2 * (c1(n3, n2, hw / 16) + c1(n2, n1, hw / 4))         # SkipUpSample 1x1 (in front of the resampling), both decoder passes
    return fl + 2 * c1(n1, 3, hw)                                # SAM's image convolution, both passes


MFMA_F32_PEAK_TF = 157.3       # MI355X_MICROARCH.md: dense fp32 MFMA


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
    B, P, de, lr = int(os.environ.get("MPR_B", "4")), int(os.environ.get("MPR_P", "128")), 7, 1e-4
    be = default_backend()
    Tn, Fn = MPRNetHip(backend=be, seed=1234), F_net(P, backend=be, seed=1235)
    st = MinimaxStep(Tn, Fn, FlatOptimizer(Tn, "RMSprop", lr / 2), FlatOptimizer(Fn, "RMSprop", lr), 1.0, 10000.0)
    st.set_de_ids([de] * B)
    de_dev = torch.tensor([de] * B, dtype=torch.int32, device="cuda")
    batches = []
    for i in range(4):
        _, x, y = make_batch(1001000 + i, B, P, [de] * B, unpaired=True)
        batches.append((x.cuda(), y.cuda()))
    gen = torch.Generator().manual_seed(77)
    alphas = [torch.rand(B, generator=gen).cuda() for _ in range(4)]

    def run(n, first=0):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(first, first + n):
            x, y = batches[i % 4]
            st.run(x, y, de_dev, alphas[i % 4], False)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    run(3)
    ms_hip = run(steps, 3)
    plans = st.planned is not None and st.planned.enabled
    n_launch = [e["plan"].n_launches for e in st.planned.cache.values()] if plans else None
    # the transport map alone: forward (activations kept) + backward, from a launch plan
    x = batches[0][0]
    d = torch.randn_like(x)
    Tn.zero_grad()
    Tn.forward(x, save=True)
    Tn.backward(d)
    torch.cuda.synchronize()
    pf = LaunchPlan(be).record(lambda: (Tn.forward(x, save=True), Tn.backward(d)))
    for _ in range(3):
        pf.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pf.replay()
    torch.cuda.synchronize()
    ms_unit = (time.perf_counter() - t0) / steps * 1e3
    pi = LaunchPlan(be).record(lambda: Tn.forward(x, save=False))
    for _ in range(3):
        pi.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pi.replay()
    torch.cuda.synchronize()
    ms_inf = (time.perf_counter() - t0) / steps * 1e3
    # the dominant kernel symbol of one replayed iteration, from the library's own per-launch device time stamps (rcot_profile_begin / _end)
    import ctypes
    buf = ctypes.create_string_buffer(1 << 18)
    torch.cuda.synchronize()
    be.L.rcot_profile_begin()
    st.run(batches[0][0], batches[0][1], de_dev, alphas[0], False)
    torch.cuda.synchronize()
    be.L.rcot_profile_end(buf, 1 << 18)
    rows = []
    for ln in buf.value.decode(errors="replace").splitlines():
        parts = ln.rsplit("|", 2)
        if len(parts) == 3 and not parts[0].startswith("#"):
            rows.append((float(parts[2]), int(parts[1]), parts[0].strip()))
    rows.sort(reverse=True)
    fl_unit = 3.0 * mprnet_fwd_flop(P, P) * B                    # forward + data gradients + weight gradients
    roof = {"bound": "mfma", "unit_of_work": f"transport map forward + backward, B={B}, {P}x{P}: {fl_unit / 1e9:.1f} GFLOP of fp32 MFMA work "
            f"(3 x {mprnet_fwd_flop(P, P) / 1e9:.2f} GFLOP forward per image; arithmetic intensity > 100 FLOP/B: MFMA-bound)",
            "achieved": round(fl_unit / (ms_unit * 1e-3) / 1e12, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
            "frac": round(fl_unit / (ms_unit * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4),
            "dominant_kernels_of_one_iteration": [{"kernel": k[:120], "launches": n, "ms": round(t, 3)} for t, n, k in rows[:6]],
            "kernel_ms_of_one_iteration": round(sum(r[0] for r in rows), 2)}
    out = {"workload": f"BASELINE configs[0]: MPRNet Net.T_net + F_net({P}), B={B}, {P}x{P}, de_id 7, unpaired, RMSprop", "steps": steps, "roofline": roof,
           "hip": {"ms_per_iteration": round(ms_hip, 2), "patches_per_s": round(B / ms_hip * 1e3, 1), "launches": n_launch,
                   "tnet_fwd_bwd_ms": round(ms_unit, 2), "tnet_fwd_bwd_launches": pf.n_launches, "tnet_inference_ms": round(ms_inf, 2)}}
    if "--hip-only" in sys.argv:
        print(json.dumps(out))
        return
    # stock ops on the same GPU
    Tm, Fm = MP.MPRNetT(seed=1234, device="cuda"), MP.FNetTorch(P, seed=1235, device="cuda")
    To, Fo = torch.optim.RMSprop(Tm.parameters(), lr=lr / 2), torch.optim.RMSprop(Fm.parameters(), lr=lr)

    def run_t(Tm, Fm, To, Fo, n, dev):
        if dev == "cuda":
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            x, y = batches[i % 4]
            MP.torch_minimax_iteration(Tm, Fm, To, Fo, x.to(dev), y.to(dev), [de] * B, alphas[i % 4].to(dev), 1.0, 10000.0, False)
        if dev == "cuda":
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    run_t(Tm, Fm, To, Fo, 2, "cuda")
    ms_t = run_t(Tm, Fm, To, Fo, max(3, steps // 2), "cuda")
    out["torch_gpu"] = {"ms_per_iteration": round(ms_t, 2), "patches_per_s": round(B / ms_t * 1e3, 1),
                        "note": "rcot_amd/mprnet.py: stock PyTorch-ROCm ops + autograd (float() reads of the losses every iteration, as the loop has them)"}
    out["hip_vs_torch_gpu"] = round(ms_t / ms_hip, 2)
    if "--no-cpu" not in sys.argv:
        threads = min(os.cpu_count() or 1, 32)
        torch.set_num_threads(threads)
        Tc, Fc = MP.MPRNetT(seed=1234), MP.FNetTorch(P, seed=1235)
        Toc, Foc = torch.optim.RMSprop(Tc.parameters(), lr=lr / 2), torch.optim.RMSprop(Fc.parameters(), lr=lr)
        run_t(Tc, Fc, Toc, Foc, 1, "cpu")
        ms_c = run_t(Tc, Fc, Toc, Foc, 2, "cpu")
        out["torch_cpu"] = {"ms_per_iteration": round(ms_c, 1), "patches_per_s": round(B / ms_c * 1e3, 2), "threads": threads,
                            "sample": "1 warm-up + 2 timed iterations"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
