"""GPU tier, whole-path parity: the HIP transport map / critic / minimax step against
 (a) golden fixtures produced by the REFERENCE itself (tests/golden/*.npz, oracle/pin_against_reference.py),
 (b) reference outputs for further seeded inputs (gpu_fixtures.npz); nothing here runs the oracle on the GPU box's host, and
 (c) size-independent properties at BASELINE.json's full size (B=8, 128x128).
Tolerances follow BASELINE.json: forward <= 1e-3 relative fp32 (we assert 1e-4); gradients 2e-3 of their norm.
"""
import numpy as np
import pytest
import torch

from conftest import relerr, seeded_tensor
from rcot_amd import params as P

pytestmark = pytest.mark.gpu


def _np_params(shapes, seed, kind):
    return {k: torch.from_numpy(v) for k, v in P.seeded_params(shapes, seed, kind).items()}


def _strided(t, n=64):
    f = t.detach().reshape(-1).cpu()
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long()
    return f[idx].numpy()


@pytest.fixture(scope="module")
def tnet():
    from rcot_amd.net_restormer import T_net
    net = T_net(decoder=True)
    net.load_state_dict(_np_params(P.tnet_param_shapes(), 11, "T"))
    return net


@pytest.mark.parametrize("tag", ["a", "b"])
def test_tnet_vs_reference_fixture(tnet, gold, tag):
    fx = gold("tnet.npz")
    B, HW, seed, _ = (int(v) for v in fx[tag + "_cfg"])
    x = seeded_tensor(seed, (B, 3, HW, HW), lo=0.0, hi=1.0).cuda()
    r = seeded_tensor(seed + 50, (B, 3, HW, HW)).cuda()
    tnet.zero_grad()
    y = tnet.forward(x, save=True)
    assert relerr(y, torch.from_numpy(fx[tag + "_y"])) < 1e-4
    assert relerr(tnet.last_res, torch.from_numpy(fx[tag + "_res"])) < 1e-4
    tnet.backward(r / r.numel())                      # loss = mean(y*r)
    torch.cuda.synchronize()
    gn = fx[tag + "_gradnorm"]
    for (name, _), ref in zip(P.tnet_param_shapes(), gn):
        g = tnet.store.g[name]
        if ref < 0:
            assert float(g.abs().max()) == 0.0, name          # dead tensors: never touched
        else:
            got = float(g.double().norm())
            assert abs(got - ref) <= 2e-3 * ref + 1e-12, (name, got, ref)
    for key in fx.files:
        if key.startswith(tag + "_gs_"):
            name = key[len(tag) + 4:]
            ref = fx[key]
            got = _strided(tnet.store.g[name])
            assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-12, name


@pytest.mark.parametrize("bi", range(7))
def test_transformer_block_vs_reference_fixture(gold, bi):
    from rcot_amd.net_restormer import ParamStore, TransformerBlockOp
    from rcot_amd.ops import HipBackend
    fx = gold("blocks.npz")
    C, heads, HW, ps, xs, gs = (int(v) for v in fx[f"blk{bi}_cfg"])
    be = HipBackend()
    shapes = P.block_param_shapes("blk", C, heads)
    st = ParamStore(be, shapes, [n for n, _ in shapes], [])
    st.load(_np_params(shapes, ps, "T"))
    blk = TransformerBlockOp(be, st, "blk", C, heads)
    blk.repack()
    x = seeded_tensor(xs, (2, C, HW, HW)).cuda()
    gy = seeded_tensor(gs, (2, C, HW, HW)).cuda()
    y, ctx = blk.forward(x, True)
    dx = blk.backward(ctx, gy)
    torch.cuda.synchronize()
    assert relerr(y, torch.from_numpy(fx[f"blk{bi}_y"])) < 2e-5
    assert relerr(dx, torch.from_numpy(fx[f"blk{bi}_dx"])) < 1e-4
    for name, _ in shapes:
        k = name[len("blk."):]
        ref_n = float(fx[f"blk{bi}_gn_{k}"])
        assert abs(float(st.g[name].double().norm()) - ref_n) <= 5e-4 * ref_n + 1e-9, name
        ref_s = fx[f"blk{bi}_gs_{k}"]
        assert np.abs(_strided(st.g[name], 256) - ref_s).max() <= 5e-4 * np.abs(ref_s).max() + 1e-9, name


@pytest.mark.parametrize("prec,tol_y,tol_g", [("fp32", 2e-5, 5e-4), ("bf16x6", 2e-5, 5e-4), ("bf16x3", 1e-4, 3e-3)])
@pytest.mark.parametrize("bi", range(6))
def test_transformer_block_b8_small_planes_vs_reference_fixture(gold, bi, prec, tol_y, tol_g):
    """One TransformerBlock forward + backward at the TRAINING batch (B = 8) on the 16x16 (8 and 4 heads), 32x32, 64x64 and 128x128 (96 and 48
    channels) planes against
    the REFERENCE's block (tests/golden/blocks_b8.npz, oracle/pin_against_reference.py --only blocks8): the batch decides the dispatch on
    these planes — the B = 2 fixtures never reach the eight-wavefront k-group GEMM, the merged dV / dQ / dK launch or the split-K forms
    B = 8 selects.  Output and input gradient through 8192 strided samples and their norms, every parameter gradient through its
    norm and 256 samples; and the kernels named above really ran (the library's own per-launch profile)."""
    import ctypes
    from rcot_amd import lib
    from rcot_amd.net_restormer import ParamStore, TransformerBlockOp
    from rcot_amd.ops import HipBackend
    fx = gold("blocks_b8.npz")
    C, heads, HW, ps, xs, gs = (int(v) for v in fx[f"b8blk{bi}_cfg"])
    be = HipBackend()
    be.prec = {"fp32": lib.PREC_FP32, "bf16x6": lib.PREC_BF16X6, "bf16x3": lib.PREC_BF16X3}[prec]
    be.x6_packs = prec == "bf16x6"
    shapes = P.block_param_shapes("blk", C, heads)
    st = ParamStore(be, shapes, [n for n, _ in shapes], [])
    st.load(_np_params(shapes, ps, "T"))
    blk = TransformerBlockOp(be, st, "blk", C, heads)
    blk.repack()
    x = seeded_tensor(xs, (8, C, HW, HW)).cuda()
    gy = seeded_tensor(gs, (8, C, HW, HW)).cuda()
    buf = ctypes.create_string_buffer(1 << 16)
    be.L.rcot_profile_begin()
    y, ctx = blk.forward(x, True)
    dx = blk.backward(ctx, gy)
    be.side_join()
    torch.cuda.synchronize()
    assert be.L.rcot_profile_end(buf, 1 << 16) > 10
    ran = buf.value.decode(errors="replace")
    if prec == "bf16x3":
        if HW <= 64:
            assert "x3p_nt_pair_kernel" in ran, ran[:400]       # the paired data + weight gradient launch (round 4), here under a reference fixture
    else:
        assert "gemm_xx_multi_kernel" in ran, ran[:400]
        if prec == "fp32" and HW <= 32:
            assert "gemm_xx_kg_kernel" in ran, ran[:400]
    for got, key, tol in ((y, "y", tol_y), (dx, "dx", 5 * tol_y)):
        ref_s, (ref_n, ref_max) = fx[f"b8blk{bi}_{key}_s"], fx[f"b8blk{bi}_{key}_n"]
        assert np.abs(_strided(got, 8192) - ref_s).max() <= tol * ref_max, key
        assert abs(float(got.double().norm()) - ref_n) <= tol * ref_n, key
    for name, _ in shapes:
        k = name[len("blk."):]
        ref_n = float(fx[f"b8blk{bi}_gn_{k}"])
        assert abs(float(st.g[name].double().norm()) - ref_n) <= tol_g * ref_n + 1e-9, name
        ref_s = fx[f"b8blk{bi}_gs_{k}"]
        assert np.abs(_strided(st.g[name], 256) - ref_s).max() <= tol_g * np.abs(ref_s).max() + 1e-9, name


def test_tnet_vs_reference_128(tnet, gold):
    """north_star forward bar: identical 128x128 patch batch, <= 1e-3 relative fp32 (asserted: 1e-4) against the REFERENCE's
    output (gpu_fixtures.npz, made by oracle/pin_against_reference.py --only gpufx from the imported reference)."""
    fx = gold("gpu_fixtures.npz")
    B, HW, seed, pseed = (int(v) for v in fx["fwd128_cfg"])
    assert pseed == 11
    x = seeded_tensor(seed, (B, 3, HW, HW), lo=0.0, hi=1.0)
    e = relerr(tnet(x.cuda()), torch.from_numpy(fx["fwd128_y"]))
    assert e < 1e-4, e


def test_tnet_full_size_properties(tnet):
    """B=8, 128x128 (BASELINE.json config 2 shape): samples are independent (batch of 8 == two batches of 4),
    the saving and non-saving forward agree bit-for-bit, and backward is linear in the output gradient."""
    x = seeded_tensor(901, (8, 3, 128, 128), lo=0.0, hi=1.0).cuda()
    y8 = tnet.forward(x, save=False)
    ya, yb = tnet.forward(x[:4].contiguous()), tnet.forward(x[4:].contiguous())
    assert relerr(torch.cat([ya, yb]), y8) < 1e-5
    ys = tnet.forward(x, save=True)
    assert torch.equal(ys, y8)
    r = seeded_tensor(902, (8, 3, 128, 128)).cuda()
    tnet.zero_grad()
    tnet.backward(r)
    g1 = tnet.store.grad.clone()
    tnet.forward(x, save=True)
    tnet.zero_grad()
    tnet.backward(-2.0 * r)
    torch.cuda.synchronize()
    assert relerr(tnet.store.grad, -2.0 * g1) < 1e-4
    assert bool(torch.isfinite(g1).all()) and float(g1.abs().max()) > 0


@pytest.mark.parametrize("ps", [64, 128])
def test_fnet_vs_reference_fixture(gold, ps):
    from rcot_amd.net_restormer import F_net
    fx = gold("fnet.npz")
    t = f"p{ps}"
    _, seed, pseed = (int(v) for v in fx[t + "_cfg"])
    net = F_net(patch_size=ps)
    net.load_state_dict(_np_params(P.fnet_param_shapes(ps), pseed, "F"))
    x = seeded_tensor(seed, (2, 3, ps, ps), lo=0.0, hi=1.0).cuda()
    names = [n for n, _ in P.fnet_param_shapes(ps)]
    # forward + input gradient (grad_outputs = ones, trainer.py:291-298)
    out = net.forward(x, save=True)
    dfdx = net.backward(torch.ones(2, device="cuda"), wgrad=False, need_dx=True)
    assert relerr(out, torch.from_numpy(fx[t + "_out"])) < 1e-4
    assert relerr(dfdx, torch.from_numpy(fx[t + "_dfdx"])) < 1e-4
    # critic-loss gradients: loss = -mean F(x)
    net.zero_grad()
    net.forward(x, save=True)
    net.backward(torch.full((2,), -0.5, device="cuda"), wgrad=True)
    for n, ref in zip(names, fx[t + "_cr_gradnorm"]):
        assert abs(float(net.store.g[n].double().norm()) - ref) <= 1e-3 * ref + 1e-12, n
    for k in fx.files:
        if k.startswith(t + "_cr_gs_"):
            ref = fx[k]
            assert np.abs(_strided(net.store.g[k[len(t) + 7:]]) - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-12, k
    # gradient penalty (double backward in the reference, explicit sweeps here)
    net.zero_grad()
    gp = net.be.empty(1)
    net.gradient_penalty_backward(x, 0.5, gp)
    torch.cuda.synchronize()
    assert abs(float(gp) - float(fx[t + "_gp"])) < 1e-4 * float(fx[t + "_gp"])
    for n, ref in zip(names, fx[t + "_gp_gradnorm"]):
        got = float(net.store.g[n].double().norm())
        if ref <= 0:
            assert got == 0.0, n                       # exact-zero bias grads / fc2.bias None
        else:
            assert abs(got - ref) <= 2e-3 * ref, (n, got, ref)
    for k in fx.files:
        if k.startswith(t + "_gp_gs_"):
            ref = fx[k]
            assert np.abs(_strided(net.store.g[k[len(t) + 7:]]) - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-12, k


@pytest.mark.parametrize("tag,opt_name", [("unpaired", "RMSprop"), ("paired", "RMSprop"), ("adam", "Adam")])
def test_minimax_iteration_vs_verbatim_reference(gold, tag, opt_name):
    """One iteration of the reference's own trainer.train() (fixture train_iter.npz) vs the HIP step."""
    from rcot_amd.net_restormer import F_net, T_net
    from rcot_amd.trainer import FlatOptimizer, MinimaxStep
    fx = gold("train_iter.npz")
    cfg = [int(v) for v in fx[tag + "_cfg"]]
    B, ps, paired, sT, sF, s1, s2, s3 = cfg[:8]
    de = cfg[8:]
    lr = 1e-4
    pT, pF = _np_params(P.tnet_param_shapes(), sT, "T"), _np_params(P.fnet_param_shapes(ps), sF, "F")
    Tn, Fn = T_net(decoder=True), F_net(patch_size=ps)
    Tn.load_state_dict(pT)
    Fn.load_state_dict(pF)
    clean = seeded_tensor(s1, (B, 3, ps, ps), lo=0.0, hi=1.0)
    deg = (clean + seeded_tensor(s2, (B, 3, ps, ps), scale=50 / 255)).clamp(0, 1)
    alpha = seeded_tensor(s3, (B, 1, 1, 1), lo=0.0, hi=1.0).view(B)
    st = MinimaxStep(Tn, Fn, FlatOptimizer(Tn, opt_name, lr / 2), FlatOptimizer(Fn, opt_name, lr), 1.0, 10000.0)
    st.set_de_ids(de)
    st.iteration(deg.cuda(), clean.cuda(), torch.tensor(de, dtype=torch.int32).cuda(), alpha.cuda(), bool(paired))
    torch.cuda.synchronize()
    s = st.scalars()
    ref = fx[tag + "_losses"]
    for got, want in zip((s["Loss_F"], s["Loss_T"], s["Loss_mse"], s["gp"]), ref):
        assert abs(got - want) <= 1e-3 * max(abs(want), 1e-3), (got, want)
    # parameter movement: per-tensor L2 norm of the update matches the reference's
    for net, p0, key in ((Tn, pT, "_Tdelta"), (Fn, pF, "_Fdelta")):
        want = fx[tag + key]
        got = np.array([float((net.store.p[n].cpu().double() - p0[n].double()).norm()) for n, _ in net.store.shapes])
        big = want > 0
        assert np.all(got[~big] == 0.0)
        assert np.abs(got[big] - want[big]).max() <= 0.05 * want[big].max()
        assert np.abs(got[big] / want[big] - 1).mean() < 0.02
    # (the gradients of the three half-steps themselves, before any optimizer touches them: tests/test_iteration_grads_gpu.py)


def test_rccl_reducer_path_single_gpu(tmp_path):
    """The bucketed side-stream RCCL all-reduce path (parallel.GradReducer) exercised on ONE GPU with a world of
    size 1 (RCOT_FORCE_REDUCER=1): same losses as the plain run, no hang, buckets actually launched."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    outs = []
    for force in ("0", "1"):
        env = dict(os.environ, RCOT_FORCE_REDUCER=force, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0",
                   WORLD_SIZE="1", LOCAL_RANK="0")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2",
                            "--patch", "64", "--no-cpu-baseline", "--no-roofline"], capture_output=True, text=True,
                           timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1]))
    a, b = outs[0]["losses_last_step"], outs[1]["losses_last_step"]
    for k in a:
        assert abs(a[k] - b[k]) <= 1e-4 * max(1.0, abs(a[k])), (k, a[k], b[k])


def test_trainer_cli_checkpoint_resume(tmp_path):
    """The reference-compatible CLI end to end: two synthetic epochs, checkpoint with the reference's file name and
    keys (trainer.py:362-371), then --resume picks up at epoch+1 (trainer.py:100-108) and keeps training."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, PYTHONPATH=ROOT)
    base = [sys.executable, "-m", "rcot_amd.trainer", "--synthetic", "--iters", "2", "--batchSize", "2", "--patch_size", "64",
            "--de_type", "denoise_50", "derain", "--pairnum", "2", "--seed", "3", "--type", "CliTest", "--sigma", "1"]
    r = subprocess.run(base + ["--nEpochs", "1"], capture_output=True, text=True, timeout=600, cwd=tmp_path, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Loss_F" in r.stdout and "Checkpoint saved" in r.stdout
    ck = os.path.join(tmp_path, "checkpoint", "model_CliTest__1_1.0.pth")
    assert os.path.isfile(ck)
    from rcot_amd.compat import load_checkpoint
    sd = load_checkpoint(ck)                      # whole network objects under the reference's class paths (compat.py)
    assert sd["epoch"] == 1 and type(sd["Tnet"]).__module__ == "Net_Restormer" and type(sd["Tnet"]).__name__ == "T_net"
    assert [k for k in sd["Tnet"].state_dict()] == [n for n, _ in P.tnet_param_shapes()]
    assert [tuple(v.shape) for v in sd["Fnet"].state_dict().values()] == [s for _, s in P.fnet_param_shapes(64)]
    assert sd["T_optimizer"]["kind"] == "RMSprop" and set(sd["T_optimizer"]["state"]["sq"]) == set(sd["Tnet"].state_dict())
    y = sd["Tnet"](torch.rand(1, 3, 64, 64).cuda())          # tester.py:54 usage: the unpickled object is callable
    assert tuple(y.shape) == (1, 3, 64, 64) and bool(torch.isfinite(y).all())
    assert os.path.isfile(os.path.join(tmp_path, "checksample", "CliTest", "validation_results.txt"))
    assert os.path.isfile(os.path.join(tmp_path, "checksample", "CliTest", "output.png"))
    r2 = subprocess.run(base + ["--nEpochs", "2", "--resume", ck], capture_output=True, text=True, timeout=600, cwd=tmp_path, env=env)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert "Epoch=2" in r2.stdout and "Epoch=1," not in r2.stdout


def test_side_stream_overlap_matches_single_stream():
    """The weight-gradient side stream (HipBackend.side_run / side_join, RCOT_OVERLAP) must only change WHEN the leaf
    kernels run: one forward + backward of the transport map on two backends (side stream on / off), same parameters and
    input -> the same flat gradient up to the float-atomic reordering of the depthwise weight gradients."""
    from rcot_amd.net_restormer import T_net
    from rcot_amd.ops import HipBackend
    sd = _np_params(P.tnet_param_shapes(), 11, "T")
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 64, 64, generator=g).cuda()
    dout = torch.randn(2, 3, 64, 64, generator=g).cuda()
    grads = []
    for overlap in (True, False):
        be = HipBackend()
        be.overlap = overlap
        if overlap:
            assert be._side is not None, "the side stream is expected to be on by default"
        net = T_net(decoder=True, backend=be)
        net.load_state_dict(sd)
        net.zero_grad()
        net.forward(x, save=True)
        net.backward(dout)
        torch.cuda.synchronize()
        grads.append(net.store.grad.detach().double().cpu().clone())
    a, b = grads
    assert torch.isfinite(a).all() and a.abs().max() > 0
    assert (a - b).abs().max() <= 2e-5 * b.abs().max(), float((a - b).abs().max() / b.abs().max())


def test_arithmetic_switch_refreshes_the_fragment_packs():
    """A repack writes only the fragment packs the CURRENT arithmetic reads (HipBackend.pack_table(items, prec)); switching
    ``backend.prec`` on a live network must therefore make the next forward repack: fp32 -> bf16x3 -> bf16x6 -> fp32 on one
    network, parameters changed in between (so that stale packs would show), each forward against a freshly built network in
    that arithmetic."""
    from rcot_amd import lib
    from rcot_amd.net_restormer import T_net
    from rcot_amd.ops import HipBackend
    sd = _np_params(P.tnet_param_shapes(), 11, "T")
    sd2 = {k: v * 1.01 for k, v in sd.items()}
    g = torch.Generator().manual_seed(6)
    x = torch.rand(2, 3, 64, 64, generator=g).cuda()
    be = HipBackend()
    be.x6_packs = True
    live = T_net(decoder=True, backend=be)
    live.load_state_dict(sd)
    for i, prec in enumerate((lib.PREC_BF16X3, lib.PREC_BF16X6, lib.PREC_FP32, lib.PREC_BF16X3)):
        params = sd2 if i % 2 == 0 else sd
        be.prec = lib.PREC_FP32
        live.load_state_dict(params)                  # repacks under fp32: the split packs keep their OLD contents
        be.prec = prec
        y = live.forward(x).clone()
        be2 = HipBackend()
        be2.x6_packs = True
        be2.prec = prec
        fresh = T_net(decoder=True, backend=be2)
        fresh.load_state_dict(params)
        y2 = fresh.forward(x)
        torch.cuda.synchronize()
        assert torch.equal(y, y2), (prec, relerr(y, y2))
