"""CPU tier: the hand-written forward/backward SCHEDULES (rcot_amd/net_restormer.py) and the minimax
step (rcot_amd/trainer.py) checked against the oracle, with the kernel layer replaced by the torch
test double (tests/host_double.py).  fp64 so that only logic errors, not rounding, can show."""
import numpy as np
import pytest
import torch

from conftest import relerr, seeded_tensor
from host_double import TorchDouble
from oracle import rcot_oracle as O
from rcot_amd import params as P
from rcot_amd.net_restormer import F_net, T_net
from rcot_amd.trainer import FlatOptimizer, MinimaxStep

D = torch.float64


def _params(shapes, seed, kind):
    return {k: torch.from_numpy(v).to(D) for k, v in P.seeded_params(shapes, seed, kind).items()}


@pytest.fixture(scope="module")
def tnet():
    net = T_net(decoder=True, backend=TorchDouble(D), seed=0)
    prm = _params(P.tnet_param_shapes(), 11, "T")
    net.load_state_dict(prm)
    return net, prm


def test_tnet_state_dict_contract(tnet):
    net, _ = tnet
    sd = net.state_dict()
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == P.tnet_param_shapes()
    lay = net.store.layout
    assert lay.n_live % 64 == 0 and all(lay.offset[n] >= lay.n_live for n in lay.order if P.tnet_is_dead(n))


def test_tnet_forward_backward_matches_oracle(tnet):
    net, prm = tnet
    x = seeded_tensor(502, (2, 3, 32, 32), lo=0.0, hi=1.0, dtype=D)
    r = seeded_tensor(552, (2, 3, 32, 32), dtype=D)
    po = {k: v.clone().requires_grad_(True) for k, v in prm.items()}
    yo, reso = O.tnet_forward(po, x, True, return_res=True)
    (yo * r).sum().backward()
    net.zero_grad()
    y = net.forward(x, save=True)
    assert relerr(y, yo) < 1e-10 and relerr(net.last_res, reso) < 1e-10
    ready = []
    net.grad_ready_hook = ready.append
    net.backward(r.clone())
    net.grad_ready_hook = None
    assert ready == sorted(ready) and ready[-1] == net.store.layout.n_live
    worst = 0.0
    for k, _ in P.tnet_param_shapes():
        if P.tnet_is_dead(k):
            assert po[k].grad is None and float(net.store.g[k].abs().max()) == 0.0
        else:
            worst = max(worst, relerr(net.store.g[k], po[k].grad))
    assert worst < 1e-8, worst
    # inference call path (no saved activations) gives the same output
    assert relerr(net(x), yo) < 1e-10


def test_tnet_with_layernorm_statistics_made_by_the_producing_product(tnet):
    """round 6: where the backend offers it (stats_ok: C <= 96, 128-pixel tiles) the statistics of a LayerNorm's input come from the
    epilogue of the product that stored that input — norm2's from the attention apply, the next block's norm1's from project_out,
    handed on by _stage_fwd — and the first block of a stage still makes its own.  Same outputs and gradients as the schedule in
    which every LayerNorm makes its own statistics."""
    net, prm = tnet
    be = TorchDouble(D)
    be.prod_stats = True
    net2 = T_net(decoder=True, backend=be, seed=0)
    net2.load_state_dict(prm)
    made, asked = [], []
    o1, o2 = be.gemm_kmajor_stats, be.conv1x1_fwd
    be.gemm_kmajor_stats = lambda *a, **k: (made.append(a[3]), o1(*a, **k))[1]
    be.conv1x1_fwd = lambda *a, **k: (asked.append((a[0].shape[0], k.get("stats") is not None, bool(k.get("ln_compute")))), o2(*a, **k))[1]
    x = seeded_tensor(502, (2, 3, 32, 32), lo=0.0, hi=1.0, dtype=D)
    r = seeded_tensor(552, (2, 3, 32, 32), dtype=D)
    outs = []
    for n in (net, net2):
        n.zero_grad()
        outs.append(n.forward(x, save=True))
        n.backward(r.clone())
    assert relerr(outs[1], outs[0]) < 1e-12
    assert made and set(made) == {48, 96}                            # the attention apply of every block on the 48- / 96-channel levels
    withst = [co for co, st, _ in asked if st]
    assert withst and set(withst) == {48, 96}                        # project_out of every block but the last of its stage
    assert any(lc for co, st, lc in asked if co in (144, 288))       # the first block of a stage: qkv makes its own statistics
    assert any((not lc) and (not st) for co, st, lc in asked if co in (144, 288))       # later blocks: qkv takes the handed-on ones
    worst = max(relerr(net2.store.g[k], net.store.g[k]) for k, _ in P.tnet_param_shapes() if not P.tnet_is_dead(k))
    assert worst < 1e-10, worst


def test_tnet_fixture_vs_reference(tnet, gold):
    """fp64 host schedule vs the REFERENCE's own fp32 output (fixture tnet.npz 'b')."""
    net, _ = tnet
    fx = gold("tnet.npz")
    B, HW, seed, _ = fx["b_cfg"]
    x = seeded_tensor(int(seed), (int(B), 3, int(HW), int(HW)), lo=0.0, hi=1.0, dtype=D)
    assert relerr(net(x), torch.from_numpy(fx["b_y"])) < 1e-5


@pytest.mark.parametrize("ps", [64])
def test_fnet_and_gp_match_oracle(ps):
    be = TorchDouble(D)
    net = F_net(patch_size=ps, backend=be, seed=0)
    prm = _params(P.fnet_param_shapes(ps), 21, "F")
    net.load_state_dict(prm)
    x = seeded_tensor(601, (2, 3, ps, ps), lo=0.0, hi=1.0, dtype=D)
    po = {k: v.clone().requires_grad_(True) for k, v in prm.items()}
    xo = x.clone().requires_grad_(True)
    oo = O.fnet_forward(po, xo)
    w = torch.tensor([-0.5, 0.7], dtype=D)
    (oo * w).sum().backward()
    net.zero_grad()
    out = net.forward(x, save=True)
    dx = net.backward(w.clone(), wgrad=True, need_dx=True)
    assert relerr(out, oo) < 1e-10 and relerr(dx, xo.grad) < 1e-9
    for k in po:
        assert relerr(net.store.g[k], po[k].grad) < 1e-9, k
    # gradient penalty via explicit sweeps vs autograd double backward
    po = {k: v.clone().requires_grad_(True) for k, v in prm.items()}
    gpo = O.gradient_penalty(po, x)
    go = O._grads(gpo, po)
    net.zero_grad()
    gp = be.empty(1)
    net.gradient_penalty_backward(x, 1.0 / 2, gp)
    assert abs(float(gp) - float(gpo)) < 1e-9 * abs(float(gpo))
    for k in po:
        if go[k] is None:
            assert k == "fc2.bias" and float(net.store.g[k].abs().max()) == 0.0
        elif float(go[k].abs().max()) == 0.0:
            assert float(net.store.g[k].abs().max()) == 0.0, k
        else:
            assert relerr(net.store.g[k], go[k]) < 1e-8, k


@pytest.mark.parametrize("opt_name,paired,de", [("RMSprop", False, [2, 3]), ("RMSprop", True, [0, 7]), ("Adam", True, [4, 1])])
def test_minimax_iteration_matches_oracle(opt_name, paired, de):
    be = TorchDouble(D)
    ps, B, lr = 32, 2, 1e-4
    Tn, Fn = T_net(decoder=True, backend=be, seed=0), F_net(patch_size=ps, backend=be, seed=1)
    pT, pF = _params(P.tnet_param_shapes(), 31, "T"), _params(P.fnet_param_shapes(ps), 32, "F")
    Tn.load_state_dict(pT)
    Fn.load_state_dict(pF)
    To, Fo = FlatOptimizer(Tn, opt_name, lr / 2), FlatOptimizer(Fn, opt_name, lr)
    clean = seeded_tensor(801, (B, 3, ps, ps), lo=0.0, hi=1.0, dtype=D)
    deg = (clean + seeded_tensor(802, (B, 3, ps, ps), scale=50 / 255, dtype=D)).clamp(0, 1)
    alpha = seeded_tensor(803, (B,), lo=0.0, hi=1.0, dtype=D)
    st = MinimaxStep(Tn, Fn, To, Fo, 1.0, 10000.0)
    st.set_de_ids(de)
    n_it = 2 if opt_name == "Adam" else 1           # two iterations: parameters (and their K-major packs) change in between
    for _ in range(n_it):
        st.iteration(deg, clean, torch.tensor(de, dtype=torch.int32), alpha, paired)
    s = st.scalars()
    qT = {k: v.clone() for k, v in pT.items()}
    qF = {k: v.clone() for k, v in pF.items()}
    mk = O.RMSprop if opt_name == "RMSprop" else O.Adam
    oT, oF = mk(qT, lr / 2), mk(qF, lr)
    for _ in range(n_it):
        logs = O.minimax_iteration(qT, qF, oT, oF, deg, clean, de, alpha.view(B, 1, 1, 1), 1.0, 10000.0, paired)
    for k in ("Loss_F", "Loss_T", "Loss_mse", "gp"):
        assert abs(s[k] - logs[k]) <= 1e-8 * max(1.0, abs(logs[k])), (k, s[k], logs[k])

    def upd_err(net, q, p0):
        num = den = 0.0
        for k, v in net.state_dict().items():
            num += float(((v - p0[k]) - (q[k].detach() - p0[k])).pow(2).sum())
            den += float((q[k].detach() - p0[k]).pow(2).sum())
        return (num / den) ** 0.5
    assert upd_err(Fn, qF, pF) < 1e-5
    assert upd_err(Tn, qT, pT) < 1e-5
    for k, _ in P.tnet_param_shapes():
        if P.tnet_is_dead(k):
            assert torch.equal(Tn.store.p[k], pT[k])


def test_c6_initial_parameter_distributions():
    """SURVEY.md 8a C6 (Net_Restormer.py:501-503 and PyTorch's constructor defaults): F_net Conv2d weights ~ N(0, 0.02), every
    other Conv2d / Linear weight and every bias ~ U(-b, b) with b = 1/sqrt(fan_in of the layer's own weight), LayerNorm affine
    1 / 0, temperature 1; reproducible per seed."""
    from rcot_amd.net_restormer import _reference_init
    for kind, shapes in (("F", P.fnet_param_shapes(128)), ("T", P.tnet_param_shapes())):
        by = dict(shapes)
        a, a2, b = _reference_init(shapes, kind, 5), _reference_init(shapes, kind, 5), _reference_init(shapes, kind, 6)
        assert all(torch.equal(a[n], a2[n]) for n, _ in shapes)
        assert any(not torch.equal(a[n], b[n]) for n, _ in shapes)
        pooled = []
        for n, shp in shapes:
            t = a[n].double()
            assert tuple(t.shape) == tuple(shp)
            if n.endswith("body.weight") or n.endswith("temperature"):
                assert bool((t == 1).all()), n
            elif n.endswith("body.bias"):
                assert bool((t == 0).all()), n
            elif kind == "F" and n.startswith("features.") and n.endswith(".weight"):
                assert abs(float(t.mean())) < 4 * 0.02 / t.numel() ** 0.5 + 1e-12, n
                if t.numel() >= 20000:
                    assert abs(float(t.std()) / 0.02 - 1) < 0.02, (n, float(t.std()))
                pooled.append(t.reshape(-1))
            else:
                wshape = by[n[:-len("bias")] + "weight"] if n.endswith(".bias") else shp
                bound = 1.0 / float(np.prod(wshape[1:])) ** 0.5
                assert float(t.abs().max()) <= bound * (1 + 1e-6), (n, float(t.abs().max()), bound)
                if t.numel() >= 20000:
                    assert abs(float(t.std()) / (bound / 3 ** 0.5) - 1) < 0.02, (n, float(t.std()), bound)
                if t.numel() >= 256:
                    assert float(t.abs().max()) > 0.9 * bound, n                # not a narrower distribution
        if pooled:
            allw = torch.cat(pooled)
            assert abs(float(allw.std()) / 0.02 - 1) < 0.005                    # 13.7 M samples


def test_whole_image_with_odd_latent_plane_matches_oracle(tnet):
    """Whole-image validation sizes the reference accepts but whose 1/8-resolution plane has an odd pixel count (40 x 56 -> 5 x 7):
    the width-padded, masked latent level (T_net._lat_pad, TransformerBlockOp.forward(wmask=...)) gives the unpadded result."""
    net, prm = tnet
    x = seeded_tensor(77, (1, 3, 40, 56), lo=0.0, hi=1.0, dtype=D)
    with torch.no_grad():
        ref = O.tnet_forward(prm, x)
    assert relerr(net(x), ref) < 1e-9


def test_grad_reducer_bucket_order_front_and_tail():
    """parallel.GradReducer: ready() releases complete buckets from the front of the flat buffer (critic-loss / generator
    backward), ready_tail() from its end (the gradient penalty's first-to-last sweep); every bucket leaves exactly once, whatever
    the interleaving, and finish() issues what is left."""
    from rcot_amd import parallel as par
    flat = torch.zeros(1050)
    red = par.GradReducer(flat, 1000, bucket_elems=100)
    red.enabled = True
    sent = []
    red._launch = lambda lo, hi: sent.append((lo, hi))
    red.begin()
    red.ready_tail(950)                 # no complete bucket yet
    assert sent == []
    red.ready_tail(800)
    assert sent == [(900, 1000), (800, 900)]
    red.ready(250)
    assert sent[2:] == [(0, 100), (100, 200)]
    red.ready_tail(0)                   # everything behind the front cursor
    assert sent[4:] == [(i, i + 100) for i in range(700, 100, -100)]
    red.ready(1000)
    assert len(sent) == 10 and sorted(sent) == [(i, i + 100) for i in range(0, 1000, 100)]
    sent.clear()
    red.begin()
    red.ready(1000)
    assert sent == [(i, i + 100) for i in range(0, 1000, 100)]


@pytest.mark.parametrize("ps", [32])
def test_gradient_penalty_tail_hook_ranges_are_final(ps):
    """F_net.gradient_penalty_backward reports grad[n_from:n_live) final through grad_tail_hook: at every call the tail of the
    flat gradient buffer already equals its value at the end of the sweep, and the calls walk towards the front."""
    be = TorchDouble(D)
    net = F_net(patch_size=ps, backend=be, seed=0)
    net.load_state_dict(_params(P.fnet_param_shapes(ps), 21, "F"))
    x = seeded_tensor(601, (2, 3, ps, ps), lo=0.0, hi=1.0, dtype=D)
    seen = []
    net.grad_tail_hook = lambda n: seen.append((n, net.store.grad[n:net.store.layout.n_live].clone()))
    net.zero_grad()
    net.gradient_penalty_backward(x, 0.5, be.empty(1))
    assert len(seen) == len(net.convs) and [n for n, _ in seen] == sorted((n for n, _ in seen), reverse=True)
    for n, snap in seen:
        assert torch.equal(snap, net.store.grad[n:net.store.layout.n_live])
    off = net.store.layout.offset
    assert seen[0][0] == off["features.0.weight"] and seen[-1][0] == off[f"features.{2 * (len(net.convs) - 1)}.weight"]


def test_launch_plan_scalar_patch_and_plan_cache_floor(monkeypatch):
    """a recorded call's by-value scalar (the learning rate of the optimizer launches) is replaced in place, bound argument
    tuples are rebuilt, other entry points and non-matching calls stay untouched; RCOT_PLAN_CACHE=0 is clamped to one plan."""
    from rcot_amd.plan import LaunchPlan, PlannedMinimax

    class _Be:
        _side = None

        def _st(self):
            return 7

    seen = []

    def mk(name):
        def fn(*a):
            seen.append((name, a))
            return 0
        fn.__name__ = name
        fn.argtypes = [lambda v: v] * 4
        return fn

    opt, other = mk("rcot_rmsprop_step"), mk("rcot_fill")
    pl = LaunchPlan(_Be())
    pl.cmds = [(opt, (100, 1, 0.5), False), (other, (100, 1, 0.5), False), (opt, (200, 1, 0.5), False), (lambda: seen.append("host"), None, False)]
    pl.replay()
    assert pl.set_scalar("rcot_rmsprop_step", 2, 0.25, where=lambda a: a[0] == 100) == 1
    seen.clear()
    pl.replay()
    assert seen == [("rcot_rmsprop_step", (100, 1, 0.25, 7)), ("rcot_fill", (100, 1, 0.5, 7)), ("rcot_rmsprop_step", (200, 1, 0.5, 7)), "host"]
    assert pl.set_scalar("rcot_adam_step", 2, 0.1) == 0

    class _Store:
        flat = torch.zeros(1)

    class _Net:
        store = _Store()

    class _Opt:
        kind = "RMSprop"

    class _Step:
        T = F = _Net()
        To = Fo = _Opt()

    monkeypatch.setenv("RCOT_PLAN_CACHE", "0")
    assert PlannedMinimax(_Step()).max_plans == 1


def test_tester_metrics_against_independent_forms():
    """rcot_amd/tester.py restates evaluate.py's PSNR (skimage's, uint8) and its own SSIM (2 x 2 window of cv2.getGaussianKernel(2, 1)
    through cv2.filter2D, cropped [5:-5]) in numpy, cv2 / skimage being absent: checked here against scipy.ndimage's correlation
    (window centre at size // 2, as OpenCV's default anchor) and the closed forms"""
    import scipy.ndimage as ndi
    from rcot_amd import tester as TS
    g = np.random.Generator(np.random.PCG64(5))
    a = g.integers(0, 256, size=(37, 45, 3), dtype=np.uint8)
    b = np.clip(a.astype(np.int64) + g.integers(-20, 21, size=a.shape), 0, 255).astype(np.uint8)
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    assert abs(TS.psnr_uint8(a, b) - 10 * np.log10(255.0 ** 2 / mse)) < 1e-12 and TS.psnr_uint8(a, a) == float("inf")
    win = np.full((2, 2), 0.25)

    def ref_plane(x, y):
        x, y = x.astype(np.float64), y.astype(np.float64)
        f = lambda t: ndi.correlate(t, win, mode="reflect")[5:-5, 5:-5]
        mu1, mu2 = f(x), f(y)
        s1, s2, s12 = f(x * x) - mu1 ** 2, f(y * y) - mu2 ** 2, f(x * y) - mu1 * mu2
        C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
        return (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 ** 2 + mu2 ** 2 + C1) * (s1 + s2 + C2))).mean()
    want = np.mean([ref_plane(a[:, :, c], b[:, :, c]) for c in range(3)])
    assert abs(TS.ssim_image(a, b) - want) < 1e-12
    assert abs(TS.ssim_image(a, a) - 1.0) < 1e-12
