"""CPU tier: BASELINE configs[0] — the reference's older MPRNet transport map (Net.py:179-216) and the minimax loop on stock PyTorch
ops (rcot_amd/mprnet.py), against fixtures made from the imported reference (oracle/pin_against_reference.py --only mprnet):
state_dict contract, forward + gradient norms at 2 x 64 x 64, and the first iterations of the verbatim trainer.train() trajectory of
configs[0] (B=4, 128x128, de_type single, unpaired, RMSprop)."""
import numpy as np
import torch

from conftest import relerr, seeded_tensor
from rcot_amd import mprnet as MP
from rcot_amd import params as P


def _params():
    shapes = MP.mprnet_param_shapes()
    prm = {k: torch.from_numpy(v) for k, v in P.seeded_params([(n, s) for n, s in shapes if not n.endswith("body.1.weight")], 71, "T").items()}
    for n, _ in shapes:
        if n.endswith("body.1.weight"):
            prm[n] = torch.full((1,), 0.2)
    return prm


def test_mprnet_contract_and_forward_backward_vs_reference(gold):
    fx = gold("mprnet.npz")
    shapes = MP.mprnet_param_shapes()
    assert len(shapes) == 127 and sum(int(np.prod(s)) for _, s in shapes) == 6842710      # Net.T_net().state_dict() (shared PReLU listed 22x)
    net = MP.MPRNetT(seed=0)
    assert list(net.state_dict()) == [n for n, _ in shapes] and len(net.parameters()) == 127 - 21
    net.load_state_dict(_params())
    B, HW, _ps, sx, sr = (int(v) for v in fx["cfg"])
    x, r = seeded_tensor(sx, (B, 3, HW, HW), lo=0.0, hi=1.0), seeded_tensor(sr, (B, 3, HW, HW))
    y = net(x)
    assert relerr(y, torch.from_numpy(fx["y"])) < 1e-5
    (y * r).mean().backward()
    # reference named_parameters() order: first occurrence of every distinct tensor, csff_* of the residual encoder unused (None -> -1)
    seen, got = set(), []
    for n, _ in shapes:
        t = net.p[n]
        if id(t) in seen:
            continue
        seen.add(id(t))
        got.append(-1.0 if t.grad is None else float(t.grad.double().norm()))
    want = fx["gn"]
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert (w < 0 and g < 0) or abs(g - w) <= 1e-4 * w + 1e-12, (g, w)


def test_mprnet_minimax_trajectory_vs_verbatim_reference(gold):
    from rcot_amd.synth import make_batch
    fx = gold("mprnet.npz")
    cfg = [int(v) for v in fx["traj_cfg"]]
    B, ps, _steps, _sT, sF, sb, sa = cfg[:7]
    de = cfg[7:]
    Tn, Fn = MP.MPRNetT(seed=0), MP.FNetTorch(ps, seed=0)
    Tn.load_state_dict(_params())
    Fn.load_state_dict({k: torch.from_numpy(v) for k, v in P.seeded_params(P.fnet_param_shapes(ps), sF, "F").items()})
    lr = 1e-4
    To, Fo = torch.optim.RMSprop(Tn.parameters(), lr=lr / 2), torch.optim.RMSprop(Fn.parameters(), lr=lr)
    for i in range(2):                                               # two of the ten fixture iterations keep the CPU tier short
        _, x, y = make_batch(sb + i, B, ps, de)
        alpha = seeded_tensor(sa + i, (B, 1, 1, 1), lo=0.0, hi=1.0).view(B)
        s = MP.torch_minimax_iteration(Tn, Fn, To, Fo, x, y, de, alpha, 1.0, 10000.0, False)
        # iteration 0 pins the arithmetic; later iterations sit behind RMSprop's sign-like first steps, where a different host
        # thread count (MKL-DNN reduction order) already moves the losses by a few 1e-3
        tol = 2e-3 if i == 0 else 1e-2
        for got, want in zip((s["Loss_F"], s["Loss_T"], s["Loss_mse"]), fx["traj"][i]):
            assert abs(got - want) <= tol * max(abs(want), 1e-3), (i, got, want)


def test_mprnet_hip_layout_covers_the_distinct_tensors():
    """host logic of the HIP form (rcot_amd/mprnet_hip.py; the network itself is GPU tier): its flat-buffer order is the 98 live tensors
    in the order its backward finishes them, then the 8 the reference's forward never reaches — together the 106 distinct tensors of
    Net.T_net().state_dict() (the shared PReLU slope once)"""
    from rcot_amd import mprnet_hip as MH
    shapes = dict(MP.mprnet_param_shapes())
    live, dead = MH.mprnet_live_order(), MH.mprnet_dead()
    assert len(live) == 98 and len(dead) == 8 and len(set(live) | set(dead)) == 106
    assert all(n in shapes for n in live + dead)
    slopes = [n for n in shapes if n.endswith("body.1.weight")]
    assert len(slopes) == 22 and [n for n in live if n.endswith("body.1.weight")] == [slopes[0]] and live[-1] == slopes[0]
    assert set(shapes) - set(slopes) == (set(live) | set(dead)) - {slopes[0]}
    # the residual branch is used once per forward: final first; the decoder and SAM's image convolution serve both passes
    assert live.index("res_shallow_feat1.0.weight") < live.index("sam12.conv2.weight") < live.index("stage1_decoder.up21.up.1.weight") \
        < live.index("stage1_encoder.down12.down.1.weight") < live.index("shallow_feat1.0.weight")
    lay = P.make_layout([(n, shapes[n]) for n in live + dead], live, dead)
    assert lay.n_live <= lay.offset[dead[0]] and lay.n_total >= sum(int(np.prod(shapes[n])) for n in live + dead)


def test_mprnet_hip_schedule_matches_stock_ops_in_fp64():
    """the HOST schedule of the HIP form (explicit forward / backward of rcot_amd/mprnet_hip.py: the in-place residual gradients, the
    SkipUpSample with its 1x1 in front, the data gradients as forward products with flipped weights, gradient accumulation over the two
    passes, the order in which gradients become final) with every entry point doubled in fp64 (tests/host_double.py) against autograd
    through the stock-ops form — itself pinned to the reference above"""
    from host_double import TorchDouble
    from rcot_amd.mprnet_hip import MPRNetHip
    D = torch.float64
    prm = {k: v.to(D) for k, v in _params().items()}
    net = MPRNetHip(backend=TorchDouble(D), seed=0)
    net.load_state_dict(prm)
    assert len(net._flip_names) == 44 and [c for c, _n, _t in net._flip_groups] == [80, 128, 176]      # the 3x3 weights of the 22 CABs
    po = {k: v.clone().requires_grad_(True) for k, v in prm.items()}
    slope = po["shallow_feat1.1.body.1.weight"]
    for k in po:
        if k.endswith("body.1.weight"):
            po[k] = slope                                               # ONE shared parameter (Net.py:185)
    x, r = seeded_tensor(91, (2, 3, 16, 24), lo=0.0, hi=1.0, dtype=D), seeded_tensor(92, (2, 3, 16, 24), dtype=D)
    yo = MP.mprnet_forward(po, x)
    (yo * r).sum().backward()
    net.zero_grad()
    y = net.forward(x, save=True)
    assert relerr(y, yo) < 1e-12 and relerr(net(x), yo) < 1e-12          # (the inference form works in place)
    ready = []
    net.grad_ready_hook = ready.append
    net.backward(r.clone())
    net.grad_ready_hook = None
    assert ready == sorted(ready) and len(ready) == 3 and ready[-1] == net.store.layout.n_live
    seen = set()
    for n, _ in MP.mprnet_param_shapes():
        t = po[n]
        if id(t) in seen:
            continue
        seen.add(id(t))
        key = net.slope_name if n.endswith("body.1.weight") else n
        if t.grad is None:
            assert float(net.store.g[key].abs().max()) == 0.0, n
        else:
            assert relerr(net.store.g[key], t.grad) < 1e-10, (n, relerr(net.store.g[key], t.grad))
    # a parameter change is followed by the flipped copies (FlatOptimizer.step calls repack())
    net.store.p["shallow_feat1.1.body.0.weight"].mul_(2.0)
    net.repack()
    w = net.store.p["shallow_feat1.1.body.0.weight"]
    assert torch.equal(net.flipped("shallow_feat1.1.body.0.weight"), w.flip(2, 3).transpose(0, 1).contiguous())
    w3 = net.store.p["stage1_decoder.decoder_level3.1.body.2.weight"]
    assert torch.equal(net.flipped("stage1_decoder.decoder_level3.1.body.2.weight"), w3.flip(2, 3).transpose(0, 1).contiguous())
