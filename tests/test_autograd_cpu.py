"""CPU tier: the torch.autograd front end (rcot_amd/autograd.py).  The reference's loop body (trainer.py:262-346, restated in
tests/ref_loop.py with the calls it makes) runs on ``TNetModule`` / ``FNetModule`` over the explicit schedules (kernel layer = fp64
test double).  It must land where the oracle's
iteration lands: same losses, same parameter updates, dead tensors untouched, ``fc2.bias`` without a gradient in the penalty step."""
import pytest
import torch

from conftest import seeded_tensor
from host_double import TorchDouble
from oracle import rcot_oracle as O
from rcot_amd import params as P
from rcot_amd.autograd import as_modules
from rcot_amd.net_restormer import F_net, T_net
from rcot_amd.trainer import freeze
from ref_loop import reference_style_iteration

D = torch.float64


def _params(shapes, seed, kind):
    return {k: torch.from_numpy(v).to(D) for k, v in P.seeded_params(shapes, seed, kind).items()}


@pytest.mark.parametrize("opt_name,paired,de,n_it", [("RMSprop", False, [2, 3], 2), ("Adam", True, [4, 1], 1)])
def test_reference_loop_body_on_autograd_modules_matches_oracle(opt_name, paired, de, n_it):
    be = TorchDouble(D)
    ps, B, lr = 32, 2, 1e-4
    Tn, Fn = T_net(decoder=True, backend=be, seed=0), F_net(patch_size=ps, backend=be, seed=1)
    pT, pF = _params(P.tnet_param_shapes(), 31, "T"), _params(P.fnet_param_shapes(ps), 32, "F")
    Tn.load_state_dict(pT)
    Fn.load_state_dict(pF)
    Tnet, Fnet = as_modules(Tn, Fn)
    assert list(Tnet.state_dict()) == [n for n, _ in P.tnet_param_shapes()] and len(list(Fnet.parameters())) == len(P.fnet_param_shapes(ps))
    mk = torch.optim.RMSprop if opt_name == "RMSprop" else torch.optim.Adam              # trainer.py:121-126
    T_optimizer, F_optimizer = mk(Tnet.parameters(), lr=lr / 2), mk(Fnet.parameters(), lr=lr)
    clean = seeded_tensor(801, (B, 3, ps, ps), lo=0.0, hi=1.0, dtype=D)
    deg = (clean + seeded_tensor(802, (B, 3, ps, ps), scale=50 / 255, dtype=D)).clamp(0, 1)
    alpha = seeded_tensor(803, (B,), lo=0.0, hi=1.0, dtype=D)
    seen = {}

    def probe(Fm):   # penalty step: biases exactly zero, fc2.bias without a gradient (as autograd gives upstream)
        g = {n: p.grad for n, p in zip(Fm._names, Fm.flat_params)}
        seen["fc2b"] = g["fc2.bias"]
        seen["bias_max"] = max(float(v.abs().max()) for n, v in g.items() if n.endswith("bias") and v is not None)
    for _ in range(n_it):
        logs = reference_style_iteration(Tnet, Fnet, T_optimizer, F_optimizer, deg, clean, de, alpha, 1.0, 10000.0, paired, probe)
    assert seen["fc2b"] is None and seen["bias_max"] == 0.0
    qT, qF = {k: v.clone() for k, v in pT.items()}, {k: v.clone() for k, v in pF.items()}
    omk = O.RMSprop if opt_name == "RMSprop" else O.Adam
    oT, oF = omk(qT, lr / 2), omk(qF, lr)
    for _ in range(n_it):
        want = O.minimax_iteration(qT, qF, oT, oF, deg, clean, de, alpha.view(B, 1, 1, 1), 1.0, 10000.0, paired)
    for k in ("Loss_F", "Loss_T", "Loss_mse", "gp"):
        assert abs(logs[k] - want[k]) <= 1e-8 * max(1.0, abs(want[k])), (k, logs[k], want[k])

    def upd_err(net, q, p0):
        num = den = 0.0
        for k, v in net.state_dict().items():
            num += float(((v - p0[k]) - (q[k].detach() - p0[k])).pow(2).sum())
            den += float((q[k].detach() - p0[k]).pow(2).sum())
        return (num / den) ** 0.5
    assert upd_err(Fn, qF, pF) < 1e-5 and upd_err(Tn, qT, pT) < 1e-5
    for k, _ in P.tnet_param_shapes():
        if P.tnet_is_dead(k):
            assert torch.equal(Tn.store.p[k], pT[k])                                  # never used upstream: grad None, not stepped


def test_autograd_modules_accumulate_and_freeze_like_modules():
    """two backward passes accumulate into .grad; a frozen critic yields input gradients only; an inference call keeps nothing"""
    be = TorchDouble(D)
    ps, B = 32, 2
    Fn = F_net(patch_size=ps, backend=be, seed=1)
    Fn.load_state_dict(_params(P.fnet_param_shapes(ps), 32, "F"))
    _, Fnet = as_modules(T_net(decoder=True, backend=be, seed=0), Fn)
    x = seeded_tensor(5, (B, 3, ps, ps), lo=0.0, hi=1.0, dtype=D)
    Fnet(x).sum().backward()
    g1 = [p.grad.clone() for p in Fnet.parameters()]
    Fnet(x).sum().backward()
    for a, p in zip(g1, Fnet.parameters()):
        assert torch.allclose(p.grad, 2 * a, rtol=1e-12, atol=0)
    freeze(Fnet)
    Fnet.zero_grad()
    xr = x.clone().requires_grad_(True)
    Fnet(xr).sum().backward()
    assert xr.grad is not None and all(p.grad is None or float(p.grad.abs().max()) == 0.0 for p in Fnet.parameters())
    with torch.no_grad():
        Fnet(x)
    assert Fn._ctx is None
