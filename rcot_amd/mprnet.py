"""BASELINE configs[0] — the reference's OLDER transport map (MPRNet-style ``Net.T_net``, Net.py:179-216) and the plumbing
run of the minimax loop on STOCK PyTorch ops (CPU, or the GPU through PyTorch-ROCm), SURVEY.md 8(f4).

This is deliberately NOT the MI355X hot path (north_star names the Restormer backbone; configs[0] is "CPU-only PyTorch, 10 minimax
steps (plumbing)"): no HIP kernels, torch autograd.  It exists so that a user of the reference who trains with ``Net.py`` finds the
same network (127 ``state_dict`` tensors, same names / shapes / order, checkpoints interchange) and the same loop behind the same
CLI (``python -m rcot_amd.trainer --backbone mprnet``).  The network is written as a table of parameters plus pure functions
over a dict of tensors; parity with the imported reference is pinned by oracle/pin_against_reference.py --only mprnet
(tests/golden/mprnet.npz) and checked on the CPU tier (tests/test_mprnet_cpu.py).

Reference sites: CAB / CALayer Net.py:36-73, SAM :19-32, Encoder :75-116, Decoder :118-144, bilinear Down/Up/SkipUp :146-176,
T_net.forward :196-216 (two passes through the SAME decoder and SAM, residual embedding added with weight 0.8), the critic
Net_Restormer.F_net (:436-522; ``Net.F_net`` only accepts 256x256, Net.py:275) and the loop trainer.py:247-346.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

N_FEAT, SCALE, REDUCTION = 80, 48, 4                         # Net.py:180 defaults
_FNET_CONVS = ((5, 1, 2), (4, 2, 1), (3, 1, 1), (4, 2, 1), (3, 1, 1), (4, 2, 1), (3, 1, 1), (4, 2, 1), (3, 1, 1), (4, 2, 1))


# ----------------------------------------------------------------------------------------------------- parameter table
def _cab(prefix: str, n: int) -> List[Tuple[str, tuple]]:
    r = n // REDUCTION
    return [(f"{prefix}.CA.conv_du.0.weight", (r, n, 1, 1)), (f"{prefix}.CA.conv_du.2.weight", (n, r, 1, 1)),
            (f"{prefix}.body.0.weight", (n, n, 3, 3)), (f"{prefix}.body.1.weight", (1,)), (f"{prefix}.body.2.weight", (n, n, 3, 3))]


def mprnet_param_shapes() -> List[Tuple[str, tuple]]:
    """the 127 tensors of ``Net.T_net().state_dict()`` in its order (every ``body.1.weight`` is the ONE shared PReLU slope)"""
    n1, n2, n3 = N_FEAT, N_FEAT + SCALE, N_FEAT + 2 * SCALE
    out: List[Tuple[str, tuple]] = []
    for pre in ("shallow_feat1", "res_shallow_feat1"):
        out += [(f"{pre}.0.weight", (n1, 3, 3, 3))] + _cab(f"{pre}.1", n1)

    def encoder(pre, csff):
        o = []
        for lvl, n in ((1, n1), (2, n2), (3, n3)):
            for i in range(2):
                o += _cab(f"{pre}.encoder_level{lvl}.{i}", n)
        o += [(f"{pre}.down12.down.1.weight", (n2, n1, 1, 1)), (f"{pre}.down23.down.1.weight", (n3, n2, 1, 1))]
        if csff:
            for side in ("enc", "dec"):
                o += [(f"{pre}.csff_{side}{lvl}.weight", (n, n, 1, 1)) for lvl, n in ((1, n1), (2, n2), (3, n3))]
        return o
    out += encoder("stage1_encoder", False)
    for lvl, n in ((1, n1), (2, n2), (3, n3)):
        for i in range(2):
            out += _cab(f"stage1_decoder.decoder_level{lvl}.{i}", n)
    out += _cab("stage1_decoder.skip_attn1", n1) + _cab("stage1_decoder.skip_attn2", n2)
    out += [("stage1_decoder.up21.up.1.weight", (n1, n2, 1, 1)), ("stage1_decoder.up32.up.1.weight", (n2, n3, 1, 1))]
    out += encoder("stage1_resencoder", True)
    out += [("sam12.conv1.weight", (n1, n1, 1, 1)), ("sam12.conv2.weight", (3, n1, 1, 1)), ("sam12.conv3.weight", (n1, 3, 1, 1))]
    return out


# ----------------------------------------------------------------------------------------------------- forward (pure functions)
def _conv(x, w):
    return F.conv2d(x, w, None, 1, w.shape[-1] // 2)


def _cab_fwd(p, pre, x):
    r = _conv(F.prelu(_conv(x, p[f"{pre}.body.0.weight"]), p[f"{pre}.body.1.weight"]), p[f"{pre}.body.2.weight"])
    g = r.mean((2, 3), keepdim=True)                                               # squeeze-and-excite gate, Net.py:36-52
    g = torch.sigmoid(_conv(F.relu(_conv(g, p[f"{pre}.CA.conv_du.0.weight"])), p[f"{pre}.CA.conv_du.2.weight"]))
    return r * g + x


def _resample(x, scale):
    return F.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=False)


def _encoder(p, pre, x, enc_outs=None, dec_outs=None):
    outs = []
    for lvl in (1, 2, 3):
        for i in range(2):
            x = _cab_fwd(p, f"{pre}.encoder_level{lvl}.{i}", x)
        if enc_outs is not None:                                                   # cross-stage feature fusion, Net.py:100-113
            x = x + _conv(enc_outs[lvl - 1], p[f"{pre}.csff_enc{lvl}.weight"]) + _conv(dec_outs[lvl - 1], p[f"{pre}.csff_dec{lvl}.weight"])
        outs.append(x)
        if lvl < 3:
            x = _conv(_resample(x, 0.5), p[f"{pre}.down{lvl}{lvl + 1}.down.1.weight"])
    return outs


def _decoder(p, pre, encs):
    e1, e2, e3 = encs
    d3 = e3
    for i in range(2):
        d3 = _cab_fwd(p, f"{pre}.decoder_level3.{i}", d3)
    x = _conv(_resample(d3, 2), p[f"{pre}.up32.up.1.weight"]) + _cab_fwd(p, f"{pre}.skip_attn2", e2)
    for i in range(2):
        x = _cab_fwd(p, f"{pre}.decoder_level2.{i}", x)
    d2 = x
    x = _conv(_resample(d2, 2), p[f"{pre}.up21.up.1.weight"]) + _cab_fwd(p, f"{pre}.skip_attn1", e1)
    for i in range(2):
        x = _cab_fwd(p, f"{pre}.decoder_level1.{i}", x)
    return [x, d2, d3]


def mprnet_forward(p: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """Net.T_net.forward, Net.py:196-216: pass 1 -> image -> residual -> residual encoder; its features (x 1) plus 0.8 x the pass-1
    encoder features go through the same decoder; the SAM's image output is the result (its feature output is unused)."""
    def shallow(pre, t):
        return _cab_fwd(p, f"{pre}.1", _conv(t, p[f"{pre}.0.weight"]))

    def image(dec1):
        return _conv(dec1, p["sam12.conv2.weight"]) + x                            # SAM, Net.py:26-28 (img branch only)
    enc = _encoder(p, "stage1_encoder", shallow("shallow_feat1", x))
    res = x - image(_decoder(p, "stage1_decoder", enc)[0])
    # the reference calls the residual encoder WITHOUT encoder_outs/decoder_outs (Net.py:208): its csff_* weights are never used
    remb = _encoder(p, "stage1_resencoder", shallow("res_shallow_feat1", res))
    fused = [r + 0.8 * e for r, e in zip(remb, enc)]
    return image(_decoder(p, "stage1_decoder", fused)[0])


def fnet_forward(p: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """Net_Restormer.F_net.forward (:508-522) on stock ops; returns [B]"""
    for i, (_k, s, pad) in enumerate(_FNET_CONVS):
        x = F.leaky_relu(F.conv2d(x, p[f"features.{2 * i}.weight"], p.get(f"features.{2 * i}.bias"), s, pad), 0.2)
    x = F.linear(x.flatten(1), p["fc.weight"], p["fc.bias"])
    x = F.leaky_relu(F.linear(x, p["fc1.weight"], p["fc1.bias"]), 0.2)
    return F.linear(x, p["fc2.weight"], p["fc2.bias"]).view(-1)


# ----------------------------------------------------------------------------------------------------- module-like holders
class TorchNet:
    """A dict of torch parameters behind the slice of the nn.Module interface the loop and the checkpoints use."""

    def __init__(self, shapes: Sequence[Tuple[str, tuple]], fwd, kind: str, seed=None, device="cpu"):
        from .net_restormer import _reference_init
        self.shapes, self._fwd, self.device = list(shapes), fwd, torch.device(device)
        init = _reference_init([(n, s) for n, s in shapes if not n.endswith("body.1.weight")], kind, seed)
        slope = torch.full((1,), 0.25)                                              # nn.PReLU() default, ONE instance (Net.py:185)
        self.p: Dict[str, torch.Tensor] = OrderedDict()
        shared = None
        for n, _s in self.shapes:
            if n.endswith("body.1.weight"):
                if shared is None:
                    shared = slope.to(self.device).requires_grad_(True)
                self.p[n] = shared
            else:
                self.p[n] = init[n].to(self.device).requires_grad_(True)

    def __call__(self, x):
        return self._fwd(self.p, x)

    def parameters(self):
        seen, out = set(), []
        for t in self.p.values():
            if id(t) not in seen:
                seen.add(id(t))
                out.append(t)
        return out

    def state_dict(self):
        return OrderedDict((n, t.detach().clone()) for n, t in self.p.items())

    def load_state_dict(self, sd, strict=True):
        missing = [n for n in self.p if n not in sd]
        if strict and (missing or any(k not in self.p for k in sd)):
            raise KeyError(f"load_state_dict: missing {missing[:3]}, unexpected {[k for k in sd if k not in self.p][:3]}")
        with torch.no_grad():
            for n, t in self.p.items():
                if n in sd:
                    t.copy_(torch.as_tensor(sd[n]).to(t))

    def zero_grad(self):
        for t in self.parameters():
            t.grad = None

    def cuda(self):
        return self

    train = eval = lambda self, *a: self


def MPRNetT(seed=None, device="cpu") -> TorchNet:
    """``Net.T_net()`` (Net.py:179-216) with PyTorch's default initialisation"""
    return TorchNet(mprnet_param_shapes(), mprnet_forward, "T", seed, device)


def FNetTorch(patch_size=128, seed=None, device="cpu") -> TorchNet:
    from . import params as P
    return TorchNet(P.fnet_param_shapes(patch_size), fnet_forward, "F", seed, device)


# ----------------------------------------------------------------------------------------------------- the loop on stock ops
def torch_minimax_iteration(Tnet, Fnet, T_opt, F_opt, degraded, target, de_id, alpha, sigma, Sigma, paired):
    """trainer.py:262-346 with torch autograd: critic step, gradient-penalty step (double backward), generator step with the
    Fourier residual-guided cost.  Returns the three printed losses and the penalty."""
    B = degraded.shape[0]
    with torch.no_grad():
        fake = Tnet(degraded)
    Fnet.zero_grad()
    loss_f = -Fnet(target).mean() + Fnet(fake).mean()                               # :266-277
    loss_f.backward()
    F_opt.step()                                                                    # :280
    Fnet.zero_grad()
    a = alpha.view(B, 1, 1, 1)
    interp = (a * target + (1 - a) * fake).requires_grad_(True)                     # :284-286
    (g,) = torch.autograd.grad(Fnet(interp), interp, torch.ones(B, device=interp.device), create_graph=True)
    gp = 10.0 * ((g.flatten(1).pow(2).sum(1).sqrt() - 1) ** 2).mean()               # :300-305
    gp.backward()
    F_opt.step()                                                                    # :308
    Fnet.zero_grad()
    Tnet.zero_grad()
    out = Tnet(degraded)                                                            # :318
    res = degraded - out
    rmse = res.pow(2).mean().sqrt()
    fr = torch.fft.fft2(res)
    pen = 0.0
    for i in range(B):                                                              # :325-332 (`**1/2` is a division by two)
        pen = pen + (fr[i].abs().pow(2).mean() / 2 if int(de_id[i]) < 3 else fr[i].abs().mean())
    loss_t = -Fnet(out).mean() + sigma * (rmse + pen)
    if paired:
        loss_t = loss_t + Sigma * (out - target).abs().mean()                       # :338-340
    loss_t.backward()
    T_opt.step()                                                                    # :346
    return dict(Loss_F=float(loss_f), Loss_T=float(loss_t), Loss_mse=float(rmse), gp=float(gp))
