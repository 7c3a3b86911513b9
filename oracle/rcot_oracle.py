"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (rcot_amd/).

CPU restatement (plain PyTorch fp32/fp64 tensor math, autograd for derivatives) of the
reference's RCOT hot path, written from the mathematics in SURVEY.md Appendix A and
checked against the imported reference by ``oracle/pin_against_reference.py`` (run in the
build container, where /root/reference exists).  The reference has no tests of its own for
this path, so the pins are: (i) direct agreement with the imported
reference on seeded inputs, recorded in oracle/PINNED.md, and (ii) the golden fixtures
under tests/golden/ produced *by the reference* with that script.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Each function cites the reference lines it restates (paths relative to /root/reference).
All functions are pure: parameters come in as a ``dict name -> tensor`` using the
reference's state_dict names.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------- blocks
def layernorm_c(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """WithBias LayerNorm over channels per pixel; Net_Restormer.py:186-189,198-200.
    x: [B,C,H,W]."""
    mu = x.mean(dim=1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=1, keepdim=True)          # biased variance
    return (x - mu) / torch.sqrt(var + 1e-5) * w.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


def mdta(x: torch.Tensor, p: Params, pre: str, heads: int) -> torch.Tensor:
    """Multi-DConv head transposed attention; Net_Restormer.py:29-50."""
    B, C, H, W = x.shape
    t = F.conv2d(x, p[pre + ".qkv.weight"])
    u = F.conv2d(t, p[pre + ".qkv_dwconv.weight"], padding=1, groups=3 * C)
    q, k, v = u.reshape(B, 3, heads, C // heads, H * W).unbind(1)
    qn = q / q.norm(dim=-1, keepdim=True).clamp_min(1e-12)    # F.normalize, :39-40
    kn = k / k.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    g = qn @ kn.transpose(-2, -1) * p[pre + ".temperature"].view(1, heads, 1, 1)
    a = torch.softmax(g, dim=-1)
    o = (a @ v).reshape(B, C, H, W)
    return F.conv2d(o, p[pre + ".project_out.weight"])


def gdfn(x: torch.Tensor, p: Params, pre: str) -> torch.Tensor:
    """Gated-Dconv feed-forward; Net_Restormer.py:80-85 (exact erf GELU)."""
    t = F.conv2d(x, p[pre + ".project_in.weight"])
    d = F.conv2d(t, p[pre + ".dwconv.weight"], padding=1, groups=t.shape[1])
    hid = d.shape[1] // 2
    g = F.gelu(d[:, :hid]) * d[:, hid:]
    return F.conv2d(g, p[pre + ".project_out.weight"])


def transformer_block(x: torch.Tensor, p: Params, pre: str, heads: int) -> torch.Tensor:
    """Net_Restormer.py:210-214."""
    x = x + mdta(layernorm_c(x, p[pre + ".norm1.body.weight"], p[pre + ".norm1.body.bias"]),
                 p, pre + ".attn", heads)
    x = x + gdfn(layernorm_c(x, p[pre + ".norm2.body.weight"], p[pre + ".norm2.body.bias"]),
                 p, pre + ".ffn")
    return x


def stage(x, p, prefix, n, heads):
    for i in range(n):
        x = transformer_block(x, p, f"{prefix}.{i}", heads)
    return x


def downsample(x, w):
    """3x3 conv C->C/2 + PixelUnshuffle(2); Net_Restormer.py:86-94."""
    return F.pixel_unshuffle(F.conv2d(x, w, padding=1), 2)


def upsample(x, w):
    """3x3 conv C->2C + PixelShuffle(2); Net_Restormer.py:103-111."""
    return F.pixel_shuffle(F.conv2d(x, w, padding=1), 2)


# ----------------------------------------------------------------------------- T_net
def tnet_forward(p: Params, inp: torch.Tensor, decoder: bool = True,
                 return_res: bool = False):
    """Two-pass Restormer transport map; Net_Restormer.py:328-434.

    Pass 1 restores, its residual is re-encoded by the res-encoder (sharing patch_embed
    and down3_4, :381,:393), and pass 2 re-runs latent/decoder with
    ``latent += 0.8*reslatent`` (:401) and the pass-1 skip tensors."""
    nb = (4, 6, 6, 8)
    hd = (1, 2, 4, 8)

    def decode(latent):
        if decoder:
            latent = transformer_block(latent, p, "noise_level3", hd[2])          # :346
            latent = F.conv2d(latent, p["reduce_noise_level3.weight"])           # :347
        d3 = upsample(latent, p["up4_3.body.0.weight"])
        d3 = F.conv2d(torch.cat([d3, e3], 1), p["reduce_chan_level3.weight"])
        d3 = stage(d3, p, "decoder_level3", nb[2], hd[2])
        if decoder:
            d3 = transformer_block(d3, p, "noise_level2", hd[2])
            d3 = F.conv2d(d3, p["reduce_noise_level2.weight"])
        d2 = upsample(d3, p["up3_2.body.0.weight"])
        d2 = F.conv2d(torch.cat([d2, e2], 1), p["reduce_chan_level2.weight"])
        d2 = stage(d2, p, "decoder_level2", nb[1], hd[1])
        if decoder:
            d2 = transformer_block(d2, p, "noise_level1", hd[2])                  # heads[2]=4 on 96 ch, :313
            d2 = F.conv2d(d2, p["reduce_noise_level1.weight"])
        d1 = upsample(d2, p["up2_1.body.0.weight"])
        d1 = torch.cat([d1, e1], 1)
        d1 = stage(d1, p, "decoder_level1", nb[0], hd[0])
        d1 = stage(d1, p, "refinement", 4, hd[0])
        return F.conv2d(d1, p["output.weight"], padding=1) + inp                  # :375 / :432

    e1 = stage(F.conv2d(inp, p["patch_embed.proj.weight"], padding=1), p, "encoder_level1", nb[0], hd[0])
    e2 = stage(downsample(e1, p["down1_2.body.0.weight"]), p, "encoder_level2", nb[1], hd[1])
    e3 = stage(downsample(e2, p["down2_3.body.0.weight"]), p, "encoder_level3", nb[2], hd[2])
    l4_in = downsample(e3, p["down3_4.body.0.weight"])
    latent = stage(l4_in, p, "latent", nb[3], hd[3])
    out1 = decode(latent)
    res = inp - out1                                                              # :377

    r1 = stage(F.conv2d(res, p["patch_embed.proj.weight"], padding=1), p, "resencoder_level1", nb[0], hd[0])
    r2 = stage(downsample(r1, p["resdown1_2.body.0.weight"]), p, "resencoder_level2", nb[1], hd[1])
    r3 = stage(downsample(r2, p["resdown2_3.body.0.weight"]), p, "resencoder_level3", nb[2], hd[2])
    r4 = stage(downsample(r3, p["down3_4.body.0.weight"]), p, "reslatent", nb[3], hd[3])
    latent2 = stage(l4_in, p, "latent", nb[3], hd[3])                             # :397 recomputed
    if decoder:
        latent2 = latent2 + 0.8 * r4                                              # :401
    out2 = decode(latent2)
    return (out2, res) if return_res else out2


# ----------------------------------------------------------------------------- F_net
_FNET_CONVS = ((5, 1, 2), (4, 2, 1), (3, 1, 1), (4, 2, 1), (3, 1, 1),
               (4, 2, 1), (3, 1, 1), (4, 2, 1), (3, 1, 1), (4, 2, 1))


def fnet_forward(p: Params, x: torch.Tensor) -> torch.Tensor:
    """WGAN critic / OT potential; Net_Restormer.py:508-522.  Returns [B]."""
    for i, (_k, s, pad) in enumerate(_FNET_CONVS):
        x = F.conv2d(x, p[f"features.{2 * i}.weight"], p.get(f"features.{2 * i}.bias"),
                     stride=s, padding=pad)
        x = F.leaky_relu(x, 0.2)
    x = x.reshape(x.shape[0], -1)
    x = F.linear(x, p["fc.weight"], p["fc.bias"])
    x = F.linear(x, p["fc1.weight"], p["fc1.bias"])          # no activation between fc and fc1
    x = F.leaky_relu(x, 0.2)
    x = F.linear(x, p["fc2.weight"], p["fc2.bias"])
    return x.view(-1, 1).squeeze(1)


# ----------------------------------------------------------------------------- losses
def fourier_penalty(res: torch.Tensor, de_id: Sequence[int]) -> torch.Tensor:
    """trainer.py:323-332.  NB ``**1/2`` parses as ``(...**1)/2`` -> halving, not sqrt;
    the penalty is SUMMED over the batch."""
    fr = torch.fft.fft2(res)
    tot = res.new_zeros(())
    for i in range(res.shape[0]):
        a = fr[i].abs()
        if int(de_id[i]) < 3:
            tot = tot + (a ** 2).mean() / 2
        else:
            tot = tot + a.mean()
    return tot


def ot_cost(degraded, restored, de_id) -> Tuple[torch.Tensor, torch.Tensor]:
    """rmse + Fourier residual penalty; trainer.py:320-332.  Returns (rmse, fourier)."""
    res = degraded - restored
    rmse = torch.sqrt(torch.mean(res ** 2))
    return rmse, fourier_penalty(res, de_id)


def gradient_penalty(pF: Params, interp: torch.Tensor) -> torch.Tensor:
    """10 * mean((||dF/dx||_2 - 1)^2); trainer.py:288-305."""
    interp = interp.detach().requires_grad_(True)
    out = fnet_forward(pF, interp)
    (g,) = torch.autograd.grad(out, interp, torch.ones_like(out), create_graph=True)
    n = torch.sqrt((g.reshape(g.shape[0], -1) ** 2).sum(1))
    return 10.0 * ((n - 1) ** 2).mean()


# ----------------------------------------------------------------------------- optimizers
class RMSprop:
    """torch.optim.RMSprop defaults (alpha .99, eps 1e-8, no momentum/centering), the
    reference's default optimizer (trainer.py:124-126).  Parameters whose grad is None
    are skipped (state untouched) exactly as torch does."""

    def __init__(self, params: Params, lr: float, alpha: float = 0.99, eps: float = 1e-8):
        self.p, self.lr, self.alpha, self.eps = params, lr, alpha, eps
        self.sq = {k: torch.zeros_like(v) for k, v in params.items()}

    def step(self, grads: Dict[str, Optional[torch.Tensor]]):
        with torch.no_grad():
            for k, g in grads.items():
                if g is None:
                    continue
                self.sq[k].mul_(self.alpha).addcmul_(g, g, value=1 - self.alpha)
                self.p[k].addcdiv_(g, self.sq[k].sqrt().add_(self.eps), value=-self.lr)


class Adam:
    """torch.optim.Adam defaults (betas .9/.999, eps 1e-8); trainer.py:121-123."""

    def __init__(self, params: Params, lr: float, b1=0.9, b2=0.999, eps=1e-8):
        self.p, self.lr, self.b1, self.b2, self.eps = params, lr, b1, b2, eps
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = {k: 0 for k in params}

    def step(self, grads):
        with torch.no_grad():
            for k, g in grads.items():
                if g is None:
                    continue
                self.t[k] += 1
                t = self.t[k]
                self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
                self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                bc1, bc2 = 1 - self.b1 ** t, 1 - self.b2 ** t
                denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
                self.p[k].addcdiv_(self.m[k], denom, value=-self.lr / bc1)


# ----------------------------------------------------------------------------- the step
def _grads(loss, params: Params) -> Dict[str, Optional[torch.Tensor]]:
    names = list(params)
    gs = torch.autograd.grad(loss, [params[n] for n in names], allow_unused=True)
    return dict(zip(names, gs))


def minimax_iteration(pT: Params, pF: Params, optT, optF, degraded, target, de_id, alpha,
                      sigma: float, Sigma: float, paired: bool, t_forward=tnet_forward):
    """One iteration of trainer.train(); trainer.py:262-346.

    critic step (WGAN loss) -> separate gradient-penalty step -> generator step with
    ``-mean F(T(x)) + sigma*(rmse + fourier) [+ Sigma*L1]``.  ``alpha`` ([B,1,1,1]) is
    injected (the reference draws it from the CPU RNG, :284).  ``paired`` is the
    reference's ``iteration < pairnum // batchSize`` (:338).  Returns the three scalars
    the reference prints (:348-354) plus the GP value."""
    for d in (pT, pF):
        for v in d.values():
            v.requires_grad_(True)
    # ---- critic ("F-sub"), :266-280
    with torch.no_grad():
        fake = t_forward(pT, degraded)
    f_loss = -fnet_forward(pF, target).mean() + fnet_forward(pF, fake).mean()
    optF.step(_grads(f_loss, pF))
    # ---- gradient penalty, its own optimizer step, :283-308
    interp = alpha * target + (1 - alpha) * fake
    gp = gradient_penalty(pF, interp)
    optF.step(_grads(gp, pF))
    # ---- generator ("T-sub"), :311-346
    out = t_forward(pT, degraded)
    out_disc = fnet_forward(pF, out)
    rmse, four = ot_cost(degraded, out, de_id)
    t_loss = -out_disc.mean() + sigma * (rmse + four)
    if paired:
        t_loss = t_loss + Sigma * (out - target).abs().mean()
    optT.step(_grads(t_loss, pT))
    return dict(Loss_F=float(f_loss), Loss_T=float(t_loss), Loss_mse=float(rmse), gp=float(gp))


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    """skimage PSNR with data_range=1 as the reference's evaluate() uses; trainer.py:225."""
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 100.0 if mse == 0 else 10.0 * math.log10(1.0 / mse)
