"""SURVEY.md 8(f4) — the reference's OLDER transport map (MPRNet-style ``Net.T_net``, Net.py:179-216) on the HIP kernels.

Same construction as ``rcot_amd/net_restormer.py``: flat fp32 parameter / gradient buffers with the reference's ``state_dict`` names,
an explicit forward / backward schedule (no autograd), every launch through the C ABI (``librcot_hip.so``), no fallback.  The
class plugs into ``MinimaxStep`` / ``PlannedMinimax`` (rcot_amd/trainer.py, plan.py) in the place of ``T_net``: BASELINE configs[0]
(``--backbone mprnet``) then runs its ten minimax steps on an MI355X instead of on stock PyTorch ops (``rcot_amd/mprnet.py`` keeps
the stock-ops form for the CPU; the two are compared tensor by tensor in tests/test_mprnet_gpu.py).

What runs where (reference sites in Net.py):
  * 3x3 convolutions (``conv``, :13-16; 3 -> 80 and C -> C for C in 80 / 128 / 176, no bias) and the 1x1 convolutions of DownSample,
    SkipUpSample and SAM (:23-24, :150, :168) — ``rcot_conv2d_fwd / _dgrad / _wgrad`` (the implicit-GEMM engine of csrc/conv_ops.hip);
  * PReLU with ONE slope shared by every CAB (``act = nn.PReLU()`` passed to all of them, :185) — ``rcot_prelu_fwd / _bwd``;
  * CALayer (:36-52): pool ``rcot_row_dot``, gate ``rcot_ca_gate_fwd / _bwd``, gated residual ``rcot_row_scale_add`` (CAB :70-72);
  * bilinear x0.5 / x2 (:149, :167) — ``rcot_bilinear_down2 / _up2`` and their adjoints.

Two choices that are not the reference's op order (results equal up to fp32 rounding):
  * SkipUpSample is ``conv1x1(upsample(x)) + y`` (:172-176).  A 1x1 convolution (channels) and a bilinear resampling (pixels) commute,
    so the product runs at the LOW resolution — a quarter of the MACs and of the operand bytes — and one launch resamples its
    (C - 48)-channel result and adds the skip: ``up2(conv1x1(x)) + y``.
  * ``T_net.forward`` calls ``self.sam12`` for its IMAGE output only (:201, :215: ``_, stage1_img``); ``conv1`` / ``conv3`` and the
    sigmoid gate feed the discarded feature output, have no gradient in the reference (``grad is None``) and are not evaluated here.
    Likewise the residual encoder is called without ``encoder_outs`` (:208): its six ``csff_*`` weights are dead parameters.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Callable, Dict, List, Optional

import torch

from .mprnet import N_FEAT, REDUCTION, SCALE, mprnet_param_shapes
from .net_restormer import ParamStore, _reference_init

_LEVEL_C = {1: N_FEAT, 2: N_FEAT + SCALE, 3: N_FEAT + 2 * SCALE}


def _is_slope(name: str) -> bool:
    return name.endswith("body.1.weight")


def _cab_names(prefix: str) -> List[str]:
    """a CAB's tensors in the order its backward finishes them"""
    return [f"{prefix}.CA.conv_du.2.weight", f"{prefix}.CA.conv_du.0.weight", f"{prefix}.body.2.weight", f"{prefix}.body.0.weight"]


def _encoder_live(pre: str) -> List[str]:
    out: List[str] = []
    for lvl in (3, 2, 1):
        if lvl < 3:
            out.append(f"{pre}.down{lvl}{lvl + 1}.down.1.weight")
        for i in (1, 0):
            out += _cab_names(f"{pre}.encoder_level{lvl}.{i}")
    return out


def mprnet_live_order() -> List[str]:
    """Live tensors in the order ``MPRNetHip.backward`` makes their gradients FINAL: the residual branch (used once), then what both
    passes share (SAM's image convolution and the decoder: final in pass 1's part of the sweep), the encoder, the shared slope."""
    d = "stage1_decoder"
    dec = _cab_names(f"{d}.decoder_level1.1") + _cab_names(f"{d}.decoder_level1.0") + _cab_names(f"{d}.skip_attn1") + [f"{d}.up21.up.1.weight"]
    dec += _cab_names(f"{d}.decoder_level2.1") + _cab_names(f"{d}.decoder_level2.0") + _cab_names(f"{d}.skip_attn2") + [f"{d}.up32.up.1.weight"]
    dec += _cab_names(f"{d}.decoder_level3.1") + _cab_names(f"{d}.decoder_level3.0")
    order = _encoder_live("stage1_resencoder") + _cab_names("res_shallow_feat1.1") + ["res_shallow_feat1.0.weight"]
    order += ["sam12.conv2.weight"] + dec
    order += _encoder_live("stage1_encoder") + _cab_names("shallow_feat1.1") + ["shallow_feat1.0.weight"]
    return order + ["shallow_feat1.1.body.1.weight"]


def mprnet_dead() -> List[str]:
    """state_dict tensors the reference's forward never reaches (their ``.grad`` stays None there; the optimizer skips them)"""
    return [f"stage1_resencoder.csff_{side}{lvl}.weight" for side in ("enc", "dec") for lvl in (1, 2, 3)] + ["sam12.conv1.weight", "sam12.conv3.weight"]


class _CAB:
    """Channel-attention block, Net.py:56-73: ``res = conv(prelu(conv(x))); res = res * CA(res); res += x``."""

    def __init__(self, net: "MPRNetHip", prefix: str):
        st, self.be, self.net = net.store, net.be, net
        self.W0, self.gW0 = st.p[f"{prefix}.body.0.weight"], st.g[f"{prefix}.body.0.weight"]
        self.W2, self.gW2 = st.p[f"{prefix}.body.2.weight"], st.g[f"{prefix}.body.2.weight"]
        n = self.W0.shape[0]
        r = n // REDUCTION
        self.D0, self.gD0 = st.p[f"{prefix}.CA.conv_du.0.weight"].view(r, n), st.g[f"{prefix}.CA.conv_du.0.weight"].view(r, n)
        self.D2, self.gD2 = st.p[f"{prefix}.CA.conv_du.2.weight"].view(n, r), st.g[f"{prefix}.CA.conv_du.2.weight"].view(n, r)
        self.slope, self.gslope = st.p[net.slope_name], st.g[net.slope_name]
        self.n, self.r = n, r
        self.W0f, self.W2f = net.flipped(f"{prefix}.body.0.weight"), net.flipped(f"{prefix}.body.2.weight")

    def forward(self, x, save: bool):
        be = self.be
        B, C, H, W = x.shape
        c1 = be.empty(B, C, H, W)
        be.conv2d_fwd(x, self.W0, None, c1, 1, 1)
        a = be.empty(B, C, H, W) if save else c1                     # (inference: in place)
        be.prelu_fwd(c1, self.slope, a)
        res = be.empty(B, C, H, W)
        be.conv2d_fwd(a, self.W2, None, res, 1, 1)
        mean, hid, gate = be.empty(B, C), be.empty(B, self.r), be.empty(B, C)
        be.row_dot(res, None, mean, 1.0 / (H * W))                   # AdaptiveAvgPool2d(1), :40
        be.ca_gate_fwd(mean, self.D0, self.D2, hid, gate)            # conv_du, :42-47
        out = be.empty(B, C, H, W) if save else res
        be.row_scale_add(res, gate, x, None, 0.0, out)               # res * y + x, :52,:71
        return out, ((x, c1, a, res, mean, hid, gate) if save else None)

    def backward(self, ctx, dout):
        """returns d/dx IN ``dout``'s storage (the residual path passes the incoming gradient through: the data gradient of the first
        convolution is accumulated onto it)"""
        be, net = self.be, self.net
        x, c1, a, res, mean, hid, gate = ctx
        B, C, H, W = x.shape
        dgate, dmean = be.empty(B, C), be.empty(B, C)
        be.row_dot(dout, res, dgate, 1.0)
        be.ca_gate_bwd(dgate, gate, hid, mean, self.D0, self.D2, self.gD0, self.gD2, dmean)
        dres = be.empty(B, C, H, W)
        be.row_scale_add(dout, gate, None, dmean, 1.0 / (H * W), dres)
        net._leaf(lambda: be.conv2d_wgrad(dres, a, self.gW2, 1, 1, 1.0), dres, a)
        da = be.empty(B, C, H, W)
        if self.W2f is not None:                                     # the data gradient as a forward product with the flipped weight
            be.conv2d_fwd(dres, self.W2f, None, da, 1, 1)
        else:
            be.conv2d_dgrad(dres, self.W2, da, 1, 1)
        be.prelu_bwd(da, c1, self.slope, da, self.gslope)
        net._leaf(lambda: be.conv2d_wgrad(da, x, self.gW0, 1, 1, 1.0), da, x)
        if self.W0f is not None:
            be.conv2d_fwd(da, self.W0f, None, dout, 1, 1, 1.0, 0, dout)      # + the residual's gradient (R = the output buffer)
        else:
            be.conv2d_dgrad(da, self.W0, dout, 1, 1, beta=1.0)
        return dout


class MPRNetHip:
    """``Net.T_net()`` (Net.py:179-216; defaults n_feat = 80, scale_unetfeats = 48, reduction = 4, bias = False) on the HIP backend.

    Interface = the slice of ``rcot_amd.net_restormer.T_net`` the training step uses: ``store`` (flat buffers), ``forward(x, save)``,
    ``backward(dout)`` (accumulates parameter gradients), ``zero_grad``, ``state_dict`` / ``load_state_dict`` with the reference's 127
    names (the shared PReLU slope is listed 22 times there, stored once here), ``grad_ready_hook`` for the bucketed all-reduce."""

    def __init__(self, backend=None, seed: Optional[int] = None):
        if backend is None:
            from .ops import default_backend
            backend = default_backend()
        self.be = be = backend
        self.shapes_all = mprnet_param_shapes()
        self.slope_name = next(n for n, _ in self.shapes_all if _is_slope(n))
        uniq = [(n, s) for n, s in self.shapes_all if not _is_slope(n) or n == self.slope_name]
        self.store = st = ParamStore(be, uniq, mprnet_live_order(), mprnet_dead())
        init = _reference_init([(n, s) for n, s in uniq if n != self.slope_name], "T", seed)
        init[self.slope_name] = torch.full((1,), 0.25)               # nn.PReLU() default
        st.load(init)
        # Flipped copies Wf[ci][co][2-ky][2-kx] of the 3x3 C -> C weights (44 tensors): with them a data gradient is a FORWARD product
        # (rcot_conv_weight_flip).  The forward kernel has a form for 64 < rows <= 80 (16 rows at a time; the data-gradient kernel computes
        # 128 rows for them: 158 -> 85 us per product at 4 x 128 x 128) and is the faster of the two at the other levels as well
        # (128 channels at 64 x 64: 74 -> 65 us).  One launch per channel count; refreshed by repack().  RCOT_MPRNET_FLIP=0: A/B.
        flip_on = os.environ.get("RCOT_MPRNET_FLIP", "1") != "0"
        self._flip_groups = []                                        # (C, names, table)
        self._flip_view: Dict[str, torch.Tensor] = {}
        names_by_c: Dict[int, List[str]] = {}
        for n, sh in uniq:
            if flip_on and len(sh) == 4 and sh[2] == 3 and sh[0] == sh[1]:
                names_by_c.setdefault(sh[0], []).append(n)
        total = sum(c * c * 9 * len(v) for c, v in names_by_c.items())
        self._flip = be.zeros(max(1, total))
        off = 0
        for c, names in sorted(names_by_c.items()):
            per, tab = c * c * 9, []
            for n in names:
                self._flip_view[n] = self._flip[off:off + per].view(c, c, 3, 3)
                tab += [st.layout.offset[n], off]
                off += per
            self._flip_groups.append((c, names, torch.tensor(tab, dtype=torch.int64, device=st.flat.device)))
        self._flip_names = [n for _c, names, _t in self._flip_groups for n in names]
        self.cab: Dict[str, _CAB] = {}
        for n, _ in uniq:
            if n.endswith(".body.0.weight"):
                pre = n[:-len(".body.0.weight")]
                self.cab[pre] = _CAB(self, pre)
        self._ctx = None
        self.repack()
        self._side_leaves = hasattr(be, "side_run")
        #: called as hook(n_final) during backward() when grad[0:n_final) of the flat buffer is final (mprnet_live_order)
        self.grad_ready_hook: Optional[Callable[[int], None]] = None

    # ------------------------------------------------------------------ nn.Module-like plumbing
    def state_dict(self):
        p = self.store.p
        return OrderedDict((n, p[self.slope_name if _is_slope(n) else n].detach().clone()) for n, _ in self.shapes_all)

    def load_state_dict(self, sd, strict: bool = True):
        missing = [n for n, _ in self.shapes_all if n not in sd]
        extra = [k for k in sd if k not in dict(self.shapes_all)]
        if strict and (missing or extra):
            raise KeyError(f"load_state_dict: missing {missing[:4]}..., unexpected {extra[:4]}...")
        self.store.load({k: v for k, v in sd.items() if k in self.store.p}, strict=False)
        self.repack()

    def zero_grad(self):
        self.store.zero_grad()

    def parameters(self):
        return [self.store.p[n] for n, _ in self.store.shapes]

    def named_parameters(self):
        return [(n, self.store.p[n]) for n, _ in self.store.shapes]

    def cuda(self):
        return self

    def train(self, mode=True):
        return self

    def eval(self):
        return self

    def flipped(self, name: str):
        return self._flip_view.get(name)

    def repack(self):
        """private weight copies follow the parameters (after every optimizer step / load): the flipped 3x3 weights, one launch"""
        for c, names, table in self._flip_groups:
            self.be.conv_weight_flip(self.store.flat, self._flip, table, len(names), c, c, 3)

    def __call__(self, x):
        return self.forward(x, save=False)

    def _leaf(self, fn, *hold):
        """a weight-gradient product: next to the data-gradient chain on the backend's side stream"""
        if self._side_leaves:
            self.be.side_run(fn, *hold)
        else:
            fn()

    def _ready(self, after_param: str):
        if self.grad_ready_hook is not None:
            self.be.side_join()
            lay = self.store.layout
            i = lay.order.index(after_param)
            self.grad_ready_hook(lay.offset[lay.order[i + 1]] if i + 1 < len(lay.order) else lay.n_live)

    # ------------------------------------------------------------------ pieces
    def _conv(self, x, name: str, R=None):
        """bias-free convolution with the weight ``name`` (3x3 pad 1 or 1x1)"""
        Wt = self.store.p[name]
        y = self.be.empty(x.shape[0], Wt.shape[0], x.shape[2], x.shape[3])
        self.be.conv2d_fwd(x, Wt, None, y, 1, Wt.shape[2] // 2, 1.0, 0, R)
        return y

    def _conv_bwd(self, x, dy, name: str, need_dx: bool = True):
        be, Wt = self.be, self.store.p[name]
        gW, pad = self.store.g[name], Wt.shape[2] // 2
        self._leaf(lambda: be.conv2d_wgrad(dy, x, gW, 1, pad, 1.0), dy, x)
        if not need_dx:
            return None
        dx = be.empty(*x.shape)
        be.conv2d_dgrad(dy, Wt, dx, 1, pad)
        return dx

    def _shallow(self, pre: str, x, save: bool):
        f = self._conv(x, f"{pre}.0.weight")
        out, c = self.cab[f"{pre}.1"].forward(f, save)
        return out, (x, c)

    def _shallow_bwd(self, pre: str, ctx, g, need_dx: bool):
        x, c = ctx
        g = self.cab[f"{pre}.1"].backward(c, g)
        return self._conv_bwd(x, g, f"{pre}.0.weight", need_dx)

    def _encoder(self, pre: str, x, save: bool):
        """Encoder.forward without cross-stage fusion, Net.py:98-116 (both encoders are called that way, :199,:208)"""
        be = self.be
        outs, cabs, pooled = [], [], []
        for lvl in (1, 2, 3):
            for i in range(2):
                x, c = self.cab[f"{pre}.encoder_level{lvl}.{i}"].forward(x, save)
                cabs.append(c)
            outs.append(x)
            if lvl < 3:
                B, C, H, W = x.shape
                pl = be.empty(B, C, H // 2, W // 2)
                be.bilinear_down2(x, pl)                                                # DownSample, :146-154
                pooled.append(pl)
                x = self._conv(pl, f"{pre}.down{lvl}{lvl + 1}.down.1.weight")
        return outs, (cabs, pooled)

    def _encoder_bwd(self, pre: str, ctx, denc):
        """``denc``: gradients w.r.t. the three outputs (consumed); returns the gradient w.r.t. the input"""
        be = self.be
        cabs, pooled = ctx
        g = denc[2]
        for lvl in (3, 2, 1):
            for i in (1, 0):
                g = self.cab[f"{pre}.encoder_level{lvl}.{i}"].backward(cabs[2 * (lvl - 1) + i], g)
            if lvl > 1:
                dpl = self._conv_bwd(pooled[lvl - 2], g, f"{pre}.down{lvl - 1}{lvl}.down.1.weight")
                be.bilinear_down2_bwd(dpl, denc[lvl - 2], beta=1.0)                     # joins the upper level's own gradient
                g = denc[lvl - 2]
        return g

    def _decoder(self, encs, save: bool):
        """Decoder.forward, Net.py:136-144; only dec1 is used by T_net.forward (:201,:215)"""
        be = self.be
        d = "stage1_decoder"
        e1, e2, e3 = encs
        cabs: Dict[str, object] = {}

        def run(name, x):
            y, cabs[name] = self.cab[name].forward(x, save)
            return y
        x = run(f"{d}.decoder_level3.1", run(f"{d}.decoder_level3.0", e3))
        d3 = x
        t = self._conv(d3, f"{d}.up32.up.1.weight")                                     # SkipUpSample with the 1x1 in front (module doc)
        s = run(f"{d}.skip_attn2", e2)
        x = be.empty(*s.shape)
        be.bilinear_up2(t, s, x)
        x = run(f"{d}.decoder_level2.1", run(f"{d}.decoder_level2.0", x))
        d2 = x
        t = self._conv(d2, f"{d}.up21.up.1.weight")
        s = run(f"{d}.skip_attn1", e1)
        x = be.empty(*s.shape)
        be.bilinear_up2(t, s, x)
        x = run(f"{d}.decoder_level1.1", run(f"{d}.decoder_level1.0", x))
        return x, (cabs, d2, d3)

    def _decoder_bwd(self, ctx, g):
        """``g``: gradient w.r.t. dec1 (consumed); returns the gradients w.r.t. the three encoder features"""
        be = self.be
        d = "stage1_decoder"
        cabs, d2, d3 = ctx

        def back(name, g):
            return self.cab[name].backward(cabs[name], g)
        g = back(f"{d}.decoder_level1.0", back(f"{d}.decoder_level1.1", g))
        dt = be.empty(*d2.shape[:1], _LEVEL_C[1], d2.shape[2], d2.shape[3])
        be.bilinear_up2_bwd(g, dt)
        de1 = back(f"{d}.skip_attn1", g)
        g = self._conv_bwd(d2, dt, f"{d}.up21.up.1.weight")
        g = back(f"{d}.decoder_level2.0", back(f"{d}.decoder_level2.1", g))
        dt = be.empty(*d3.shape[:1], _LEVEL_C[2], d3.shape[2], d3.shape[3])
        be.bilinear_up2_bwd(g, dt)
        de2 = back(f"{d}.skip_attn2", g)
        g = self._conv_bwd(d3, dt, f"{d}.up32.up.1.weight")
        de3 = back(f"{d}.decoder_level3.0", back(f"{d}.decoder_level3.1", g))
        return [de1, de2, de3]

    # ------------------------------------------------------------------ the network
    def forward(self, x, save: bool = False):
        """T_net.forward, Net.py:196-216.  x: [B, 3, H, W] fp32, H and W multiples of 4."""
        be = self.be
        x = x.contiguous()
        if x.shape[2] % 4 or x.shape[3] % 4:
            raise ValueError("MPRNetHip: H and W must be multiples of 4 (two x0.5 levels; the reference's testers crop to that)")
        f0, c_sf = self._shallow("shallow_feat1", x, save)
        enc, c_enc = self._encoder("stage1_encoder", f0, save)
        d1, c_dec1 = self._decoder(enc, save)
        img1 = self._conv(d1, "sam12.conv2.weight", R=x)                                # SAM: img = conv2(x) + x_img, :27
        res = be.empty(*x.shape)
        be.axpby(x, img1, res, 1.0, -1.0)                                               # :204
        g0, c_rsf = self._shallow("res_shallow_feat1", res, save)
        remb, c_renc = self._encoder("stage1_resencoder", g0, save)
        for r, e in zip(remb, enc):
            be.axpby(r, e, r, 1.0, 0.8)                                                 # :209 (in place: nothing reads remb again)
        d1b, c_dec2 = self._decoder(remb, save)
        out = self._conv(d1b, "sam12.conv2.weight", R=x)
        if save:
            self._ctx = (c_sf, c_enc, c_dec1, d1, c_rsf, c_renc, c_dec2, d1b)
        return out

    def backward(self, dout):
        """dout = d(loss)/d(output); accumulates every live parameter gradient (zero_grad() first)"""
        be = self.be
        c_sf, c_enc, c_dec1, d1, c_rsf, c_renc, c_dec2, d1b = self._ctx
        self._ctx = None
        dout = dout.contiguous()
        # pass 2 (the output): SAM image convolution, decoder
        g = self._conv_bwd(d1b, dout, "sam12.conv2.weight")
        dfused = self._decoder_bwd(c_dec2, g)
        denc = []
        for t in dfused:                                                                # fused = remb + 0.8 enc
            e = be.empty(*t.shape)
            be.axpby(t, None, e, 0.8, 0.0)
            denc.append(e)
        g = self._encoder_bwd("stage1_resencoder", c_renc, dfused)
        dres = self._shallow_bwd("res_shallow_feat1", c_rsf, g, need_dx=True)
        self._ready("res_shallow_feat1.0.weight")
        # pass 1: res = x - img1, img1 = conv2(dec1) + x
        dimg = be.empty(*dres.shape)
        be.axpby(dres, None, dimg, -1.0, 0.0)
        g = self._conv_bwd(d1, dimg, "sam12.conv2.weight")
        de = self._decoder_bwd(c_dec1, g)
        self._ready("stage1_decoder.decoder_level3.0.body.0.weight")
        for a, t in zip(denc, de):
            be.axpby(a, t, a, 1.0, 1.0)
        g = self._encoder_bwd("stage1_encoder", c_enc, denc)
        self._shallow_bwd("shallow_feat1", c_sf, g, need_dx=False)
        if self._side_leaves:
            be.side_join()
        self._ready(self.slope_name)
        return None
