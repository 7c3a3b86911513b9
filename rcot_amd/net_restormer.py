"""Host-side mirror of the reference's ``Net_Restormer.py`` interface for the RCOT hot path.

``T_net`` (two-pass Restormer transport map, reference Net_Restormer.py:215-434) and ``F_net``
(WGAN-GP potential, :436-522) keep the reference's constructor arguments, ``state_dict`` names /
shapes / order and call semantics, but contain no PyTorch compute: ``forward`` / ``backward`` are
explicit schedules of librcot_hip.so launches (through a backend object) over activations that stay
resident in HBM.  Parameters and gradients live in flat fp32 buffers ordered by the time their
gradient becomes final in the backward sweep, so the data-parallel layer can all-reduce completed
ranges while the sweep continues and one fused optimizer launch updates the whole network.

Algebraic savings vs the reference schedule (results unchanged):
  * ``latent`` is evaluated once; the reference recomputes it on the same input (:397).  Its
    backward runs once on the sum of both upstream gradients.
  * torch.cat + 1x1 reduce (:351-352, :360-361) is a two-source 1x1 projection (no cat buffer).
  * LayerNorm is applied while the 1x1 projection stages its operand; only (mu, rstd) are stored.
  * ``attn @ v`` and ``project_out`` are one projection with the per-image matrix W_o*blockdiag(A).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Callable, Dict, List, Optional

import numpy as np
import os

import torch

from . import params as P

# A/B switch (profiles/r05_ab_wgrad_after*.txt): a side-stream 1x1 weight gradient starts BEHIND its data gradient instead of with it
# which of a block's three 1x1 weight gradients start BEHIND their data gradient instead of with it: bit 0 project_out, bit 1 project_in,
# bit 2 qkv (A/B switch; "1" of round 5 = all three = 7)
_WGRAD_AFTER = {"0": 0, "1": 7}.get(os.environ.get("RCOT_WGRAD_AFTER", "0"), None)
if _WGRAD_AFTER is None:
    _WGRAD_AFTER = int(os.environ["RCOT_WGRAD_AFTER"])


# =============================================================================== parameter store
class ParamStore:
    """Flat fp32 parameter / gradient buffers with named views (reference state_dict names)."""

    def __init__(self, be, shapes, live_order: List[str], dead: List[str]):
        self.be = be
        self.shapes = list(shapes)
        self.layout = P.make_layout(shapes, live_order, dead)
        self.flat = be.zeros(self.layout.n_total)
        self.grad = be.zeros(self.layout.n_total)
        self.p: Dict[str, torch.Tensor] = {}
        self.g: Dict[str, torch.Tensor] = {}
        for name, shp in self.shapes:
            o = self.layout.offset[name]
            n = int(np.prod(shp))
            self.p[name] = self.flat[o:o + n].view(*shp)
            self.g[name] = self.grad[o:o + n].view(*shp)

    def load(self, sd, strict: bool = True):
        missing = [n for n, _ in self.shapes if n not in sd]
        extra = [k for k in sd if k not in self.p]
        if strict and (missing or extra):
            raise KeyError(f"load_state_dict: missing {missing[:4]}..., unexpected {extra[:4]}...")
        for n, shp in self.shapes:
            if n in sd:
                v = sd[n]
                v = torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v.detach()
                if tuple(v.shape) != tuple(shp):
                    raise ValueError(f"{n}: shape {tuple(v.shape)} != {tuple(shp)}")
                self.p[n].copy_(v.to(dtype=self.flat.dtype))

    def state_dict(self):
        return OrderedDict((n, self.p[n].detach().clone()) for n, _ in self.shapes)

    def zero_grad(self):
        self.be.fill(self.grad, 0.0)


def _reference_init(shapes, kind: str, seed: Optional[int]):
    """Parameter distributions of the reference's constructors: PyTorch's defaults — Conv2d / Linear weights
    and biases U(-1/sqrt(fan_in), 1/sqrt(fan_in)) with the fan_in of the layer's OWN weight — except F_net's Conv2d weights,
    N(0, 0.02) (Net_Restormer.py:501-503; the biases keep the default); LayerNorm weight 1 / bias 0 (:179-181) and temperature 1
    (:22).  Uses torch's CPU generator."""
    g = torch.Generator()
    if seed is not None:
        g.manual_seed(seed)
    else:
        g.seed()
    by_name = dict(shapes)
    out = {}
    for name, shp in shapes:
        if name.endswith("body.weight") or name.endswith("temperature"):
            t = torch.ones(shp)
        elif name.endswith("body.bias"):
            t = torch.zeros(shp)
        elif name.endswith(".bias"):
            wshape = by_name[name[:-len("bias")] + "weight"]
            t = (torch.rand(shp, generator=g) * 2 - 1) / np.sqrt(int(np.prod(wshape[1:])))
        elif kind == "F" and name.startswith("features."):
            t = torch.randn(shp, generator=g) * 0.02
        else:
            t = (torch.rand(shp, generator=g) * 2 - 1) / np.sqrt(int(np.prod(shp[1:])))
        out[name] = t
    return out


# =============================================================================== operators
class TransformerBlockOp:
    """x + MDTA(LN(x)); then + GDFN(LN(.))  — Net_Restormer.py:201-214 (math: SURVEY.md A.1-A.3)."""

    def __init__(self, be, store: ParamStore, prefix: str, dim: int, heads: int):
        self.be, self.C, self.heads, self.c, self.hid = be, dim, heads, dim // heads, P.ffn_hidden(dim)
        p, g = store.p, store.g
        n = lambda s: f"{prefix}.{s}"
        self.names = [k for k, _ in P.block_param_shapes(prefix, dim, heads)]
        self.w1, self.b1 = p[n("norm1.body.weight")], p[n("norm1.body.bias")]
        self.w2, self.b2 = p[n("norm2.body.weight")], p[n("norm2.body.bias")]
        self.temp = p[n("attn.temperature")].view(heads)
        self.Wqkv = p[n("attn.qkv.weight")].view(3 * dim, dim)
        self.Wdw = p[n("attn.qkv_dwconv.weight")].view(3 * dim, 9)
        self.Wo = p[n("attn.project_out.weight")].view(dim, dim)
        self.Win = p[n("ffn.project_in.weight")].view(2 * self.hid, dim)
        self.Wdw2 = p[n("ffn.dwconv.weight")].view(2 * self.hid, 9)
        self.Wout = p[n("ffn.project_out.weight")].view(dim, self.hid)
        self.gw1, self.gb1 = g[n("norm1.body.weight")], g[n("norm1.body.bias")]
        self.gw2, self.gb2 = g[n("norm2.body.weight")], g[n("norm2.body.bias")]
        self.gtemp = g[n("attn.temperature")].view(heads)
        self.gWqkv = g[n("attn.qkv.weight")].view(3 * dim, dim)
        self.gWdw = g[n("attn.qkv_dwconv.weight")].view(3 * dim, 9)
        self.gWo = g[n("attn.project_out.weight")].view(dim, dim)
        self.gWin = g[n("ffn.project_in.weight")].view(2 * self.hid, dim)
        self.gWdw2 = g[n("ffn.dwconv.weight")].view(2 * self.hid, 9)
        self.gWout = g[n("ffn.project_out.weight")].view(dim, self.hid)
        # private K-major repacks of the four 1x1 weights (refreshed by repack() after every optimizer step)
        # pack = (WT, WP, fold, split): fold = (WTf, c12) for the two projections behind a LayerNorm (LN-folded operand + row
        # constants), split = (WTs, WPs, WTfs) the pre-split bf16 fragment packs of the bf16x3 producer / consumer kernel
        # + split6 = the three-term form of ``split`` (bf16x6 arithmetic) when the backend keeps those packs, else None
        def mk(W, folded=False):
            st, sp = be.split_shapes(*W.shape)
            fold = tuple(be.zeros(*s) for s in be.fold_shapes(*W.shape)) if folded else None
            split6 = None
            if getattr(be, "x6_packs", False):
                st6, sp6 = be.split6_shapes(*W.shape)
                split6 = (be.zeros(*st6), be.zeros(*sp6), be.zeros(*st6) if folded else None)
            return tuple(be.zeros(*s) for s in be.pack_shapes(*W.shape)) + (fold, (be.zeros(*st), be.zeros(*sp), be.zeros(*st) if folded else None), split6)
        self.pk_qkv, self.pk_o, self.pk_in, self.pk_out = mk(self.Wqkv, True), mk(self.Wo), mk(self.Win, True), mk(self.Wout)

    def pack_items(self):
        return [(self.Wqkv, self.pk_qkv[0], self.pk_qkv[1], (self.w1, self.b1) + self.pk_qkv[2], self.pk_qkv[3], self.pk_qkv[4]),
                (self.Wo, self.pk_o[0], self.pk_o[1], None, self.pk_o[3], self.pk_o[4]),
                (self.Win, self.pk_in[0], self.pk_in[1], (self.w2, self.b2) + self.pk_in[2], self.pk_in[3], self.pk_in[4]),
                (self.Wout, self.pk_out[0], self.pk_out[1], None, self.pk_out[3], self.pk_out[4])]

    def repack(self):
        for W, WT, WP, fold, split, split6 in self.pack_items():
            self.be.pack_weight(W, WT, WP, fold, split, split6)

    def _wgrad(self, dY, X, gW, ln, part):
        """gW += dY LN?(X)^T: as slabs in third ``part`` of the workspace (descriptor for block_param_reduce) when the backend
        and the shape allow, else the complete product now (returns None)."""
        be = self.be
        d = be.conv1x1_wgrad_slabs(dY, X, gW, ln=ln, region=(part, 3))
        if d is None:
            be.conv1x1_wgrad(dY, X, gW, ln=ln, beta=1.0)
        return d

    def _dgrad_wgrad(self, slabs, W, dY, dX, X, gW, ln, packed, part):
        """The two products of one incoming gradient of a 1x1 projection: dX = W^T dY and gW += dY LN?(X)^T (as slabs in third
        ``part`` of the slab arena where the backend can).  ONE launch where the backend has the paired kernel (workgroups of both
        products in one grid); else the weight gradient goes to the side stream next to the data-gradient chain."""
        be = self.be
        pair = getattr(be, "conv1x1_dgrad_wgrad_slabs", None)
        d = pair(W, dY, dX, X, gW, ln=ln, packed=packed, region=(part, 3)) if pair is not None else None
        if d is not None:
            slabs.append(d)
            return
        if not getattr(be, "side_wgrad", True):
            be.conv1x1_dgrad(W, dY, dX, packed=packed)
            slabs.append(self._wgrad(dY, X, gW, ln, part))
            return
        hold = (dY, X) + ((ln[0], ln[1]) if ln is not None else ())
        # (the weight gradient starts WITH its data gradient: started behind it — next to the bandwidth- / latency-bound kernels that follow
        # instead of next to another MFMA-bound product — it closes the block later: 77.2 -> 81.3 ms per iteration in exact fp32, 74.0 -> 76.5
        # in bf16x6, profiles/r05_ab_wgrad_after.txt)
        if (_WGRAD_AFTER >> part) & 1:
            be.conv1x1_dgrad(W, dY, dX, packed=packed)
            be.side_run(lambda: slabs.append(self._wgrad(dY, X, gW, ln, part)), *hold)
            return
        be.side_run(lambda: slabs.append(self._wgrad(dY, X, gW, ln, part)), *hold)
        be.conv1x1_dgrad(W, dY, dX, packed=packed)

    def _woT_heads(self, B):
        """W_o^T (from the pack) as [B (broadcast), heads, c, C]: rows h*c+i of W_o^T for every head."""
        return self.pk_o[0].view(self.heads, self.c, self.C).unsqueeze(0).expand(B, -1, -1, -1)

    def _wo_heads(self, B):
        """W_o as [B (broadcast), heads, C, c]: the column block of every head."""
        return self.Wo.view(self.C, self.heads, self.c).permute(1, 0, 2).unsqueeze(0).expand(B, -1, -1, -1)

    def _head_cols(self, M):
        """[B, C, C] -> [B, heads, C, c] view of the per-head column blocks."""
        return M.view(M.shape[0], self.C, self.heads, self.c).permute(0, 2, 1, 3)

    def _qkv_views(self, u):
        B, C3, H, W = u.shape
        N = H * W
        uu = u.view(B, 3, self.heads, self.c, N)
        return uu[:, 0], uu[:, 1], u.view(B, 3, self.C, N)[:, 2].unsqueeze(1)

    def forward(self, x, save: bool, wmask: Optional[int] = None, in_stats=None, want_stats: bool = False):
        """``in_stats`` = (mu, rs): the norm1 statistics of ``x``, made by the launch that stored ``x`` (the previous block of the stage);
        ``want_stats``: leave the norm1 statistics of the RESULT in ``self.out_stats`` for the next block (both only where the backend's
        stats_ok() holds: exact fp32, C <= 96, 128-pixel tiles — else ``out_stats`` stays None and every LayerNorm makes its own).
        ``wmask`` (whole-image validation only): the planes are padded on the right from ``wmask`` real columns to a width
        that makes H*W a multiple of 4 (the kernels move pixels in 16-byte pieces); the padding columns are re-zeroed before
        every spatial or pixel-reducing operation, so that the real pixels see exactly the zero padding / sums of the unpadded
        plane.  Not available with ``save`` (training patches never need it)."""
        assert not (save and wmask is not None)
        be, C, hd, c, hid = self.be, self.C, self.heads, self.c, self.hid
        B, _, H, W = x.shape
        N = H * W
        fast = be.kmajor_worth(C, N, B)           # K-major LDS-DMA GEMM (csrc/gemm_glds.hip): full 128-pixel tiles, >= 256 workgroups
        # round 6: where one row tile of the exact-fp32 kernel holds every channel of its pixels (C <= 96) the LayerNorm statistics of a
        # tensor are made by the epilogue of the product that STORES it (y below for norm2; the block's result for the next block's norm1)
        pstats = wmask is None and getattr(be, "stats_ok", None) is not None and be.stats_ok(C, N, B)
        self.out_stats = None
        t = be.empty(B, 3 * C, H, W)
        if in_stats is not None and pstats:
            mu1, rs1 = in_stats
            be.conv1x1_fwd(self.Wqkv, x, t, ln=(mu1, rs1, self.w1, self.b1), packed=self.pk_qkv)
        else:
            mu1, rs1 = be.empty(B, N), be.empty(B, N)
            # (mu1, rs1) are made by the projection kernel itself where it can (rcot_gemm_kmajor ln_compute), else by rcot_ln_stats
            be.conv1x1_fwd(self.Wqkv, x, t, ln=(mu1, rs1, self.w1, self.b1), packed=self.pk_qkv, ln_compute=True)
        if wmask is not None:
            t[..., wmask:].zero_()
        u = be.empty(B, 3 * C, H, W)
        be.dwconv3x3(t, self.Wdw, u)
        if wmask is not None:
            u[..., wmask:].zero_()
        sq = be.empty(B, 2 * C)
        Q, K, V = self._qkv_views(u)
        Gn, A, MfT = be.empty(B, hd, c, c), be.empty(B, hd, c, c), be.empty(B, C, C)
        # MfT[b][h*c+j][m] = sum_i A[b,h][i][j] W_o[m][h*c+i]: (W_o blockdiag(A))^T, the K-major operand of y = Mf V
        if not be.attn_core_fwd(u, self.temp, self.pk_o[0], sq, Gn, A, MfT):     # small images: the whole chain in 1-2 launches
            be.row_sumsq(u[:, :2 * C], sq)
            Graw = be.bmm_nt_slabs(Q, K)              # q k^T as split-K slabs: the softmax kernel sums them (no reduce launch)
            if Graw is None:
                Graw = be.empty(B, hd, c, c)
                be.bmm_nt(Q, K, Graw)
            be.attn_softmax(Graw, sq, self.temp, Gn, A)
            be.bmm_nn(A, self._woT_heads(B), MfT.view(B, hd, c, C), transA=True)
        y = be.empty(B, C, H, W)
        mu2, rs2 = be.empty(B, N), be.empty(B, N)
        if fast and pstats:
            be.gemm_kmajor_stats(MfT.unsqueeze(1), V, y.view(B, 1, C, N), C, C, x.view(B, 1, C, N), (mu2, rs2))
        elif fast:
            be.gemm_kmajor(MfT.unsqueeze(1), V, y.view(B, 1, C, N), C, C, R=x.view(B, 1, C, N))
        else:
            be.bmm_nn(MfT.unsqueeze(1), V, y.view(B, 1, C, N), transA=True, R=x.view(B, 1, C, N))
        pp = be.empty(B, 2 * hid, H, W)
        be.conv1x1_fwd(self.Win, y, pp, ln=(mu2, rs2, self.w2, self.b2), packed=self.pk_in, ln_compute=not (fast and pstats))
        if wmask is not None:
            pp[..., wmask:].zero_()
        gg = be.empty(B, hid, H, W)
        be.gdfn_gate_fwd(pp, self.Wdw2, gg)
        out = be.empty(B, C, H, W)
        if want_stats and pstats and fast:
            self.out_stats = (be.empty(B, N), be.empty(B, N))
            be.conv1x1_fwd(self.Wout, gg, out, R=y, packed=self.pk_out, stats=self.out_stats)
        else:
            be.conv1x1_fwd(self.Wout, gg, out, R=y, packed=self.pk_out)
        ctx = (x, mu1, rs1, t, u, sq, Gn, A, y, mu2, rs2, pp, gg) if save else None
        return out, ctx

    def backward(self, ctx, dout):
        be, C, hd, c, hid = self.be, self.C, self.heads, self.c, self.hid
        x, mu1, rs1, t, u, sq, Gn, A, y, mu2, rs2, pp, gg = ctx
        B, _, H, W = x.shape
        N = H * W
        fast = be.kmajor_worth(C, N, B)
        # ---- GDFN
        slabs = []       # weight gradients left as split-K slabs on the side stream; block_param_reduce() adds them up
        dg = be.empty(B, hid, H, W)
        self._dgrad_wgrad(slabs, self.Wout, dout, dg, gg, self.gWout, None, self.pk_out, 0)
        dp = be.empty(B, 2 * hid, H, W)
        be.gdfn_bwd(pp, self.Wdw2, dg, dp, self.gWdw2)    # gate backward, rotated depthwise conv and its weight gradient: one pass
        del dg
        gln = be.empty(B, C, H, W)
        self._dgrad_wgrad(slabs, self.Win, dp, gln, y, self.gWin, (mu2, rs2, self.w2, self.b2), self.pk_in, 1)
        del dp
        dy = be.empty(B, C, H, W)
        be.ln_bwd(gln, y, mu2, rs2, self.w2, dout, dy, None, None, slot=0)     # norm2: dw/db partials deferred (slot 0)
        # ---- MDTA
        Q, K, V = self._qkv_views(u)
        dy4 = dy.view(B, 1, C, N)
        dM, Mf = be.empty(B, C, C), be.empty(B, C, C)
        # (rcot_attn_core_bwd also takes dM as the <= 8 split-K slabs of rcot_bmm_nt_slabs and sums them while staging; wired in for the
        # 32x32 / 16x16 planes in round 5 it removed a reduce launch and NO time — 340.0 vs 341.0 us per 16x16 block backward,
        # profiles/r05_ab_small_levels.txt — while the split factor depends on the batch (16 at B = 2): not kept in the schedule)
        be.bmm_nt(dy4, V, dM.unsqueeze(1))                           # dM = dY V^T
        du = be.empty(B, 3 * C, H, W)
        dQ, dK, dV = self._qkv_views(du)
        dWo_part, dtemp_part = be.empty(B, C, C), be.empty(B, hd)
        Eq, EqT = be.empty(B, hd, c, c), be.empty(B, hd, c, c)
        Dq, Dk = be.empty(B, C), be.empty(B, C)
        if be.attn_core_bwd(dM, self.Wo, A, Gn, sq, self.temp, Mf, dWo_part, dtemp_part, Eq, EqT, Dq, Dk):
            pass        # Mf = W_o blockdiag(A), dW_o (per image), dA = W_o^T dM and the softmax backward: ONE launch, fp32 MFMA
        elif be.attn_fused_ok(c):
            # the same as two launches (row chunks, then the c x c tail)
            be.attn_bwd_fused(dM, self.Wo, A, Gn, sq, self.temp, Mf, dWo_part, dtemp_part, Eq, EqT, Dq, Dk)
        else:
            be.bmm_nn(self._wo_heads(B), A, self._head_cols(Mf))     # Mf = W_o blockdiag(A): K-major operand of dV = Mf^T dY
            dA = be.empty(B, hd, c, c)
            dMh = self._head_cols(dM)
            be.bmm_nn(self._wo_heads(B), dMh, dA, transA=True)      # dA[b,h] = W_o[:,h]^T dM[b][:,h]
            be.bmm_nt(dMh, A, self._head_cols(dWo_part))            # dW_o[:,h] (per image) = dM[b][:,h] A[b,h]^T
            be.attn_bwd_small(dA, A, Gn, sq, self.temp, dtemp_part, Eq, EqT, Dq, Dk)
        multi = getattr(be, "gemm_kmajor_multi", None)
        if fast and c % 16 == 0 and multi is not None and multi([
                (Mf.unsqueeze(1), dy4, dV, C, C, None, None),                      # dV = Mf^T dY
                (EqT, K, dQ, c, c, Q, Dq.view(B, hd, c)),                          # dQ = Eq K + Dq.Q
                (Eq, Q, dK, c, c, K, Dk.view(B, hd, c))]):                         # dK = Eq^T Q + Dk.K
            pass        # the three data gradients of the attention apply are independent: one launch (exact-fp32 kernel)
        elif fast and c % 16 == 0:
            be.gemm_kmajor(Mf.unsqueeze(1), dy4, dV, C, C)
            be.gemm_kmajor(EqT, K, dQ, c, c, R=Q, rowscale=Dq.view(B, hd, c))
            be.gemm_kmajor(Eq, Q, dK, c, c, R=K, rowscale=Dk.view(B, hd, c))
        elif fast:
            be.gemm_kmajor(Mf.unsqueeze(1), dy4, dV, C, C)
            be.bmm_nn(Eq, K, dQ, R=Q, rowscale=Dq.view(B, hd, c))
            be.bmm_nn(Eq, Q, dK, transA=True, R=K, rowscale=Dk.view(B, hd, c))
        else:
            be.bmm_nn(Mf.unsqueeze(1), dy4, dV, transA=True)
            be.bmm_nn(Eq, K, dQ, R=Q, rowscale=Dq.view(B, hd, c))
            be.bmm_nn(Eq, Q, dK, transA=True, R=K, rowscale=Dk.view(B, hd, c))
        dt = be.empty(B, 3 * C, H, W)
        be.dwconv3x3_bwd(du, t, self.Wdw, dt, self.gWdw)          # data + weight gradient of the qkv depthwise conv, one pass
        del du
        self._dgrad_wgrad(slabs, self.Wqkv, dt, gln, x, self.gWqkv, (mu1, rs1, self.w1, self.b1), self.pk_qkv, 2)
        dx = be.empty(B, C, H, W)
        be.ln_bwd(gln, x, mu1, rs1, self.w1, dy, dx, None, None, slot=1)       # norm1: deferred (slot 1)
        # one launch closes the block: both LayerNorms' dw/db, dW_o and dtau summed over the batch + the three 1x1 weight gradients
        # from their slabs (no reduce launches of their own).  It goes to the side stream behind the slab kernels; the main
        # stream does not wait for it (T_net.backward joins before the gradients are read)
        be.block_param_reduce(C, self.gw2, self.gb2, self.gw1, self.gb1, dWo_part, self.gWo, dtemp_part, self.gtemp, slabs, close_block=True)
        return dx


class Conv3x3Op:
    """Dense 3x3 conv (pad 1, no bias) with optional PixelUnshuffle (cmap 1) / PixelShuffle (cmap 2)
    folded into the store — Net_Restormer.py:86-94, 103-111, 113-122, 326."""

    def __init__(self, be, store, name, cmap=0):
        self.be, self.W, self.gW, self.cmap, self.name = be, store.p[name], store.g[name], cmap, name
        # bf16x3 arithmetic, channel counts that are multiples of 16 (the Down/Upsample convolutions from level 2 on): forward and
        # data gradient as split-bf16 K-major products over a padded channel-major copy (rcot_conv_pcm_*, ~100 TF/s against
        # ~55 of the implicit-GEMM engine at these shapes); the (un)shuffle then is its own small launch.  The transport map is
        # not the critic: its gradients keep their bars with split products (tests/test_iteration_grads_gpu.py).
        Co, Ci = self.W.shape[0], self.W.shape[1]
        self._pcm = (hasattr(be, "conv_pcm_fwd") and os.environ.get("RCOT_TCONV_PCM", "1") != "0" and Ci % 16 == 0 and Co % 16 == 0
                     and min(Ci, Co) >= int(os.environ.get("RCOT_TCONV_PCM_MINC", "48")))
        self._packs = None
        self._pcm_wgrad = os.environ.get("RCOT_TCONV_PCM_WGRAD", "1") != "0"     # (A/B switch: weight gradient on the engine)

    def repack(self):
        """operand packs of the padded-plane products (after every parameter change; made at first use in bf16x3)"""
        if self._pcm and getattr(self.be, "prec", 0) == 1:
            old = self._packs or (None, None)
            self._packs = (self.be.conv_pcm_pack(self.W, "fwd", old[0]), self.be.conv_pcm_pack(self.W, "dgrad", old[1]))
        else:
            self._packs = None

    def _pcm_packs(self, H, W):
        if not (self._pcm and getattr(self.be, "prec", 0) == 1 and W % 4 == 0):
            return None
        if self._packs is None:
            self.repack()
        return self._packs

    def out_shape(self, x):
        B, _, H, W = x.shape
        Co = self.W.shape[0]
        if self.cmap == 1:
            return (B, 4 * Co, H // 2, W // 2)
        if self.cmap == 2:
            return (B, Co // 4, 2 * H, 2 * W)
        return (B, Co, H, W)

    def forward(self, x, R=None):
        be = self.be
        y = be.empty(*self.out_shape(x))
        pk = self._pcm_packs(x.shape[2], x.shape[3]) if R is None else None
        if pk is not None:
            B, _, H, W = x.shape
            d = y if not self.cmap else be.empty(B, self.W.shape[0], H, W)
            be.conv_pcm_fwd(x, pk[0], None, d, 3, 1.0)
            if self.cmap:
                be.pixel_shuffle(d, y, self.cmap)                   # 1: PixelUnshuffle(2), 2: PixelShuffle(2)
            return y
        be.conv2d_fwd(x, self.W, None, y, 1, 1, 1.0, self.cmap, R)
        return y

    def backward(self, x, dy, need_dx=True, dx_out=None, beta=0.0):
        be = self.be
        B, _, H, W = x.shape
        Co = self.W.shape[0]
        if self.cmap:
            d = be.empty(B, Co, H, W)
            be.pixel_shuffle(dy, d, 2 if self.cmap == 1 else 1)     # inverse permutation
            dy = d
        pkw = self._pcm_packs(H, W)
        prepped = pkw is not None and self._pcm_wgrad and be.conv_pcm_wgrad(dy, x, self.gW, 1.0)
        if not prepped:
            be.conv2d_wgrad(dy, x, self.gW, 1, 1, beta=1.0)
        if not need_dx:
            return None
        dx = dx_out if dx_out is not None else be.empty(*x.shape)
        pk = pkw if beta == 0.0 else None
        if pk is not None:
            be.conv_pcm_dgrad(dy, pk[1], dx, 3, prepped=prepped)
        else:
            be.conv2d_dgrad(dy, self.W, dx, 1, 1, beta=beta)
        return dx


class Conv1x1Op:
    """1x1 conv without bias over one source, or over the channel-concatenation of two sources
    (torch.cat + reduce_chan_level*, Net_Restormer.py:351-352) without materialising the cat."""

    def __init__(self, be, store, name):
        self.be = be
        w = store.p[name]
        self.W, self.gW = w.view(w.shape[0], w.shape[1]), store.g[name].view(w.shape[0], w.shape[1])
        self._pk = {}

    def _pack(self, lo, hi):
        key = (lo, hi)
        if key not in self._pk:
            W = self.W[:, lo:hi]
            st, sp = self.be.split_shapes(*W.shape)
            split6 = None
            if getattr(self.be, "x6_packs", False):
                st6, sp6 = self.be.split6_shapes(*W.shape)
                split6 = (self.be.zeros(*st6), self.be.zeros(*sp6), None)
            pk = tuple(self.be.zeros(*s) for s in self.be.pack_shapes(*W.shape)) + (None, (self.be.zeros(*st), self.be.zeros(*sp), None), split6)
            self.be.pack_weight(W, pk[0], pk[1], None, pk[3], pk[4])
            self._pk[key] = pk
        return self._pk[key]

    def pack_items(self):
        return [(self.W[:, lo:hi], pk[0], pk[1], None, pk[3], pk[4]) for (lo, hi), pk in self._pk.items()]

    def repack(self):
        for W, WT, WP, _, split, split6 in self.pack_items():
            self.be.pack_weight(W, WT, WP, None, split, split6)

    def forward(self, x1, x2=None):
        be = self.be
        B, C1, H, W = x1.shape
        y = be.empty(B, self.W.shape[0], H, W)
        be.conv1x1_fwd(self.W[:, :C1], x1, y, packed=self._pack(0, C1))
        if x2 is not None:
            be.conv1x1_fwd(self.W[:, C1:], x2, y, beta=1.0, packed=self._pack(C1, self.W.shape[1]))
        return y

    def backward(self, x1, dy, x2=None, dx2_out=None, beta2=0.0):
        """returns dx1; writes dx2 into dx2_out (beta2 = 1 accumulates into a skip gradient)."""
        be = self.be
        C1 = x1.shape[1]
        be.conv1x1_wgrad(dy, x1, self.gW[:, :C1], beta=1.0)
        dx1 = be.empty(*x1.shape)
        be.conv1x1_dgrad(self.W[:, :C1], dy, dx1, packed=self._pack(0, C1))
        if x2 is not None:
            be.conv1x1_wgrad(dy, x2, self.gW[:, C1:], beta=1.0)
            be.conv1x1_dgrad(self.W[:, C1:], dy, dx2_out, beta=beta2, packed=self._pack(C1, self.W.shape[1]))
        return dx1


def _stage_fwd(blocks, x, save, wmask=None):
    ctxs = []
    st = None                                    # norm1 statistics of x, when the previous block's last product made them (round 6)
    for i, b in enumerate(blocks):
        if wmask is not None:
            x, c = b.forward(x, save, wmask)
        else:
            x, c = b.forward(x, save, in_stats=st, want_stats=i + 1 < len(blocks))
            st = b.out_stats
        ctxs.append(c)
    return x, ctxs


def _stage_bwd(blocks, ctxs, dy):
    for b, c in zip(reversed(blocks), reversed(ctxs)):
        dy = b.backward(c, dy)
    return dy


# =============================================================================== T_net
def _tnet_live_order():
    """Parameter names in the order their gradients become final during T_net.backward."""
    nb, d = P.NUM_BLOCKS, P.DIM
    blk = lambda pre, dim, h: [k for k, _ in P.block_param_shapes(pre, dim, h)]

    def stage_rev(pre, n, dim, h):
        out = []
        for i in reversed(range(n)):
            out += blk(f"{pre}.{i}", dim, h)
        return out
    o: List[str] = []
    o += stage_rev("reslatent", nb[3], 8 * d, P.HEADS[3])
    o += stage_rev("resencoder_level3", nb[2], 4 * d, P.HEADS[2]) + ["resdown2_3.body.0.weight"]
    o += stage_rev("resencoder_level2", nb[1], 2 * d, P.HEADS[1]) + ["resdown1_2.body.0.weight"]
    o += stage_rev("resencoder_level1", nb[0], d, P.HEADS[0])
    o += ["output.weight"] + stage_rev("refinement", P.NUM_REFINEMENT, 2 * d, P.HEADS[0])
    o += stage_rev("decoder_level1", nb[0], 2 * d, P.HEADS[0]) + ["up2_1.body.0.weight", "reduce_noise_level1.weight"]
    o += blk("noise_level1", 2 * d, P.HEADS[2]) + stage_rev("decoder_level2", nb[1], 2 * d, P.HEADS[1])
    o += ["reduce_chan_level2.weight", "up3_2.body.0.weight", "reduce_noise_level2.weight"]
    o += blk("noise_level2", 4 * d, P.HEADS[2]) + stage_rev("decoder_level3", nb[2], 4 * d, P.HEADS[2])
    o += ["reduce_chan_level3.weight", "up4_3.body.0.weight", "reduce_noise_level3.weight"]
    o += blk("noise_level3", 8 * d, P.HEADS[2])
    o += stage_rev("latent", nb[3], 8 * d, P.HEADS[3]) + ["down3_4.body.0.weight"]
    o += stage_rev("encoder_level3", nb[2], 4 * d, P.HEADS[2]) + ["down2_3.body.0.weight"]
    o += stage_rev("encoder_level2", nb[1], 2 * d, P.HEADS[1]) + ["down1_2.body.0.weight"]
    o += stage_rev("encoder_level1", nb[0], d, P.HEADS[0]) + ["patch_embed.proj.weight"]
    return o


class T_net:
    """Two-pass Restormer transport map.  Same constructor signature / defaults as the reference
    (Net_Restormer.py:216-227); only the reference's default architecture is implemented."""

    def __init__(self, inp_channels=3, out_channels=3, dim=48, num_blocks=(4, 6, 6, 8), num_refinement_blocks=4,
                 heads=(1, 2, 4, 8), ffn_expansion_factor=2.66, bias=False, LayerNorm_type="WithBias",
                 decoder=False, backend=None, seed: Optional[int] = None):
        if (inp_channels, out_channels, dim, tuple(num_blocks), num_refinement_blocks, tuple(heads),
                ffn_expansion_factor, bias, LayerNorm_type) != (3, 3, 48, (4, 6, 6, 8), 4, (1, 2, 4, 8), 2.66, False, "WithBias"):
            raise NotImplementedError("only the reference's default T_net architecture is built")
        if backend is None:
            from .ops import default_backend
            backend = default_backend()
        self.be = be = backend
        self.decoder = decoder
        shapes = P.tnet_param_shapes()
        live = _tnet_live_order()
        dead = [n for n, _ in shapes if P.tnet_is_dead(n)]
        assert sorted(live + dead) == sorted(n for n, _ in shapes)
        self.store = st = ParamStore(be, shapes, live, dead)
        st.load(_reference_init(shapes, "T", seed))
        d, nb, h = P.DIM, P.NUM_BLOCKS, P.HEADS
        S = lambda pre, n, dm, hh: [TransformerBlockOp(be, st, f"{pre}.{i}", dm, hh) for i in range(n)]
        self.patch_embed = Conv3x3Op(be, st, "patch_embed.proj.weight")
        self.enc1, self.enc2, self.enc3 = S("encoder_level1", nb[0], d, h[0]), S("encoder_level2", nb[1], 2 * d, h[1]), S("encoder_level3", nb[2], 4 * d, h[2])
        self.res1, self.res2, self.res3 = S("resencoder_level1", nb[0], d, h[0]), S("resencoder_level2", nb[1], 2 * d, h[1]), S("resencoder_level3", nb[2], 4 * d, h[2])
        self.down1_2, self.down2_3, self.down3_4 = (Conv3x3Op(be, st, f"down{a}.body.0.weight", 1) for a in ("1_2", "2_3", "3_4"))
        self.resdown1_2, self.resdown2_3 = (Conv3x3Op(be, st, f"resdown{a}.body.0.weight", 1) for a in ("1_2", "2_3"))
        self.latent, self.reslatent = S("latent", nb[3], 8 * d, h[3]), S("reslatent", nb[3], 8 * d, h[3])
        self.noise3 = TransformerBlockOp(be, st, "noise_level3", 8 * d, h[2])
        self.noise2 = TransformerBlockOp(be, st, "noise_level2", 4 * d, h[2])
        self.noise1 = TransformerBlockOp(be, st, "noise_level1", 2 * d, h[2])
        self.rn3, self.rn2, self.rn1 = (Conv1x1Op(be, st, f"reduce_noise_level{i}.weight") for i in (3, 2, 1))
        self.up4_3, self.up3_2, self.up2_1 = (Conv3x3Op(be, st, f"up{a}.body.0.weight", 2) for a in ("4_3", "3_2", "2_1"))
        self.rc3, self.rc2 = Conv1x1Op(be, st, "reduce_chan_level3.weight"), Conv1x1Op(be, st, "reduce_chan_level2.weight")
        self.dec3, self.dec2, self.dec1 = S("decoder_level3", nb[2], 4 * d, h[2]), S("decoder_level2", nb[1], 2 * d, h[1]), S("decoder_level1", nb[0], 2 * d, h[0])
        self.refine = S("refinement", P.NUM_REFINEMENT, 2 * d, h[0])
        self.output = Conv3x3Op(be, st, "output.weight")
        self._ctx = None
        self.last_res = None
        self._pack_tabs = {}                     # (item count, arithmetic) -> device descriptor table (never freed: recorded plans point at it)
        self._packed_prec = None
        self.repack()
        #: called as hook(n_final) during backward when grad[0:n_final) of the flat buffer is final
        self.grad_ready_hook: Optional[Callable[[int], None]] = None

    def repack(self):
        """Refresh the private K-major copies of every 1x1 weight; must follow any change of the parameters
        (optimizer step, load_state_dict).  One launch for the whole network (descriptor table on the device)."""
        items = []
        for stage in (self.enc1, self.enc2, self.enc3, self.res1, self.res2, self.res3, self.latent, self.reslatent,
                      self.dec3, self.dec2, self.dec1, self.refine, [self.noise3, self.noise2, self.noise1],
                      [self.rn3, self.rn2, self.rn1, self.rc3, self.rc2]):
            for op in stage:
                items.extend(op.pack_items())
        prec = getattr(self.be, "prec", 0)
        # One table per arithmetic (only the fragment packs that arithmetic reads), kept for the life of the network: the launch
        # plans recorded under an arithmetic have the raw pointers of ITS table in their rcot_pack_weights call, so a table must
        # survive a switch to another arithmetic and back (round 4 kept one table and freed it on every switch).
        ent = self._pack_tabs.get((len(items), prec))
        if ent is None:
            if getattr(self.be, "_plan", None) is not None:
                raise RuntimeError("weight-pack descriptor table created while a launch plan is being recorded "
                                   "(the warm-up pass did not cover this arithmetic)")
            tab, total = self.be.pack_table(items, prec)
            ent = self._pack_tabs[(len(items), prec)] = (tab, total, items)       # items keep the views alive
        self._packed_prec = prec
        self.be.pack_weights(ent[0], ent[1])
        for cv in (self.down1_2, self.down2_3, self.down3_4, self.resdown1_2, self.resdown2_3, self.up4_3, self.up3_2, self.up2_1):
            if cv._packs is not None:
                cv.repack()

    # ---- reference-compatible conveniences
    def state_dict(self):
        return self.store.state_dict()

    def load_state_dict(self, sd, strict=True):
        self.store.load(sd, strict)
        if hasattr(self, "repack"):
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

    def __call__(self, inp_img, noise_emb=None):
        return self.forward(inp_img, save=False)

    # ---- 1/8-resolution planes whose pixel count is not a multiple of 4 (whole images with H/8 and W/8 both odd, e.g. 200 x 200:
    # the reference accepts them, trainer.py:195-198): width-padded planes with masked padding columns (TransformerBlockOp.forward)
    def _lat_pad(self, t):
        B, C, H, W = t.shape
        if (H * W) % 4 == 0:
            return t, None
        Wp = (W + 3) // 4 * 4
        tp = self.be.zeros(B, C, H, Wp)
        tp[..., :W].copy_(t)
        return tp, W

    @staticmethod
    def _lat_unpad(tp, W):
        return tp if W is None else tp[..., :W].contiguous()

    # ---- decoder half (shared by both passes)
    def _decode(self, latent, e3, e2, e1, inp, save, wmask=None):
        """``latent`` arrives width-padded when ``wmask`` is given (see _lat_pad)"""
        be = self.be
        c = {}
        if self.decoder:
            n3, c["n3"] = self.noise3.forward(latent, save, wmask) if wmask is not None else self.noise3.forward(latent, save)
            z = self._lat_unpad(self.rn3.forward(n3), wmask)
            c["n3o"] = n3
        else:
            z = self._lat_unpad(latent, wmask)
        u3 = self.up4_3.forward(z)
        d3i = self.rc3.forward(u3, e3)
        d3, c["d3"] = _stage_fwd(self.dec3, d3i, save)
        if self.decoder:
            n2, c["n2"] = self.noise2.forward(d3, save)
            z2 = self.rn2.forward(n2)
            c["n2o"] = n2
        else:
            z2 = d3
        u2 = self.up3_2.forward(z2)
        d2i = self.rc2.forward(u2, e2)
        d2, c["d2"] = _stage_fwd(self.dec2, d2i, save)
        if self.decoder:
            n1, c["n1"] = self.noise1.forward(d2, save)
            z1 = self.rn1.forward(n1)
            c["n1o"] = n1
        else:
            z1 = d2
        u1 = self.up2_1.forward(z1)
        B, C1, H, W = u1.shape
        cat1 = be.empty(B, 2 * C1, H, W)                       # torch.cat([u1, e1], 1), :369
        be.axpby(u1, None, cat1[:, :C1], 1.0, 0.0)
        be.axpby(e1, None, cat1[:, C1:], 1.0, 0.0)
        d1, c["d1"] = _stage_fwd(self.dec1, cat1, save)
        d1, c["rf"] = _stage_fwd(self.refine, d1, save)
        out = self.output.forward(d1, R=inp)                   # conv + inp_img, :375 / :432
        if save:
            c.update(z=z, u3=u3, z2=z2, u2=u2, z1=z1, d1t=d1)
        return out, (c if save else None)

    def _decode_bwd(self, c, dout, e3, e2, e1, de, first):
        """Backward of _decode.  Accumulates skip gradients into de['e1'|'e2'|'e3'] (overwrites when
        ``first``) and returns d(latent)."""
        be = self.be
        b2 = 0.0 if first else 1.0
        dd1 = self.output.backward(c["d1t"], dout)
        dd1 = _stage_bwd(self.refine, c["rf"], dd1)
        dcat = _stage_bwd(self.dec1, c["d1"], dd1)
        C1 = dcat.shape[1] // 2
        du1 = be.empty(dcat.shape[0], C1, dcat.shape[2], dcat.shape[3])
        be.axpby(dcat[:, :C1], None, du1, 1.0, 0.0)
        if first:
            be.axpby(dcat[:, C1:], None, de["e1"], 1.0, 0.0)
        else:
            be.axpby(dcat[:, C1:], de["e1"], de["e1"], 1.0, 1.0)
        dz1 = self.up2_1.backward(c["z1"], du1)
        if self.decoder:
            dn1 = self.rn1.backward(c["n1o"], dz1)
            dd2 = self.noise1.backward(c["n1"], dn1)
        else:
            dd2 = dz1
        dd2i = _stage_bwd(self.dec2, c["d2"], dd2)
        du2 = self.rc2.backward(c["u2"], dd2i, e2, de["e2"], b2)
        dz2 = self.up3_2.backward(c["z2"], du2)
        if self.decoder:
            dn2 = self.rn2.backward(c["n2o"], dz2)
            dd3 = self.noise2.backward(c["n2"], dn2)
        else:
            dd3 = dz2
        dd3i = _stage_bwd(self.dec3, c["d3"], dd3)
        du3 = self.rc3.backward(c["u3"], dd3i, e3, de["e3"], b2)
        dz = self.up4_3.backward(c["z"], du3)
        if self.decoder:
            dn3 = self.rn3.backward(c["n3o"], dz)
            return self.noise3.backward(c["n3"], dn3)
        return dz

    # ---- forward / backward
    def forward(self, inp, save: bool = False):
        """inp: [B,3,H,W] fp32 on the device, H and W multiples of 8.  Returns the restored image.
        With ``save`` the activations needed by ``backward`` are kept."""
        be = self.be
        if getattr(be, "prec", 0) != self._packed_prec:          # the arithmetic changed since the last repack: its fragment packs are stale
            self.repack()
        inp = inp.contiguous()
        pe = self.patch_embed.forward(inp)
        e1, c_e1 = _stage_fwd(self.enc1, pe, save)
        x2 = self.down1_2.forward(e1)
        e2, c_e2 = _stage_fwd(self.enc2, x2, save)
        x3 = self.down2_3.forward(e2)
        e3, c_e3 = _stage_fwd(self.enc3, x3, save)
        l4, wmask = self._lat_pad(self.down3_4.forward(e3))
        assert wmask is None or not save, "training patches need (H/8)*(W/8) % 4 == 0"
        lat, c_lat = _stage_fwd(self.latent, l4, save, wmask)
        out1, c_dec1 = self._decode(lat, e3, e2, e1, inp, save, wmask)
        res = be.empty(*inp.shape)
        be.axpby(inp, out1, res, 1.0, -1.0)                     # :377
        self.last_res = res
        rpe = self.patch_embed.forward(res)                     # patch_embed is reused, :381
        r1, c_r1 = _stage_fwd(self.res1, rpe, save)
        rx2 = self.resdown1_2.forward(r1)
        r2, c_r2 = _stage_fwd(self.res2, rx2, save)
        rx3 = self.resdown2_3.forward(r2)
        r3, c_r3 = _stage_fwd(self.res3, rx3, save)
        rl4, _ = self._lat_pad(self.down3_4.forward(r3))       # down3_4 is reused, :393
        r4, c_r4 = _stage_fwd(self.reslatent, rl4, save, wmask)
        if self.decoder:
            lat2 = be.empty(*lat.shape)
            be.axpby(lat, r4, lat2, 1.0, 0.8)                   # latent += 0.8*reslatent, :401
        else:
            lat2 = lat
        out2, c_dec2 = self._decode(lat2, e3, e2, e1, inp, save, wmask)
        if save:
            self._ctx = dict(inp=inp, pe=pe, e1=e1, e2=e2, e3=e3, x2=x2, x3=x3, l4=l4, c_e1=c_e1, c_e2=c_e2,
                             c_e3=c_e3, c_lat=c_lat, c_dec1=c_dec1, c_dec2=c_dec2, res=res, rpe=rpe, r1=r1, r2=r2,
                             r3=r3, rx2=rx2, rx3=rx3, rl4=rl4, c_r1=c_r1, c_r2=c_r2, c_r3=c_r3, c_r4=c_r4)
        return out2

    def _ready(self, after_param: str):
        if self.grad_ready_hook is not None:
            self.be.side_join()                  # deferred block closes write parameter gradients on the side stream
            lay = self.store.layout
            i = lay.order.index(after_param)
            nxt = lay.order[i + 1] if i + 1 < len(lay.order) else None
            end = lay.offset[nxt] if (nxt is not None and not P.tnet_is_dead(nxt)) else lay.n_live
            self.grad_ready_hook(end)

    def backward(self, dout):
        """Accumulates d(loss)/d(parameters) into the flat gradient buffer given d(loss)/d(output)."""
        be, k = self.be, self._ctx
        assert k is not None, "forward(save=True) must precede backward"
        e1, e2, e3 = k["e1"], k["e2"], k["e3"]
        de = dict(e1=be.empty(*e1.shape), e2=be.empty(*e2.shape), e3=be.empty(*e3.shape))
        dlat = self._decode_bwd(k["c_dec2"], dout.contiguous(), e3, e2, e1, de, first=True)
        if self.decoder:
            dr4 = be.empty(*dlat.shape)
            be.axpby(dlat, None, dr4, 0.8, 0.0)
            drl4 = _stage_bwd(self.reslatent, k["c_r4"], dr4)
            dr3 = self.down3_4.backward(k["r3"], drl4)
            drx3 = _stage_bwd(self.res3, k["c_r3"], dr3)
            dr2 = self.resdown2_3.backward(k["r2"], drx3)
            drx2 = _stage_bwd(self.res2, k["c_r2"], dr2)
            dr1 = self.resdown1_2.backward(k["r1"], drx2)
            drpe = _stage_bwd(self.res1, k["c_r1"], dr1)
            self._ready("resencoder_level1.0.ffn.project_out.weight")
            dres = self.patch_embed.backward(k["res"], drpe)
            dout1 = be.empty(*dres.shape)
            be.axpby(dres, None, dout1, -1.0, 0.0)                 # res = inp - out1
            dlat1 = self._decode_bwd(k["c_dec1"], dout1, e3, e2, e1, de, first=False)
            be.axpby(dlat, dlat1, dlat, 1.0, 1.0)
        self._ready("noise_level3.ffn.project_out.weight")
        dl4 = _stage_bwd(self.latent, k["c_lat"], dlat)
        self.down3_4.backward(e3, dl4, dx_out=de["e3"], beta=1.0)
        self._ready("down3_4.body.0.weight")
        dx3 = _stage_bwd(self.enc3, k["c_e3"], de["e3"])
        self.down2_3.backward(e2, dx3, dx_out=de["e2"], beta=1.0)
        self._ready("down2_3.body.0.weight")
        dx2 = _stage_bwd(self.enc2, k["c_e2"], de["e2"])
        self.down1_2.backward(e1, dx2, dx_out=de["e1"], beta=1.0)
        self._ready("down1_2.body.0.weight")
        dpe = _stage_bwd(self.enc1, k["c_e1"], de["e1"])
        self.patch_embed.backward(k["inp"], dpe, need_dx=False)
        be.side_join()                           # every parameter gradient is final from here on
        self._ready("patch_embed.proj.weight")
        self._ctx = None


# =============================================================================== F_net
def _fnet_live_order(patch_size):
    names = [n for n, _ in P.fnet_param_shapes(patch_size)]
    feats = [n for n in names if n.startswith("features.")]
    # reverse layer order (gradient-final order); fc2.bias last: the gradient-penalty step never
    # produces a gradient for it (autograd gives None), so that step's optimizer range stops before it.
    by_layer = sorted(feats, key=lambda s: (-int(s.split(".")[1]), s.endswith("bias")))
    return ["fc2.weight", "fc1.weight", "fc1.bias", "fc.weight", "fc.bias"] + by_layer + ["fc2.bias"]


class F_net:
    """WGAN-GP critic / OT potential: 10x [conv + LeakyReLU(0.2)] + 3 Linear (Net_Restormer.py:436-522).
    Activations are stored post-LeakyReLU; the ReLU mask is recovered from their sign."""

    def __init__(self, patch_size=64, backend=None, seed: Optional[int] = None):
        if backend is None:
            from .ops import default_backend
            backend = default_backend()
        self.be = be = backend
        self.patch_size = patch_size
        shapes = P.fnet_param_shapes(patch_size)
        self.store = st = ParamStore(be, shapes, _fnet_live_order(patch_size), [])
        st.load(_reference_init(shapes, "F", seed))
        self.convs = []
        for i, (cin, cout, k, s, pad, bias) in enumerate(P.FNET_CONVS):
            wn, bn = f"features.{2 * i}.weight", f"features.{2 * i}.bias"
            self.convs.append(dict(W=st.p[wn], gW=st.g[wn], b=st.p.get(bn) if bias else None,
                                   gb=st.g.get(bn) if bias else None, s=s, pad=pad, cout=cout))
        self.n_live_gp = st.layout.offset["fc2.bias"]          # optimizer range of the GP step
        self._ctx = None
        # RCOT_CONV_PCM=1 (opt-in, bf16x3 arithmetic only): the nine inner convolutions (Ci >= 64) and their data gradients as
        # split-bf16 K-major products over padded channel-major operands (rcot_conv_pcm_*, csrc/conv_pcm.hip), ~2x the
        # implicit-GEMM engine on those layers (-2.5 ms/step at B=8, 128x128).  NOT the default: each product is 5e-6 accurate,
        # but the critic's LeakyReLU masks, the GP double backward and RMSprop's sign-like first steps amplify that past the
        # gradient-parity bars of tests/test_iteration_grads_gpu.py (1 - cos of the GP gradients 5e-4..8e-4 vs 4e-4; DESIGN.md
        # section 6), so the critic's convolutions stay exact fp32 in both arithmetics unless asked otherwise.
        self._pcm = hasattr(be, "conv_pcm_fwd") and os.environ.get("RCOT_CONV_PCM", "0") == "1"
        self._packs = {}
        self._stale = True
        self._mask_fold = os.environ.get("RCOT_MASK_FOLD", "1") != "0"      # (A/B switch: separate rcot_lrelu_bwd launches)
        self._side_leaves = hasattr(be, "side_run") and os.environ.get("RCOT_F_SIDE", "1") != "0"
        #: called as hook(n_final) during backward(wgrad=True) when grad[0:n_final) of the flat buffer is final (the layout
        #: follows the critic-loss backward: fc2, fc1, fc, then the convolutions last to first)
        self.grad_ready_hook: Optional[Callable[[int], None]] = None
        #: called as hook(n_from) during gradient_penalty_backward() when grad[n_from:n_live) is final: that sweep produces the
        #: weight gradients first layer to last, i.e. from the END of the same layout
        self.grad_tail_hook: Optional[Callable[[int], None]] = None

    def _leaf(self, fn, *hold):
        """a parameter-gradient product of the backward sweeps (nothing in the sweep reads its result): next to the data-gradient
        chain on the backend's side stream (RCOT_F_SIDE=0: in line)"""
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

    def _ready_tail(self, from_param: str):
        if self.grad_tail_hook is not None:
            self.be.side_join()
            self.grad_tail_hook(self.store.layout.offset[from_param])

    def _pcm_layer(self, li, H, W):
        """the (forward, data-gradient) pack pair of conv ``li`` for an H x W input, or None (the layer runs on rcot_conv2d_*)"""
        if not (self._pcm and getattr(self.be, "prec", 0) == 1):
            return None
        if self._stale:
            self.repack()
        cv = self.convs[li]
        Co, Ci, k, _ = cv["W"].shape
        if li in self._packs and self.be.conv_pcm_ok(Ci, Co, k, cv["s"], cv["pad"], H, W):
            return self._packs[li]
        return None

    def repack(self):
        """pre-split packs of the inner conv weights in forward and data-gradient operand order (after every parameter change;
        deferred to the next use while the exact-fp32 arithmetic is selected)"""
        if not self._pcm:
            return
        if getattr(self.be, "prec", 0) != 1:
            self._stale = True
            return
        self._stale = False
        for li, cv in enumerate(self.convs):
            Wt = cv["W"]
            k = Wt.shape[2]
            if Wt.shape[1] % 16 or Wt.shape[0] % 16 or (k, cv["s"], cv["pad"]) not in ((3, 1, 1), (4, 2, 1)):
                continue
            old = self._packs.get(li, (None, None))
            self._packs[li] = (self.be.conv_pcm_pack(Wt, "fwd", old[0]), self.be.conv_pcm_pack(Wt, "dgrad", old[1]))

    state_dict = T_net.state_dict
    load_state_dict = T_net.load_state_dict
    zero_grad = T_net.zero_grad
    parameters = T_net.parameters
    named_parameters = T_net.named_parameters
    cuda = T_net.cuda
    train = T_net.train
    eval = T_net.eval

    def __call__(self, x):
        return self.forward(x, save=False)

    def forward(self, x, save: bool = False):
        be, p = self.be, self.store.p
        B = x.shape[0]
        acts = [x.contiguous()]
        a = acts[0]
        for cv in self.convs:
            H, W = a.shape[2], a.shape[3]
            k = cv["W"].shape[2]
            OH, OW = (H + 2 * cv["pad"] - k) // cv["s"] + 1, (W + 2 * cv["pad"] - k) // cv["s"] + 1
            y = be.empty(B, cv["cout"], OH, OW)
            pk = self._pcm_layer(len(acts) - 1, H, W)
            if pk is not None:
                be.conv_pcm_fwd(a, pk[0], cv["b"], y, k, 0.2)
            else:
                be.conv2d_fwd(a, cv["W"], cv["b"], y, cv["s"], cv["pad"], 0.2, 0, None)
            acts.append(y)
            a = y
        flat = a.view(B, -1)
        f1 = be.empty(B, p["fc.weight"].shape[0])
        be.linear_fwd(flat, p["fc.weight"], p["fc.bias"], f1)
        f2 = be.empty(B, 64)
        be.linear_fwd(f1, p["fc1.weight"], p["fc1.bias"], f2, lrelu=0.2)
        out = be.empty(B, 1)
        be.linear_fwd(f2, p["fc2.weight"], p["fc2.bias"], out)
        if save:
            self._ctx = (acts, f1, f2)
        return out.view(B)

    def backward(self, dout, wgrad: bool = True, need_dx: bool = False, keep_vz: bool = False):
        """dout: [B] = d(loss)/d(F(x)).  Accumulates parameter gradients (``wgrad``) and returns
        d(loss)/dx when ``need_dx``.  ``keep_vz`` keeps the masked per-layer gradients (sweep v of the
        gradient penalty) and returns them too."""
        be, p, g = self.be, self.store.p, self.store.g
        acts, f1, f2 = self._ctx
        B = dout.shape[0]
        d3 = dout.contiguous().view(B, 1)
        if wgrad:
            self._leaf(lambda: (be.linear_wgrad(d3, f2, g["fc2.weight"], 1.0), be.bias_grad(d3, g["fc2.bias"])), d3, f2)
        df2 = be.empty(B, 64)
        be.linear_dgrad(d3, p["fc2.weight"], df2)
        vz2 = be.empty(B, 64)
        be.lrelu_bwd(df2, f2, vz2)
        flat = acts[-1].view(B, -1)
        if wgrad:
            self._leaf(lambda: (be.linear_wgrad(vz2, f1, g["fc1.weight"], 1.0), be.bias_grad(vz2, g["fc1.bias"])), vz2, f1)
        v1 = be.empty(*f1.shape)
        be.linear_dgrad(vz2, p["fc1.weight"], v1)
        if wgrad:
            self._leaf(lambda: (be.linear_wgrad(v1, flat, g["fc.weight"], 1.0), be.bias_grad(v1, g["fc.bias"])), v1, flat)
            self._ready("fc.bias")
        da = be.empty(*acts[-1].shape)
        be.linear_dgrad(v1, p["fc.weight"], da.view(B, -1))
        vzs = [None] * len(self.convs)
        masked = False      # da already carries the LeakyReLU mask of acts[li + 1] (folded into the store of the data gradient)
        for li in reversed(range(len(self.convs))):
            cv = self.convs[li]
            if masked:
                dz = da
            else:
                dz = be.empty(*acts[li + 1].shape)
                be.lrelu_bwd(da, acts[li + 1], dz)
            if keep_vz:
                vzs[li] = dz
            if wgrad:
                def leaf(dz=dz, x=acts[li], cv=cv):
                    be.conv2d_wgrad(dz, x, cv["gW"], cv["s"], cv["pad"], 1.0)
                    if cv["gb"] is not None:
                        be.bias_grad(dz, cv["gb"])
                self._leaf(leaf, dz, acts[li])
                self._ready(f"features.{2 * li}.bias" if cv["gb"] is not None else f"features.{2 * li}.weight")
            if li > 0 or need_dx:
                da = be.empty(*acts[li].shape)
                pk = self._pcm_layer(li, acts[li].shape[2], acts[li].shape[3])
                masked = False
                if pk is not None:
                    be.conv_pcm_dgrad(dz, pk[1], da, cv["W"].shape[2])
                elif li > 0 and self._mask_fold:
                    # the next step multiplies by the LeakyReLU mask of acts[li]: the same expression, in this launch's store
                    be.conv2d_dgrad(dz, cv["W"], da, cv["s"], cv["pad"], 0.0, mask=acts[li], mslope=0.2)
                    masked = True
                else:
                    be.conv2d_dgrad(dz, cv["W"], da, cv["s"], cv["pad"], 0.0)
            else:
                da = None
        if wgrad:
            be.side_join()                                           # every parameter gradient is final from here on
        if keep_vz:
            return da, (vzs, vz2, v1)
        return da

    def gp_input_gradient(self, gout):
        """First half of the double backward (after ``forward(x, save=True)``): dx = d<gout, F(x)>/dx and the linearisation
        context (saved activations and the masked per-layer gradients of sweep v) that gp_param_gradients()
        needs.  No parameter gradient is touched."""
        acts, f1, f2 = self._ctx
        gx, (vzs, vz2, v1) = self.backward(gout, wgrad=False, need_dx=True, keep_vz=True)
        return gx, (acts, f1, f2, vzs, vz2, v1, gout)

    def gp_param_gradients(self, u, lin):
        """Second half: accumulates d<u, dx>/d(parameters) for dx of gp_input_gradient() — ``u`` swept through the net
        linearised at the saved activations.  Bias gradients are exactly zero and fc2.bias receives none, as with autograd."""
        be, p, g = self.be, self.store.p, self.store.g
        acts, f1, f2, vzs, vz2, v1, gout = lin
        B = u.shape[0]
        for li, cv in enumerate(self.convs):                         # sweep u through the linearised net
            self._leaf(lambda v=vzs[li], u=u, cv=cv: be.conv2d_wgrad(v, u, cv["gW"], cv["s"], cv["pad"], 1.0), vzs[li], u)
            self._ready_tail(f"features.{2 * li}.weight")            # this layer's range and everything behind it is final
            y = be.empty(*acts[li + 1].shape)
            pk = self._pcm_layer(li, u.shape[2], u.shape[3])
            if pk is not None:
                be.conv_pcm_fwd(u, pk[0], None, y, cv["W"].shape[2], 1.0)
                be.lrelu_bwd(y, acts[li + 1], y)
            elif self._mask_fold:
                be.conv2d_fwd(u, cv["W"], None, y, cv["s"], cv["pad"], 1.0, 0, None, mask=acts[li + 1], mslope=0.2)
            else:
                be.conv2d_fwd(u, cv["W"], None, y, cv["s"], cv["pad"], 1.0, 0, None)
                be.lrelu_bwd(y, acts[li + 1], y)
            u = y
        uf = u.view(B, -1)
        self._leaf(lambda: be.linear_wgrad(v1, uf, g["fc.weight"], 1.0), v1, u)
        u1 = be.empty(*f1.shape)
        be.linear_fwd(uf, p["fc.weight"], None, u1)
        self._leaf(lambda: be.linear_wgrad(vz2, u1, g["fc1.weight"], 1.0), vz2, u1)
        u2 = be.empty(B, 64)
        be.linear_fwd(u1, p["fc1.weight"], None, u2)
        be.lrelu_bwd(u2, f2, u2)
        be.linear_wgrad(gout.contiguous().view(B, 1), u2, g["fc2.weight"], 1.0)
        be.side_join()

    def gradient_penalty_backward(self, interp, inv_global_batch: float, gp_out):
        """Gradient penalty 10*mean((||dF/dx||-1)^2) and its parameter gradients as explicit sweeps
        (reference: trainer.py:283-307 via double backward)."""
        be = self.be
        B = interp.shape[0]
        self.forward(interp, save=True)
        ones = be.empty(B)
        be.fill(ones, 1.0)
        gx, lin = self.gp_input_gradient(ones)
        norms, u = be.empty(B), be.empty(*gx.shape)
        be.gp_penalty(gx, norms, u, gp_out, inv_global_batch)
        self.gp_param_gradients(u, lin)
        self._ctx = None
