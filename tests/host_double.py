"""TEST DOUBLE — not part of the product, never imported by rcot_amd/.

``TorchDouble`` re-states the semantics of every ``HipBackend`` method (rcot_amd/ops.py) with plain
torch CPU tensor math, so that the CPU-only test tier can exercise the HOST LOGIC of the framework
(the hand-written forward/backward schedules in rcot_amd/net_restormer.py, gradient accumulation,
flat buffers, the minimax step, the data-parallel reducer) against the oracle without a GPU.
It says nothing about the HIP kernels; those are checked by the ``-m gpu`` tests, one entry point
at a time, against the same oracle.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _ln_apply(x, ln):
    if ln is None:
        return x
    mu, rs, w, b = ln
    B, C = x.shape[0], x.shape[1]
    xx = x.reshape(B, C, -1)
    return ((xx - mu[:, None, :]) * rs[:, None, :] * w.view(1, C, 1) + b.view(1, C, 1)).reshape(x.shape)


class TorchDouble:
    name = "torch-double"

    def __init__(self, dtype=torch.float32):
        self.device = torch.device("cpu")
        self.dtype = dtype

    def side_run(self, fn, *hold):
        fn()

    def side_join(self):
        pass

    def empty(self, *shape):
        return torch.full(shape, float("nan"), dtype=self.dtype)      # poison: catches reads of unwritten memory

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=self.dtype)

    # ---- K-major fast path (the packs are USED when given, so stale packs are caught by the CPU tier)
    @staticmethod
    def pack_shapes(Co, Ci):
        r16, r4 = (lambda v: (v + 15) // 16 * 16), (lambda v: (v + 3) // 4 * 4)
        return (r16(Ci), r4(Co)), (r16(Co), r4(Ci))

    @staticmethod
    def fold_shapes(Co, Ci):
        r16, r4 = (lambda v: (v + 15) // 16 * 16), (lambda v: (v + 3) // 4 * 4)
        return (r16(Ci), r4(Co)), (2, r4(Co))

    @staticmethod
    def split_shapes(Co, Ci):
        c = lambda v, q: (v + q - 1) // q
        return (c(Ci, 16) * c(Co, 32) * 512,), (c(Co, 16) * c(Ci, 32) * 512,)

    @staticmethod
    def split6_shapes(Co, Ci):
        c = lambda v, q: (v + q - 1) // q
        return (c(Ci, 16) * c(Co, 32) * 768,), (c(Co, 16) * c(Ci, 32) * 768,)

    def pack_weight(self, W, WT, WP, fold=None, split=None, split6=None):
        """``split`` / ``split6`` (the pre-split bf16 fragment packs of the GPU kernels) have no fp64 meaning: shapes are checked only."""
        Co, Ci = W.shape
        if split is not None:
            st, sp = self.split_shapes(Co, Ci)
            assert tuple(split[0].shape) == st and tuple(split[1].shape) == sp and (split[2] is None or tuple(split[2].shape) == st)
        WT.zero_()
        WP.zero_()
        WT[:Ci, :Co] = W.t()
        WP[:Co, :Ci] = W
        if fold is not None:
            lnw, lnb, WTf, c12 = fold
            WTf.zero_()
            c12.zero_()
            WTf[:Ci, :Co] = (W * lnw.view(1, Ci)).t()
            c12[0, :Co] = W @ lnw
            c12[1, :Co] = W @ lnb

    def pack_table(self, items, prec=None):
        return list(items), len(items)

    def pack_weights(self, table, total):
        for item in table:
            self.pack_weight(*item)

    @staticmethod
    def kmajor_ok(N, K, a_rows):
        return N % 64 == 0 and a_rows >= (K + 15) // 16 * 16

    @staticmethod
    def kmajor_worth(M, N, Z):
        return N % 64 == 0

    def gemm_kmajor(self, At, Bm, C, M, K, R=None, rowscale=None, ln=None, beta=0.0, fold=None, split=None):
        assert Bm.shape[3] % 64 == 0 and At.shape[2] >= (K + 15) // 16 * 16
        assert float(At[..., K:, :].abs().max()) == 0.0 if At.shape[2] > K else True
        a = At[..., :K, :M].transpose(-1, -2)
        b = Bm
        if ln is not None:
            mu, rs, w, bb = ln
            b = (Bm - mu[:, None, None, :]) * rs[:, None, None, :] * w.view(1, 1, -1, 1) + bb.view(1, 1, -1, 1)
        r = a @ b
        if R is not None:
            r = r + (R * rowscale.unsqueeze(-1) if rowscale is not None else R)
        C.copy_(r + (beta * C if beta != 0.0 else 0))

    # ---- data contract: the reference's numpy chain (util/image_utils.py:133-163, util/degradation_utils.py:21-27)
    def patch_prep(self, clean_img, deg_img, y0, x0, P, mode, sigma, seed, deg_out, clean_out):
        import numpy as np

        def aug(a):
            if mode == 0:
                return a
            k = {1: 0, 2: 1, 3: 1, 4: 2, 5: 2, 6: 3, 7: 3}[mode]
            out = np.rot90(a, k=k) if k else a
            return np.flipud(out) if mode in (1, 3, 5, 7) else out
        c = aug(clean_img.cpu().numpy()[y0:y0 + P, x0:x0 + P]).copy()
        if deg_img is None:
            noise = np.random.Generator(np.random.PCG64(seed)).standard_normal(c.shape)
            d = np.clip(c + noise * sigma, 0, 255).astype(np.uint8)
        else:
            d = aug(deg_img.cpu().numpy()[y0:y0 + P, x0:x0 + P]).copy()
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).to(self.dtype) / 255.0
        clean_out.copy_(to(c))
        deg_out.copy_(to(d))

    # ---- 1x1
    def conv1x1_fwd(self, W, X, Y, ln=None, R=None, beta=0.0, packed=None, ln_compute=False, stats=None):
        B, Ci = X.shape[0], X.shape[1]
        Co = W.shape[0]
        if stats is not None:                      # the statistics of the RESULT (made by the product's epilogue on the GPU)
            assert ln is None and beta == 0.0
            self.conv1x1_fwd(W, X, Y, R=R, packed=packed)
            self.ln_stats(Y, stats[0], stats[1])
            return
        if ln_compute:
            self.ln_stats(X, ln[0], ln[1])
        if packed is not None and ln is not None and len(packed) > 2 and packed[2] is not None:
            # the LN-folded pack is USED (as the bf16x3 kernel uses it), so a stale fold is caught by the CPU tier
            WTf, c12 = packed[2]
            mu, rs = ln[0], ln[1]
            raw = torch.einsum("co,bcn->bon", WTf[:Ci, :Co], X.reshape(B, Ci, -1))
            r = (rs[:, None, :] * raw - (rs * mu)[:, None, :] * c12[0, :Co].view(1, Co, 1) + c12[1, :Co].view(1, Co, 1)).reshape(Y.shape)
        else:
            if packed is not None:
                W = packed[0][:W.shape[1], :W.shape[0]].t()
            r = torch.einsum("oc,bcn->bon", W, _ln_apply(X, ln).reshape(B, Ci, -1)).reshape(Y.shape)
        if R is not None:
            r = r + R
        if beta != 0.0:
            r = r + beta * Y
        Y.copy_(r)

    # round 6: LayerNorm statistics made by the epilogue of the product that stores the tensor (HipBackend.gemm_kmajor_stats);
    # ``prod_stats`` switches TransformerBlock.forward / _stage_fwd onto it in the CPU tier
    prod_stats = False

    def stats_ok(self, M, N, B):
        return self.prod_stats and M <= 96 and N % 128 == 0

    def gemm_kmajor_stats(self, At, Bm, C, M, K, R, stats):
        self.gemm_kmajor(At, Bm, C, M, K, R=R)
        mu, rs = stats
        m = C[:, 0].mean(1)
        v = ((C[:, 0] - m[:, None]) ** 2).mean(1)
        mu.copy_(m)
        rs.copy_(1.0 / torch.sqrt(v + 1e-5))

    def conv1x1_dgrad(self, W, dY, dX, beta=0.0, packed=None):
        B, Co = dY.shape[0], dY.shape[1]
        if packed is not None:
            W = packed[1][:W.shape[0], :W.shape[1]]
        r = torch.einsum("oc,bon->bcn", W, dY.reshape(B, Co, -1)).reshape(dX.shape)
        dX.copy_(r + (beta * dX if beta != 0.0 else 0))

    def conv1x1_wgrad_slabs(self, dY, X, dW, ln=None, region=(0, 1)):
        """the GPU leaves split-K slabs for block_param_reduce; here the finished product is the descriptor"""
        B = X.shape[0]
        r = torch.einsum("bon,bcn->oc", dY.reshape(B, dY.shape[1], -1), _ln_apply(X, ln).reshape(B, X.shape[1], -1))
        return (dW, r)

    def conv1x1_wgrad(self, dY, X, dW, ln=None, beta=1.0):
        B = X.shape[0]
        r = torch.einsum("bon,bcn->oc", dY.reshape(B, dY.shape[1], -1), _ln_apply(X, ln).reshape(B, X.shape[1], -1))
        dW.copy_(r + (beta * dW if beta != 0.0 else 0))

    # ---- bmm
    def bmm_nn(self, A, Bm, C, transA=False, R=None, rowscale=None, beta=0.0):
        a = A.transpose(-1, -2) if transA else A
        r = a @ Bm
        if R is not None:
            r = r + (R * rowscale.unsqueeze(-1) if rowscale is not None else R)
        C.copy_(r + (beta * C if beta != 0.0 else 0))

    def bmm_nt_slabs(self, A, Bm):
        """the GPU hands slabs to attn_softmax; here the finished product plays the descriptor"""
        return A @ Bm.transpose(-1, -2)

    def bmm_nt(self, A, Bm, C):
        C.copy_(A @ Bm.transpose(-1, -2))

    # ---- linear
    def linear_fwd(self, X, W, bias, Y, lrelu=1.0):
        r = F.linear(X, W, bias)
        Y.copy_(F.leaky_relu(r, lrelu) if lrelu != 1.0 else r)

    def linear_dgrad(self, dY, W, dX):
        dX.copy_(dY @ W)

    def linear_wgrad(self, dY, X, dW, beta=1.0):
        dW.copy_(dY.t() @ X + (beta * dW if beta != 0.0 else 0))

    # ---- conv2d
    def conv2d_fwd(self, X, Wt, bias, Y, stride, pad, lrelu=1.0, cmap=0, R=None, mask=None, mslope=1.0):
        r = F.conv2d(X, Wt, bias, stride=stride, padding=pad)
        if lrelu != 1.0:
            r = F.leaky_relu(r, lrelu)
        if R is not None:
            r = r + R
        if cmap == 1:
            r = F.pixel_unshuffle(r, 2)
        elif cmap == 2:
            r = F.pixel_shuffle(r, 2)
        if mask is not None:
            r = torch.where(mask > 0, r, r * mslope)
        Y.copy_(r)

    def conv2d_dgrad(self, dY, Wt, dX, stride, pad, beta=0.0, mask=None, mslope=1.0):
        r = torch.nn.grad.conv2d_input(dX.shape, Wt, dY, stride=stride, padding=pad)
        r = r + (beta * dX if beta != 0.0 else 0)
        if mask is not None:
            r = torch.where(mask > 0, r, r * mslope)
        dX.copy_(r)

    def conv2d_wgrad(self, dY, X, dWt, stride, pad, beta=1.0):
        r = torch.nn.grad.conv2d_weight(X, dWt.shape, dY, stride=stride, padding=pad)
        dWt.copy_(r + (beta * dWt if beta != 0.0 else 0))

    def pixel_shuffle(self, inp, out, mode):
        out.copy_(F.pixel_unshuffle(inp, 2) if mode == 1 else F.pixel_shuffle(inp, 2))

    # ---- LayerNorm
    def ln_stats(self, x, mu, rs):
        B, C = x.shape[0], x.shape[1]
        xx = x.reshape(B, C, -1)
        m = xx.mean(1)
        v = ((xx - m[:, None]) ** 2).mean(1)
        mu.copy_(m)
        rs.copy_(1.0 / torch.sqrt(v + 1e-5))

    def ln_bwd(self, g, x, mu, rs, w, dres, dx, dw, db, slot=None):
        B, C = x.shape[0], x.shape[1]
        xx, gg = x.reshape(B, C, -1), g.reshape(B, C, -1)
        xh = (xx - mu[:, None]) * rs[:, None]
        gh = gg * w.view(1, C, 1)
        r = rs[:, None] * (gh - gh.mean(1, keepdim=True) - xh * (gh * xh).mean(1, keepdim=True))
        r = r.reshape(x.shape)
        if dres is not None:
            r = r + dres
        dx.copy_(r)
        if slot is None:
            dw.add_((gg * xh).sum((0, 2)))
            db.add_(gg.sum((0, 2)))
        else:                                         # deferred: consumed by block_param_reduce
            if not hasattr(self, "_ln_def"):
                self._ln_def = {}
            self._ln_def[slot] = ((gg * xh).sum((0, 2)), gg.sum((0, 2)))

    def block_param_reduce(self, C, gw1, gb1, gw2, gb2, dWo_part, gWo, dtemp_part, gtemp, slabs=(), close_block=False):
        for d in slabs:
            if d is not None:
                d[0].add_(d[1])
        (a1, b1), (a2, b2) = self._ln_def.pop(0), self._ln_def.pop(1)
        gw1.add_(a1); gb1.add_(b1); gw2.add_(a2); gb2.add_(b2)
        gWo.add_(dWo_part.sum(0))
        gtemp.add_(dtemp_part.sum(0))

    # ---- depthwise
    def dwconv3x3(self, x, w, y, flip=False):
        C = x.shape[1]
        k = w.view(C, 1, 3, 3)
        if flip:
            k = k.flip(-1, -2)
        y.copy_(F.conv2d(x, k, padding=1, groups=C))

    def gdfn_gate_fwd(self, p, w, g):
        C2 = p.shape[1]
        d = F.conv2d(p, w.view(C2, 1, 3, 3), padding=1, groups=C2)
        h = C2 // 2
        g.copy_(F.gelu(d[:, :h]) * d[:, h:])

    def gdfn_gate_bwd(self, p, w, dg, dd, dw=None):
        C2 = p.shape[1]
        h = C2 // 2
        d = F.conv2d(p, w.view(C2, 1, 3, 3), padding=1, groups=C2)
        a, b = d[:, :h], d[:, h:]
        cdf = 0.5 * (1 + torch.erf(a / math.sqrt(2)))
        pdf = torch.exp(-0.5 * a * a) / math.sqrt(2 * math.pi)
        dd[:, :h].copy_(dg * b * (cdf + a * pdf))
        dd[:, h:].copy_(dg * a * cdf)
        if dw is not None:
            self.dwconv3x3_wgrad(dd, p, dw)

    def gdfn_bwd(self, p, w, dg, dp, dw):
        dd = torch.empty_like(p)
        self.gdfn_gate_bwd(p, w, dg, dd, dw=dw)
        self.dwconv3x3(dd, w, dp, flip=True)

    def dwconv3x3_bwd(self, dy, x, w, dx, dw):
        self.dwconv3x3(dy, w, dx, flip=True)
        self.dwconv3x3_wgrad(dy, x, dw)

    def dwconv3x3_wgrad(self, dy, x, dw):
        C = x.shape[1]
        r = torch.nn.grad.conv2d_weight(x, (C, 1, 3, 3), dy, padding=1, groups=C)
        dw.add_(r.view(C, 9))

    # ---- attention small
    def row_sumsq(self, x, out):
        out.copy_((x.reshape(x.shape[0], x.shape[1], -1) ** 2).sum(-1))

    def attn_softmax(self, Graw, sq, temp, Gn, A):
        B, hd, c, _ = Graw.shape
        C = hd * c
        nq = sq[:, :C].sqrt().clamp_min(1e-12).view(B, hd, c, 1)
        nk = sq[:, C:].sqrt().clamp_min(1e-12).view(B, hd, 1, c)
        g = Graw / (nq * nk)
        Gn.copy_(g)
        A.copy_(torch.softmax(g * temp.view(1, hd, 1, 1), -1))

    def attn_core_fwd(self, u, temp, WoT, sq, Gn, A, MfT):
        """same support rule as rcot_attn_core_fwd, so that the CPU tier walks both routes of the schedule"""
        B, hd, c, _ = Gn.shape
        C, N = hd * c, u.shape[2] * u.shape[3]
        if (C % 16) or (N % 256) or N > getattr(self, 'attn_core_maxn', 4096) or (N > 4096 and N % 512) or c not in (24, 48, 96):
            return False
        uu = u.reshape(B, 3, hd, c, N)
        self.row_sumsq(u[:, :2 * C], sq)
        self.attn_softmax(uu[:, 0] @ uu[:, 1].transpose(-1, -2), sq, temp, Gn, A)
        # MfT[b][h c + j][m] = sum_i A[b,h][i][j] WoT[h c + i][m]
        W = WoT[:C, :C].reshape(hd, c, C)
        MfT.view(B, hd, c, C).copy_(torch.einsum("bhij,him->bhjm", A, W))
        return True

    @staticmethod
    def attn_fused_ok(c):
        return c in (48, 96)

    def attn_core_bwd(self, dM, Wo, A, Gn, sq, temp, Mf, dWo_part, dtemp_part, Eq, EqT, Dq, Dk):
        B, hd, c, _ = A.shape
        if ((hd * c) % 16) or c not in (24, 48, 96):
            return False
        self.attn_bwd_fused(dM, Wo, A, Gn, sq, temp, Mf, dWo_part, dtemp_part, Eq, EqT, Dq, Dk)
        return True

    def attn_bwd_fused(self, dM, Wo, A, Gn, sq, temp, Mf, dWo_part, dtemp_part, Eq, EqT, Dq, Dk):
        B, hd, c, _ = A.shape
        C = hd * c
        Wh = Wo.view(C, hd, c).permute(1, 0, 2).unsqueeze(0)            # [1, hd, C, c]
        Dh = dM.view(B, C, hd, c).permute(0, 2, 1, 3)                    # [B, hd, C, c]
        Mf.view(B, C, hd, c).permute(0, 2, 1, 3).copy_(Wh @ A)
        dWo_part.view(B, C, hd, c).permute(0, 2, 1, 3).copy_(Dh @ A.transpose(-1, -2))
        dA = Wh.transpose(-1, -2) @ Dh
        self.attn_bwd_small(dA.contiguous(), A, Gn, sq, temp, dtemp_part, Eq, EqT, Dq, Dk)

    def attn_bwd_small(self, dA, A, Gn, sq, temp, dtemp_part, Eq, EqT, Dq, Dk):
        B, hd, c, _ = A.shape
        C = hd * c
        dS = A * (dA - (dA * A).sum(-1, keepdim=True))
        sg = dS * Gn
        dtemp_part.copy_(sg.sum((-1, -2)))
        q2, k2 = sq[:, :C].view(B, hd, c), sq[:, C:].view(B, hd, c)
        nq, nk = q2.sqrt().clamp_min(1e-12), k2.sqrt().clamp_min(1e-12)
        tau = temp.view(1, hd, 1, 1)
        Eq.copy_(tau * dS / (nq.unsqueeze(-1) * nk.unsqueeze(-2)))
        EqT.copy_(Eq.transpose(-1, -2))
        Dq.copy_((-(tau.squeeze(-1)) * sg.sum(-1) / q2).reshape(B, C))
        Dk.copy_((-(tau.squeeze(-1)) * sg.sum(-2) / k2).reshape(B, C))

    def batch_reduce(self, src, dst, beta=1.0):
        dst.copy_(src.sum(0).reshape(dst.shape) + (beta * dst if beta != 0.0 else 0))

    # ---- elementwise
    def lrelu_bwd(self, dy, a, dz, slope=0.2):
        dz.copy_(torch.where(a > 0, dy, dy * slope))

    def bias_grad(self, dz, db):
        db.add_(dz.reshape(dz.shape[0], dz.shape[1], -1).sum((0, 2)))

    def axpby(self, x, y, out, a=1.0, b=1.0):
        out.copy_(a * x + (b * y if y is not None else 0))

    def fill(self, t, v=0.0):
        t.fill_(v)

    # ---- MPRNet pieces (csrc/mprnet_ops.hip)
    def prelu_fwd(self, x, slope, y):
        y.copy_(torch.where(x > 0, x, slope * x))

    def prelu_bwd(self, dy, x, slope, dx, dslope):
        dslope.add_(torch.where(x > 0, torch.zeros_like(x), x * dy).sum())
        dx.copy_(torch.where(x > 0, dy, slope * dy))

    def row_dot(self, a, b, out, scale=1.0):
        out.copy_(((a * b) if b is not None else a).sum((2, 3)).reshape(out.shape) * scale)

    def row_scale_add(self, a, s, x, t, tscale, out):
        B, C = a.shape[:2]
        r = a * s.reshape(B, C, 1, 1)
        if x is not None:
            r = r + x
        if t is not None:
            r = r + t.reshape(B, C, 1, 1) * tscale
        out.copy_(r)

    def ca_gate_fwd(self, mean, W1, W2, hid, gate):
        hid.copy_(torch.relu(mean @ W1.t()))
        gate.copy_(torch.sigmoid(hid @ W2.t()))

    def ca_gate_bwd(self, dgate, gate, hid, mean, W1, W2, dW1, dW2, dmean):
        ds = dgate * gate * (1 - gate)
        dh = (ds @ W2) * (hid > 0)
        dmean.copy_(dh @ W1)
        dW2.add_(ds.t() @ hid)
        dW1.add_(dh.t() @ mean)

    def bilinear_down2(self, x, y):
        y.copy_(F.interpolate(x, scale_factor=0.5, mode="bilinear", align_corners=False))

    def bilinear_down2_bwd(self, dy, dx, beta=0.0):
        dx.copy_(0.25 * dy.repeat_interleave(2, 2).repeat_interleave(2, 3) + (beta * dx if beta != 0.0 else 0))

    def bilinear_up2(self, x, skip, y):
        y.copy_(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False) + (skip if skip is not None else 0))

    def bilinear_up2_bwd(self, dy, dx):
        z = torch.zeros_like(dx).requires_grad_(True)
        F.interpolate(z, scale_factor=2, mode="bilinear", align_corners=False).backward(dy)      # the adjoint of a linear map
        dx.copy_(z.grad)

    def conv_weight_flip(self, src, dst, table, n, Co, Ci, K):
        per = Co * Ci * K * K
        for j in range(n):
            so, do = int(table[2 * j]), int(table[2 * j + 1])
            w = src[so:so + per].view(Co, Ci, K, K)
            dst[do:do + per].copy_(w.flip(2, 3).transpose(0, 1).reshape(-1))

    def lerp(self, t, f, alpha, out):
        al = alpha.view(-1, *([1] * (t.dim() - 1)))
        out.copy_(al * t + (1 - al) * f)

    def gp_penalty(self, g, norms, u0, gp_out, inv_global_batch):
        B = g.shape[0]
        n = g.reshape(B, -1).norm(dim=1)
        norms.copy_(n)
        coef = 20.0 * inv_global_batch * (n - 1) / n
        u0.copy_(g * coef.view(B, *([1] * (g.dim() - 1))))
        gp_out.fill_(float(10.0 * inv_global_batch * ((n - 1) ** 2).sum()))

    # ---- OT cost
    def ot_reduce(self, degraded, restored, target, sums):
        B = degraded.shape[0]
        r2 = ((degraded - restored) ** 2).reshape(B, -1).sum(1)
        sums.zero_()
        sums[:B] = r2
        sums[2 * B] = r2.sum()
        if target is not None:
            l1 = (restored - target).abs().reshape(B, -1).sum(1)
            sums[B:2 * B] = l1
            sums[2 * B + 1] = l1.sum()

    def ot_spectrum(self, degraded, restored, de_id, gF, spec):
        res = degraded - restored
        fr = torch.fft.fft2(res)
        mag = fr.abs()
        spec.copy_(mag.reshape(res.shape[0], -1).sum(1))
        u = torch.where(mag > 0, fr / mag.clamp_min(1e-30), torch.zeros_like(fr))
        gF.copy_(torch.fft.ifft2(u).real / 3.0)

    def ot_grad(self, degraded, restored, target, de_id, gF, sums, spec, dout, scal, sigma, Sigma, global_batch):
        B = degraded.shape[0]
        per = degraded.numel() // B
        Mg = float(global_batch) * per
        rmse = torch.sqrt(sums[2 * B] / Mg)
        res = degraded - restored
        l2 = (de_id < 3).view(B, 1, 1, 1)
        branch = torch.where(l2, res / 3.0, gF if gF is not None else torch.zeros_like(res))
        g = -sigma * (res / (Mg * rmse) + branch)
        if target is not None:
            g = g + Sigma / Mg * torch.sign(restored - target)
        dout.add_(g)
        four = torch.where(de_id < 3, sums[:B] / 6.0, spec / per).sum()
        scal[0], scal[1], scal[2] = rmse, four, sums[2 * B + 1] / Mg

    # ---- optimizers
    def rmsprop_step(self, p, g, sq, n, lr, alpha=0.99, eps=1e-8, grad_scale=1.0):
        gg = g[:n] * grad_scale
        sq[:n].mul_(alpha).addcmul_(gg, gg, value=1 - alpha)
        p[:n].addcdiv_(gg, sq[:n].sqrt() + eps, value=-lr)

    def adam_step(self, p, g, m, v, n, lr, step, b1=0.9, b2=0.999, eps=1e-8, grad_scale=1.0):
        gg = g[:n] * grad_scale
        m[:n].mul_(b1).add_(gg, alpha=1 - b1)
        v[:n].mul_(b2).addcmul_(gg, gg, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        p[:n].addcdiv_(m[:n], v[:n].sqrt() / math.sqrt(bc2) + eps, value=-lr / bc1)
