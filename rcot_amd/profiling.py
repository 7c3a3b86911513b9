"""Per-entry-point HIP-event timing of a HipBackend, with the ALGORITHMIC work of every call
(flops for the MFMA GEMM family, bytes = every tensor argument once for the HBM-bound kernels).
Used by bench.py for the ``roofline`` object; events are recorded on the stream the kernels are
launched on (torch's current stream)."""
from __future__ import annotations

import ctypes
from collections import defaultdict
from typing import Dict

import torch

GEMM_OPS = ("gemm_kmajor", "gemm_kmajor_stats", "gemm_kmajor_multi", "conv1x1_fwd", "conv1x1_dgrad", "conv1x1_dgrad_wgrad_slabs", "conv1x1_wgrad", "conv1x1_wgrad_slabs", "bmm_nn", "bmm_nt",
            "bmm_nt_slabs", "linear_fwd", "linear_dgrad",
            "linear_wgrad", "conv2d_fwd", "conv2d_dgrad", "conv2d_wgrad")
OTHER_OPS = ("ln_stats", "ln_bwd", "dwconv3x3", "gdfn_gate_fwd", "gdfn_gate_bwd", "gdfn_bwd", "dwconv3x3_wgrad", "dwconv3x3_bwd", "row_sumsq",
             "attn_softmax", "attn_bwd_small", "batch_reduce", "block_param_reduce", "lrelu_bwd", "bias_grad", "axpby", "fill", "lerp", "gp_penalty",
             "pixel_shuffle", "pack_weight", "ot_reduce", "ot_spectrum", "ot_grad", "rmsprop_step", "adam_step")


def _numel(t):
    """distinct elements a kernel has to touch: broadcast (stride-0) dimensions count once"""
    if not isinstance(t, torch.Tensor):
        return 0
    n = 1
    for sz, st in zip(t.shape, t.stride()):
        if st != 0:
            n *= sz
    return n


def _flops(name, a, kw):
    if name == "conv1x1_dgrad_wgrad_slabs":          # (W, dY, dX, X, dW): both products
        return 4.0 * a[0].shape[0] * a[0].shape[1] * a[1].shape[0] * (a[1].numel() // (a[1].shape[0] * a[1].shape[1]))
    if name.startswith("conv1x1"):
        # (W, X, Y) / (W, dY, dX) / (dY, X, dW): 2 * Co * Ci * B * N
        ts = [t for t in a[:3]]
        w = ts[0] if name not in ("conv1x1_wgrad", "conv1x1_wgrad_slabs") else ts[2]
        x = ts[1]
        return 2.0 * w.shape[0] * w.shape[1] * x.shape[0] * (x.numel() // (x.shape[0] * x.shape[1]))
    if name == "gemm_kmajor_multi":                  # [(At, Bm, C, M, K, R, rowscale), ...]
        return sum(2.0 * it[2].shape[0] * it[2].shape[1] * it[3] * it[2].shape[3] * it[4] for it in a[0])
    if name in ("gemm_kmajor", "gemm_kmajor_stats"):
        At, Bm, C = a[:3]
        return 2.0 * C.shape[0] * C.shape[1] * C.shape[2] * C.shape[3] * Bm.shape[2]
    if name == "bmm_nn":
        A, Bm, C = a[:3]
        return 2.0 * C.shape[0] * C.shape[1] * C.shape[2] * C.shape[3] * Bm.shape[2]
    if name in ("bmm_nt", "bmm_nt_slabs"):
        A, Bm = a[:2]
        return 2.0 * A.shape[0] * A.shape[1] * A.shape[2] * Bm.shape[2] * A.shape[3]
    if name.startswith("linear"):
        if name == "linear_fwd":
            X, W = a[0], a[1]
        elif name == "linear_dgrad":
            X, W = a[0], a[1]
        else:
            X, W = a[0], a[2]
        return 2.0 * X.shape[0] * W.shape[0] * W.shape[1]
    if name == "conv2d_fwd":
        Wt, Y = a[1], a[3]
    elif name == "conv2d_dgrad":
        Wt, Y = a[1], a[0]
    else:
        Wt, Y = a[2], a[0]
    co, ci, kh, kw_ = Wt.shape
    if name == "conv2d_fwd" and (kw.get("cmap", a[7] if len(a) > 7 else 0)):
        pix = Y.numel() // Y.shape[0] // co
    else:
        pix = Y.shape[2] * Y.shape[3]
    return 2.0 * Y.shape[0] * co * pix * ci * kh * kw_


class OpTimer:
    """Wraps the kernel methods of one backend instance; ``summary()`` after a device sync."""

    def __init__(self, be):
        self.be = be
        self._kbuf = ctypes.create_string_buffer(192)
        self._kname = getattr(be.L, "rcot_last_kernel", None)       # symbol of the kernel the last dispatcher launched (api.hip)
        self.records = []
        self._calls = {}
        self._depth = 0
        self._orig = {}
        for name in GEMM_OPS + OTHER_OPS:
            fn = getattr(be, name)
            self._orig[name] = fn
            setattr(be, name, self._wrap(name, fn))

    def _wrap(self, name, fn):
        gemm = name in GEMM_OPS

        def timed(*a, **kw):
            if self._depth:                     # an op that delegates (conv1x1 -> gemm_kmajor) is timed once, outermost
                return fn(*a, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            seq0 = self._kname(self._kbuf, 192) if self._kname is not None else 0
            if name == "block_param_reduce" and kw.get("close_block") and not getattr(self.be, "defer_close", False):
                # the launch that closes a block first joins the side stream: that wait (the block's weight-gradient kernels, up to
                # milliseconds) is not this kernel's time — join before the bracket opens (the call's own join is then a no-op)
                self.be.side_join()
            s.record()
            self._depth += 1
            try:
                r = fn(*a, **kw)
            finally:
                self._depth -= 1
            e.record()
            sym = name                           # entry points with ONE kernel behind them are named by the entry point
            if self._kname is not None and self._kname(self._kbuf, 192) != seq0:
                sym = self._kbuf.value.decode()
            nbytes = 4.0 * (sum(_numel(t) for t in a) + sum(_numel(t) for t in kw.values()))
            if name == "gemm_kmajor_multi":
                nbytes = 4.0 * sum(_numel(t) for it in a[0] for t in it)
            ln = kw.get("ln")
            if ln is not None:
                nbytes += 4.0 * sum(_numel(t) for t in ln)
            key = name + str(tuple(tuple(t.shape) for t in (a[:4] if name != "gemm_kmajor_multi" else [it[2] for it in a[0]]) if isinstance(t, torch.Tensor)))
            if ln is not None:
                key += "+ln"
            self.records.append((name, s, e, _flops(name, a, kw) if gemm else 0.0, nbytes, key, sym))
            if key not in self._calls:
                self._calls[key] = (fn, a, kw)
            return r
        return timed

    def remove(self):
        for name, fn in self._orig.items():
            setattr(self.be, name, fn)

    def summary(self) -> Dict[str, dict]:
        torch.cuda.synchronize()
        out = defaultdict(lambda: dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
        for name, s, e, fl, by, _key, _sym in self.records:
            d = out[name]
            d["calls"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += fl
            d["bytes"] += by
        return dict(out)

    def by_shape(self, top: int = 25):
        """[(op+shapes, calls, ms, TFLOP/s or GB/s)] sorted by time — where to look when tuning."""
        torch.cuda.synchronize()
        agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
        for name, s, e, fl, by, key, _sym in self.records:
            d = agg[key]
            d[0] += 1
            d[1] += s.elapsed_time(e)
            d[2] += fl
            d[3] += by
        rows = []
        for k, (n, ms, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
            rate = f"{fl / ms / 1e9:.1f} TF/s" if fl > 0 else f"{by / ms / 1e6:.0f} GB/s"
            rows.append((k, n, round(ms, 3), rate))
        return rows

    def by_symbol(self):
        """{kernel symbol: dict(ms, calls, bytes, flops, entry_points, shapes={shape key: [calls, ms, bytes, flops]})} of the recorded
        step, in-situ (HIP events around every launch inside the iteration): what ``rocprofv3 --kernel-trace --stats`` lists per
        symbol, with the ALGORITHMIC work of every launch next to it."""
        torch.cuda.synchronize()
        out = {}
        for name, s, e, fl, by, key, sym in self.records:
            d = out.setdefault(sym, dict(ms=0.0, calls=0, bytes=0.0, flops=0.0, entry_points=set(), shapes={}))
            t = s.elapsed_time(e)
            d["ms"] += t
            d["calls"] += 1
            d["bytes"] += by
            d["flops"] += fl
            d["entry_points"].add(name)
            q = d["shapes"].setdefault(key, [0, 0.0, 0.0, 0.0])
            q[0] += 1
            q[1] += t
            q[2] += by
            q[3] += fl
        return out

    def replay_dominant(self, reps: int = 30, gemm_only: bool = False):
        """Re-launch the single (op, shape) that took the most time in the recorded step — ANY entry point, MFMA GEMM or
        HBM-bound kernel — ``reps`` times back-to-back between two HIP events on the launch stream: the per-launch
        duration without host gaps.  Returns dict(key, name, ms, flops, bytes, calls, step_ms)."""
        torch.cuda.synchronize()
        agg = defaultdict(lambda: [0.0, 0.0, 0.0, 0, ""])
        for name, s, e, fl, by, key, _sym in self.records:
            if gemm_only and name not in GEMM_OPS:
                continue
            d = agg[key]
            d[0] += s.elapsed_time(e)
            d[1], d[2] = fl, by
            d[3] += 1
            d[4] = name
        key = max(agg, key=lambda k: agg[k][0])
        fn, a, kw = self._calls[key]
        for _ in range(3):
            fn(*a, **kw)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn(*a, **kw)
        e.record()
        torch.cuda.synchronize()
        t, fl, by, n, name = agg[key]
        return dict(key=key, name=name, ms=s.elapsed_time(e) / reps, flops=fl, bytes=by, calls=n, step_ms=t)
