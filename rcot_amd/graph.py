"""HIP-graph replay of the minimax iteration (reference loop body trainer.py:247-346).

One iteration is ~2 800 kernel launches whose shapes, pointers and scalar arguments do not change from one
iteration to the next (static batch shape, flat parameter buffers, activations from a private pool), while enqueueing
them from Python costs ~17 us each: ~50 ms of host time per iteration, more than the kernels will need once they are
tuned.  ``GraphedMinimax`` therefore captures the launch sequence of ``MinimaxStep.iteration`` ONCE per distinct
configuration (batch shape, paired flag, learning rates, spectral branch) into HIP graphs and afterwards only copies the
new batch into the captured input buffers and replays.

Data-parallel runs: RCCL collectives are host-driven and stay OUTSIDE the graphs.  The capture is cut wherever the
gradient reducer wants to launch a bucket or finish (``GradReducer`` calls back through ``host_action``), which gives
a list  [graph_0, action_0, graph_1, action_1, ...]  replayed in order: every segment is one hipGraphLaunch, every
action is the eager all-reduce of a bucket on the reducer's side stream, exactly as in the eager path.

PyTorch is used for what it is here for: ``torch.cuda.CUDAGraph`` wraps hipGraph capture/instantiate/launch and gives
the captured launches a private memory pool (so the pointers baked into the graph stay valid).
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch


class SegmentedCapture:
    """Capture a launch sequence into HIP-graph segments separated by host-side actions."""

    def __init__(self):
        self.items: List[object] = []          # CUDAGraph | callable
        self.pool = None
        self._g: Optional[torch.cuda.CUDAGraph] = None
        self._ctx = None
        self.stream = torch.cuda.Stream()
        self.capturing = False

    # ---- capture
    def _begin(self):
        self._g = torch.cuda.CUDAGraph()
        kw = {"pool": self.pool} if self.pool is not None else {}
        self._ctx = torch.cuda.graph(self._g, stream=self.stream, **kw)
        self._ctx.__enter__()

    def _end(self):
        self._ctx.__exit__(None, None, None)
        if self.pool is None:
            self.pool = self._g.pool()
        self.items.append(self._g)
        self._g = self._ctx = None

    def host_action(self, fn: Callable[[], None]):
        """Called (through the reducer) from inside the captured region: cut the graph here, run ``fn`` eagerly now and at
        the same position of every replay, and continue capturing into the next segment."""
        self._end()
        with torch.cuda.stream(self.stream):
            fn()
        self.items.append(fn)
        self._begin()

    def capture(self, body: Callable[[], None]):
        self.capturing = True
        torch.cuda.synchronize()
        self._begin()
        try:
            body()
        finally:
            self._end()
            self.capturing = False
        torch.cuda.synchronize()

    # ---- replay
    def replay(self):
        if len(self.items) == 1:
            self.items[0].replay()
            return
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for it in self.items:
                if isinstance(it, torch.cuda.CUDAGraph):
                    it.replay()
                else:
                    it()
        cur.wait_stream(self.stream)

    @property
    def n_graphs(self):
        return sum(isinstance(i, torch.cuda.CUDAGraph) for i in self.items)


class GraphedMinimax:
    """Drop-in for ``MinimaxStep.iteration``: same arguments and return value, HIP-graph replay underneath."""

    def __init__(self, step, warmup: int = 1):
        self.step = step
        self.cache = {}
        self.warmup = warmup
        self._warmed = False
        self.enabled = step.T.store.flat.is_cuda
        # the Adam kernels take the step count BY VALUE (bias correction), so an Adam run cannot be replayed
        if step.To.kind != "RMSprop" or step.Fo.kind != "RMSprop":
            self.enabled = False

    def _state_tensors(self):
        st = self.step
        out = [st.T.store.flat, st.F.store.flat]
        for o in (st.To, st.Fo):
            out += list(o._state_tensors().values())
        return out

    def _key(self, degraded, paired):
        st = self.step
        return (tuple(degraded.shape), bool(paired), st.To.param_groups[0]["lr"], st.Fo.param_groups[0]["lr"],
                bool(st._any_spectral), int(st.be.prec))

    def _prepare(self, degraded, target, de_id, alpha, paired):
        st = self.step
        if not self._warmed:
            # One eager pass before the first capture (one-time hipFuncSetAttribute calls inside the launchers, lazily
            # created weight packs and their device tables, allocator warm-up).  It must not count as a training
            # iteration: parameters and optimizer state are put back afterwards.
            saved = [t.clone() for t in self._state_tensors()]
            for _ in range(self.warmup):
                st.iteration(degraded, target, de_id, alpha, paired)
            for t, s_ in zip(self._state_tensors(), saved):
                t.copy_(s_)
            del saved
            for net in (st.T, st.F):
                if hasattr(net, "repack"):
                    net.repack()
            self._warmed = True
        x, y, d, a = degraded.clone(), target.clone(), de_id.clone(), alpha.clone()
        cap = SegmentedCapture()
        box = {}
        reducers = [r for r in (st.redT, st.redF) if r.enabled]
        for r in reducers:
            r.host_action = cap.host_action

        def body():
            box["out"] = st.iteration(x, y, d, a, paired)
        try:
            cap.capture(body)
        finally:
            for r in reducers:
                r.host_action = None
        return dict(cap=cap, x=x, y=y, d=d, a=a, out=box["out"], logs=dict(st.logs))

    def iteration(self, degraded, target, de_id, alpha, paired: bool):
        st = self.step
        if not self.enabled:
            return st.iteration(degraded, target, de_id, alpha, paired)
        key = self._key(degraded, paired)
        ent = self.cache.get(key)
        if ent is None:
            st.be.pcm_pinning = True       # padded-plane buffers touched from here to the end of the capture are baked into it
            try:
                ent = self._prepare(degraded, target, de_id, alpha, paired)
            finally:
                st.be.pcm_pinning = False
            self.cache[key] = ent              # (capturing executes nothing: fall through to the first replay)
        ent["x"].copy_(degraded, non_blocking=True)
        ent["y"].copy_(target, non_blocking=True)
        ent["d"].copy_(de_id, non_blocking=True)
        ent["a"].copy_(alpha, non_blocking=True)
        ent["cap"].replay()
        st.logs = ent["logs"]
        return ent["out"]
