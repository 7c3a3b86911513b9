"""Host-side launch plans: the minimax iteration (reference loop body trainer.py:247-346) as a recorded command list.

One iteration is ~2 700 kernel launches whose shapes, pointers and scalar arguments do not change from one iteration to the
next, while walking the Python schedule (views, shape checks, dispatch decisions, ~30 allocations per block) costs ~18 us
per launch: ~50 ms of host time per iteration — as much as the kernels of the three small levels of the network need, so
any kernel gain there would disappear behind the host.  HIP-graph replay (rcot_amd/graph.py) removes the host but adds ~2 us
of GPU time to every node on ROCm 7.2 (68 vs 63 ms for the transport-map unit).

A ``LaunchPlan`` is the third way: the schedule runs ONCE with a recording proxy in place of the library handle — every
``rcot_*`` call that launched something is kept as (function, argument tuple) — while every tensor the schedule allocates comes
from a private memory pool, so the addresses stay valid.  Afterwards an iteration is a loop over that list: the same eager
launches on the same stream in the same order (no graph nodes, nothing between the kernels that was not there before), at the
price of one ctypes call each (~2 us + the launch itself).  Host-driven steps inside the iteration (the data-parallel
reducer's collectives, the few torch-side fills) are kept as Python callables at their position in the list.

What makes this valid is what makes graph capture valid: static shapes, no host read of device values inside the region,
scalars passed by value that do not change (the learning rate is part of the cache key; Adam's step count is not static, so
Adam runs eagerly).  Recording executes the kernels (unlike a graph capture), so the recording pass IS a training iteration.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional

import torch

from . import lib as _lib

_HOST_ONLY = {"rcot_abi_version", "rcot_ln_bwd_rows", "rcot_last_kernel"}          # no launch, no stream argument


class _RecordingLib:
    """Stands in for the ctypes library handle of a HipBackend while a plan is recorded."""

    def __init__(self, real, cmds: list, side_handle):
        self._real, self._cmds, self._side = real, cmds, side_handle

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if name in _HOST_ONLY or not name.startswith("rcot_"):
            return fn
        cmds, side = self._cmds, self._side

        def rec(*a):
            rc = fn(*a)
            if rc == 0:                          # (EUNSUPPORTED launched nothing: the caller takes another route)
                # every entry point takes the stream LAST: replay supplies the current stream, or the backend's side stream for
                # what was launched there (weight gradients next to the data-gradient chain)
                cmds.append((fn, a[:-1], side is not None and a[-1] == side))
            return rc
        self.__dict__[name] = rec
        return rec


class LaunchPlan:
    """A recorded launch sequence; ``replay()`` re-issues it on the calling thread's current stream."""

    def __init__(self, backend):
        self.be = backend
        self.cmds: List[tuple] = []
        self.pool = None
        self.keep = []                           # whatever must outlive the recording (static inputs, results)
        self._bound = {}                         # stream handle -> the command list bound to it (replay)

    @property
    def n_launches(self):
        return sum(1 for c in self.cmds if c[1] is not None)

    # ---- recording
    def host_action(self, fn: Callable[[], None]):
        """a host-driven step at this position (collectives, torch-side fills): runs now and at every replay"""
        fn()
        self.cmds.append((fn, None, False))

    def record(self, body: Callable[[], None]):
        be = self.be
        assert getattr(be, "_plan", None) is None, "plans do not nest"
        real = be.L
        self.pool = torch.cuda.MemPool()
        # the backend's side stream stays in use while recording: its launches are marked, and the cross-stream waits / events of
        # the schedule (HipBackend._host) are kept as host actions at their positions
        self._side = be._side.cuda_stream if getattr(be, "_side", None) is not None else None
        be.L = _RecordingLib(real, self.cmds, self._side)
        be._plan = self
        be.pcm_pinning = True
        try:
            with torch.cuda.use_mem_pool(self.pool):
                body()
                be.side_join()                   # the plan ends joined: a replay starts from the state the recording started from
        finally:
            be.L, be._plan = real, None
            be.pcm_pinning = False
        return self

    # ---- replay
    def _bind(self, st):
        """the command list with the stream argument in place and every scalar already a ctypes object of the entry point's
        declared type: a replayed call then skips ctypes' per-argument conversion (3.8 -> 2.3 us for the 39-argument
        rcot_gemm_kmajor, ~2 ms of host time per iteration)"""
        side, out = getattr(self, "_side", None), []
        for fn, a, on_side in self.cmds:
            if a is None:
                out.append((fn, None))
                continue
            full = a + ((side if on_side else st),)
            out.append((fn, tuple(t(v) if type(v) in (int, float) else v for t, v in zip(fn.argtypes, full))))
        return out

    def replay(self):
        st = self.be._st()
        bound = self._bound.get(st)
        if bound is None:
            bound = self._bound[st] = self._bind(st)
        for fn, a in bound:
            if a is None:
                fn()
            else:
                rc = fn(*a)
                if rc:
                    _lib.check(rc, getattr(fn, "__name__", "rcot_*") + " (plan replay)")


class PlannedMinimax:
    """Drop-in for ``MinimaxStep.iteration``: same arguments and return value, a LaunchPlan per configuration underneath."""

    def __init__(self, step, warmup: int = 1):
        self.step = step
        self.cache = {}
        self.warmup = warmup
        self._warmed = set()                     # (batch shape, arithmetic, spectral branch) that ran eagerly once
        self.enabled = step.T.store.flat.is_cuda and hasattr(torch.cuda, "MemPool")
        # the Adam kernels take the step count BY VALUE (bias correction): not a static argument
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
                bool(st._any_spectral), int(st.be.prec), st.grad_probe is not None)

    def _prepare(self, degraded, target, de_id, alpha, paired):
        st = self.step
        wkey = (tuple(degraded.shape), int(st.be.prec), bool(st._any_spectral))
        if wkey not in self._warmed:
            # One eager pass first in THIS shape and arithmetic, as before a graph capture: whatever the schedule creates lazily
            # (one-time hipFuncSetAttribute calls inside the launchers, weight packs and their device tables, the padded-plane
            # operands and index tables of the split-bf16 convolutions) must exist before the recording, in the ordinary pool — a
            # first use INSIDE the recording ended in a memory access fault on the next replay (scripts/dbg/prec_switch_repro.py:
            # fp32 plan, then the first bf16x3 iteration recorded without a warm-up).  The pass must not count as a training
            # iteration: parameters and optimizer state are put back.
            saved = [t.clone() for t in self._state_tensors()]
            for _ in range(self.warmup):
                st.iteration(degraded, target, de_id, alpha, paired)
            for t, s_ in zip(self._state_tensors(), saved):
                t.copy_(s_)
            del saved
            for net in (st.T, st.F):
                if hasattr(net, "repack"):
                    net.repack()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()             # the recording allocates its working set again, in its own pool
            self._warmed.add(wkey)
        plan = LaunchPlan(st.be)
        box = {}
        reducers = [r for r in (st.redT, st.redF) if r.enabled]
        for r in reducers:
            r.host_action = plan.host_action

        def body():
            box["io"] = (degraded.clone(), target.clone(), de_id.clone(), alpha.clone())      # static inputs, in the plan's pool
            box["out"] = st.iteration(*box["io"], paired)
        try:
            plan.record(body)
        finally:
            for r in reducers:
                r.host_action = None
        x, y, d, a = box["io"]
        return dict(plan=plan, x=x, y=y, d=d, a=a, out=box["out"], logs=dict(st.logs))

    def iteration(self, degraded, target, de_id, alpha, paired: bool):
        st = self.step
        if not self.enabled:
            return st.iteration(degraded, target, de_id, alpha, paired)
        key = self._key(degraded, paired)
        ent = self.cache.get(key)
        if ent is None:
            ent = self.cache[key] = self._prepare(degraded, target, de_id, alpha, paired)     # (recording ran the iteration)
            return ent["out"]
        ent["x"].copy_(degraded, non_blocking=True)
        ent["y"].copy_(target, non_blocking=True)
        ent["d"].copy_(de_id, non_blocking=True)
        ent["a"].copy_(alpha, non_blocking=True)
        ent["plan"].replay()
        st.logs = ent["logs"]
        return ent["out"]


def plan_default() -> bool:
    """RCOT_PLAN (default 1): iterate through recorded launch plans; RCOT_GRAPH=1 selects HIP-graph replay instead."""
    return os.environ.get("RCOT_PLAN", "1") != "0" and os.environ.get("RCOT_GRAPH", "0") != "1"
