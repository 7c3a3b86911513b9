"""Host-side launch plans: the minimax iteration (reference loop body trainer.py:247-346) as a recorded command list.

One iteration is ~2 700 kernel launches whose shapes, pointers and scalar arguments do not change from one iteration to the
next, while walking the Python schedule (views, shape checks, dispatch decisions, ~30 allocations per block) costs ~18 us
per launch: ~50 ms of host time per iteration — as much as the kernels of the three small levels of the network need, so
any kernel gain there would disappear behind the host.  HIP-graph replay removes the host but adds ~2 us of GPU time to every
node on ROCm 7.2 (68 vs 63 ms for the transport-map unit in round 3, 75.9 vs 68.9 ms in round 4): it was removed in round 5.

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

import ctypes
import gc
import os
from typing import Callable, List, Optional

import torch

from . import lib as _lib

_HOST_ONLY = {"rcot_abi_version", "rcot_debug_nt_coop", "rcot_ln_bwd_rows", "rcot_last_kernel", "rcot_kmajor_desc_size", "rcot_profile_begin", "rcot_profile_end"}          # no launch, no stream argument


class _RecordingLib:
    """Stands in for the ctypes library handle of a HipBackend while a plan is recorded."""

    def __init__(self, real, cmds: list, side_handle, symbols: Optional[list] = None):
        self._real, self._cmds, self._side, self._syms = real, cmds, side_handle, symbols
        self._kbuf = ctypes.create_string_buffer(192)
        self._kname = getattr(real, "rcot_last_kernel", None)

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if name in _HOST_ONLY or not name.startswith("rcot_"):
            return fn
        cmds, side, syms, kname, kbuf = self._cmds, self._side, self._syms, self._kname, self._kbuf

        def rec(*a):
            seq0 = kname(kbuf, 192) if (syms is not None and kname is not None) else 0
            rc = fn(*a)
            if rc == 0:                          # (EUNSUPPORTED launched nothing: the caller takes another route)
                # every entry point takes the stream LAST: replay supplies the current stream, or the backend's side stream for
                # what was launched there (weight gradients next to the data-gradient chain)
                cmds.append((fn, a[:-1], side is not None and a[-1] == side))
                if syms is not None:             # the kernel symbol the dispatcher chose (rcot_last_kernel), else the entry point's name
                    noted = kname is not None and kname(kbuf, 192) != seq0
                    syms.append((len(cmds) - 1, kbuf.value.decode() if noted else name))
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
        self._pins = ()                          # padded-plane cache entries this plan's recorded arguments point into
        self.symbols: List[tuple] = []           # (index into cmds, kernel symbol) of every recorded launch (measurement aid)

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
        # No finalizer may run inside the pool context below: a torch.cuda.MemPool that dies there (an earlier MinimaxStep's plans
        # left as cyclic garbage) empties its cache in its destructor, which asserts that no thread allocates into a pool —
        # std::terminate, "Fatal Python error: Aborted" in whatever line the collector happened to fire (seen in round 5 in
        # tests/test_plan_gpu.py under `pytest -q`, never under `-v`: collection timing).  So: collect NOW, outside the context,
        # and keep the cyclic collector off while the schedule is recorded (reference counting still frees everything acyclic).
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        self.pool = torch.cuda.MemPool()
        # the backend's side stream stays in use while recording: its launches are marked, and the cross-stream waits / events of
        # the schedule (HipBackend._host) are kept as host actions at their positions
        self._side = be._side.cuda_stream if getattr(be, "_side", None) is not None else None
        be.L = _RecordingLib(real, self.cmds, self._side, self.symbols)
        be._plan = self
        pinning, touched = be.pcm_pinning, be._pcm_touched
        be.pcm_pinning, be._pcm_touched = True, set()
        ok = False
        try:
            with torch.cuda.use_mem_pool(self.pool):
                body()
                be.side_join()                   # the plan ends joined: a replay starts from the state the recording started from
            ok = True
        finally:
            if gc_was_on:
                gc.enable()
            be.L, be._plan = real, None
            mine, be._pcm_touched, be.pcm_pinning = be._pcm_touched, touched, pinning
            if ok:
                self._pins = tuple(mine)
                for k in mine:
                    be._pcm_pinned[k] = be._pcm_pinned.get(k, 0) + 1
            else:
                # a schedule that raised half-way: join the side stream with the real library handle back in place, forget what
                # was held for it, and give the half-filled pool back (nothing recorded will ever be replayed)
                try:
                    be.side_join()
                    torch.cuda.synchronize()
                finally:
                    be._held.clear()
                    self.cmds.clear()
                    self.keep.clear()
                    self.pool = None
        return self

    def release(self):
        """Drop the plan: its pool (the iteration's working set) goes back to the allocator and the padded-plane cache entries it
        pinned may be evicted again.  The caller must have synchronised with the last replay."""
        be = self.be
        for k in self._pins:
            n = be._pcm_pinned.get(k, 0) - 1
            if n > 0:
                be._pcm_pinned[k] = n
            else:
                be._pcm_pinned.pop(k, None)
        self._pins = ()
        self.cmds.clear()
        self.symbols.clear()
        self._bound.clear()
        self.keep.clear()
        self.pool = None

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

    def set_scalar(self, entry_point: str, argpos: int, value, where=None):
        """Replace argument ``argpos`` of every recorded call of ``entry_point`` (those for which ``where(args)`` holds) by ``value``:
        a by-value scalar that changes BETWEEN replays without changing the schedule (the learning rate of the fused optimizer
        launches: trainer.py:228-243 decays it ten times over a default run).  Returns the number of calls changed."""
        n = 0
        for i, (fn, a, on_side) in enumerate(self.cmds):
            if a is None or getattr(fn, "__name__", "") != entry_point or (where is not None and not where(a)):
                continue
            self.cmds[i] = (fn, a[:argpos] + (value,) + a[argpos + 1:], on_side)
            n += 1
        if n:
            self._bound.clear()                  # bound argument tuples are rebuilt at the next replay
        return n

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


def time_symbol(plan: LaunchPlan, symbol: str, reps: int = 5):
    """Kernel time per step of ONE kernel symbol without launch brackets: the plan's recorded launches of that symbol (those of the
    side stream included, here on the launch stream), re-issued back to back in recorded order, ``reps`` cycles between two HIP
    events; returns (ms per cycle, launches per cycle).  What it contains beyond the kernels' own durations (the figure rocprofv3 lists) is the
    dispatch gap between two dependent-by-stream-order launches (1-2 us); what it lacks is the in-situ neighbourhood (operands a
    neighbour just left in the Infinity Cache).  The launches overwrite activations / accumulate into gradient buffers of the
    plan's pool: call it only between iterations (every iteration starts from zero_grad and recomputes its activations)."""
    st = plan.be._st()
    idx = [i for i, sym in plan.symbols if sym == symbol]
    calls = []
    for i in idx:
        fn, a, _on_side = plan.cmds[i]
        full = a + (st,)
        calls.append((fn, tuple(t(v) if type(v) in (int, float) else v for t, v in zip(fn.argtypes, full))))
    if not calls:
        return None, 0
    for fn, a in calls:                          # one untimed cycle
        fn(*a)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        for fn, a in calls:
            fn(*a)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps, len(calls)


class PlannedMinimax:
    """Drop-in for ``MinimaxStep.iteration``: same arguments and return value, a LaunchPlan per configuration underneath."""

    def __init__(self, step, warmup: int = 1):
        self.step = step
        self.cache = {}                          # key -> entry; dict order = least recently used first
        self.max_plans = max(1, int(os.environ.get("RCOT_PLAN_CACHE", "6")))
        self.recordings = 0                      # plans recorded so far (a training run records one per batch shape / branch, not per step)
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
        # (the learning rates are NOT part of the key: they are by-value arguments of the three optimizer launches and are patched into
        # the recorded calls when they change — _set_lr; rounds 4-5 keyed on them and re-recorded the iteration at every decay step)
        return (tuple(degraded.shape), bool(paired), bool(st._any_spectral), int(st.be.prec), st.grad_probe is not None)

    def _lrs(self):
        st = self.step
        return (float(st.To.param_groups[0]["lr"]), float(st.Fo.param_groups[0]["lr"]))

    def _set_lr(self, ent):
        """bring the optimizer launches of a recorded iteration to the optimizers' current learning rates"""
        lrs = self._lrs()
        if ent["lr"] == lrs:
            return
        st = self.step
        pT, pF = st.T.store.flat.data_ptr(), st.F.store.flat.data_ptr()
        nT = ent["plan"].set_scalar("rcot_rmsprop_step", 4, lrs[0], where=lambda a: a[0] == pT)
        nF = ent["plan"].set_scalar("rcot_rmsprop_step", 4, lrs[1], where=lambda a: a[0] == pF)
        if nT != ent["n_opt"][0] or nF != ent["n_opt"][1]:
            raise RuntimeError(f"launch plan: {nT} + {nF} optimizer launches found, {ent['n_opt']} recorded")
        ent["lr"] = lrs

    def _sync_packs(self):
        """A recorded iteration starts with kernels that read the weight packs of ITS arithmetic and ends with the launch that
        refreshes them.  When the previous iteration ran in another arithmetic (``be.prec`` switched between calls) the packs this
        one reads predate that iteration's optimizer steps: refresh them now, before the replay (eagerly, outside any plan)."""
        st = self.step
        for net in (st.T, st.F):
            if not hasattr(net, "repack"):
                continue
            if getattr(net, "_packed_prec", st.be.prec) != st.be.prec or (getattr(net, "_stale", False) and st.be.prec == _lib.PREC_BF16X3):
                net.repack()

    def _evict(self, new_key):
        """Each entry owns a private memory pool with the whole iteration's working set: the cache is capped (RCOT_PLAN_CACHE, at least
        one), least recently used first.  A dropped plan is released after a device synchronisation (its last replay may still run)."""
        dead = []
        while len(self.cache) - len(dead) >= self.max_plans:
            dead.append(next(k for k in self.cache if k not in dead))
        if dead:
            torch.cuda.synchronize()             # the last replay of a dropped plan may still be running
            for k in dead:
                self.cache.pop(k)["plan"].release()
            gc.collect()                         # (their pools die here, not at a moment of the collector's choosing: LaunchPlan.record)
            torch.cuda.empty_cache()

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
        self.recordings += 1
        if self.recordings in (16, 64, 256):     # a healthy run records a handful of plans; say so when something re-records all the time
            import warnings
            warnings.warn(f"rcot_amd.plan: {self.recordings} launch plans recorded so far — the iteration's configuration (batch shape, "
                          "paired / spectral branch, arithmetic) keeps changing, or RCOT_PLAN_CACHE is too small; every recording costs "
                          "an eager iteration plus a device synchronisation (RCOT_PLAN=0 walks the schedule instead)")
        pT, pF = st.T.store.flat.data_ptr(), st.F.store.flat.data_ptr()
        n_opt = tuple(sum(1 for fn, a_, _s in plan.cmds if a_ is not None and getattr(fn, "__name__", "") == "rcot_rmsprop_step" and a_[0] == q)
                      for q in (pT, pF))
        return dict(plan=plan, x=x, y=y, d=d, a=a, out=box["out"], logs=dict(st.logs), lr=self._lrs(), n_opt=n_opt)

    def iteration(self, degraded, target, de_id, alpha, paired: bool):
        st = self.step
        if not self.enabled:
            return st.iteration(degraded, target, de_id, alpha, paired)
        self._sync_packs()
        key = self._key(degraded, paired)
        ent = self.cache.pop(key, None)
        if ent is None:
            self._evict(key)
            ent = self.cache[key] = self._prepare(degraded, target, de_id, alpha, paired)     # (recording ran the iteration)
            return ent["out"]
        self.cache[key] = ent                    # most recently used last
        self._set_lr(ent)
        ent["x"].copy_(degraded, non_blocking=True)
        ent["y"].copy_(target, non_blocking=True)
        ent["d"].copy_(de_id, non_blocking=True)
        ent["a"].copy_(alpha, non_blocking=True)
        ent["plan"].replay()
        st.logs = ent["logs"]
        return ent["out"]


def plan_default() -> bool:
    """RCOT_PLAN (default 1): iterate through recorded launch plans; 0 walks the Python schedule every iteration."""
    return os.environ.get("RCOT_PLAN", "1") != "0"
