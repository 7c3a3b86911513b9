"""Checkpoint interchange with the reference (trainer.py:362-371 saves, tester.py:54 / tester_noise.py:63 /
trainer.py:100-117 load).

The reference pickles WHOLE MODULES: ``torch.save({"epoch": e, "Tnet": Tnet, "Fnet": Fnet})`` and its consumers
either call the unpickled object (``torch.load(p)["Tnet"](x)``) or read ``.state_dict()`` from it.  The class path in
those pickles is ``Net_Restormer.T_net`` / ``Net_Restormer.F_net``.  This module provides picklable counterparts
of the HIP-backed networks under exactly those names (re-exported by the top-level ``Net_Restormer.py`` shim):

* our checkpoints unpickle, in an environment where the shim is importable as ``Net_Restormer``, to HIP-backed
  networks that can be called like the reference's modules (tester.py:54);
* the pickled state imitates a flat ``nn.Module`` (every tensor in ``_parameters`` under its state_dict name, no
  sub-modules), so even when the pickle is opened where ``Net_Restormer`` is the REFERENCE's file, the object is
  a reference ``T_net`` whose ``.state_dict()`` returns the 816 / 22 tensors — which is all the reference's resume path
  (trainer.py:105-106) uses;
* reference-made checkpoints (real module trees) unpickle here too: unknown ``Net_Restormer.*`` classes are
  manufactured as empty ``nn.Module`` containers by the shim, and ``__setstate__`` flattens the tree.

Construction of the HIP network is deferred until the object is first used for compute, so checkpoints can be
inspected / converted on a machine without a GPU.
"""
from __future__ import annotations

import copyreg
from collections import OrderedDict

import torch

from . import net_restormer as _nr


def _flatten_module_state(state) -> "OrderedDict[str, torch.Tensor]":
    """state = the ``__dict__`` of a pickled nn.Module (ours: flat; the reference's: a tree of sub-modules)."""
    sd = OrderedDict()
    for n, p in (state.get("_parameters") or {}).items():
        if p is not None:
            sd[n] = p.detach()
    for n, b in (state.get("_buffers") or {}).items():
        if b is not None:
            sd[n] = b.detach()
    for n, m in (state.get("_modules") or {}).items():
        if m is None:
            continue
        for k, v in m.state_dict().items():
            sd[n + "." + k] = v.detach()
    return sd


def _module_like_state(sd, extra):
    """A state dict dressed as the ``__dict__`` of a childless nn.Module (see the module docstring)."""
    st = dict(torch.nn.Module().__dict__)
    st["training"] = False
    st["_parameters"] = OrderedDict((k, torch.nn.Parameter(v.detach().cpu().clone(), requires_grad=True)) for k, v in sd.items())
    st.update(extra)
    return st


class _LazyNet:
    """Mixin: pickling + deferred construction for the HIP-backed networks."""

    _impl_cls = None            # rcot_amd.net_restormer.T_net / F_net
    _ctor_keys = ()

    def __init__(self, *a, **kw):
        object.__setattr__(self, "_lazy_sd", None)
        object.__setattr__(self, "_lazy_kw", None)
        self._impl_cls.__init__(self, *a, **kw)
        object.__setattr__(self, "_ctor", {k: getattr(self, k) for k in self._ctor_keys})

    # ---- construction without a GPU (converters, CPU tests)
    @classmethod
    def from_state_dict(cls, sd, **ctor):
        self = object.__new__(cls)
        object.__setattr__(self, "_lazy_sd", OrderedDict((k, v.detach().cpu().clone()) for k, v in sd.items()))
        object.__setattr__(self, "_lazy_kw", dict(ctor))
        object.__setattr__(self, "_ctor", dict(ctor))
        return self

    def _build(self):
        sd, kw = self._lazy_sd, self._lazy_kw or {}
        object.__setattr__(self, "_lazy_sd", None)
        self._impl_cls.__init__(self, **kw)
        self._impl_cls.load_state_dict(self, sd)

    def __getattr__(self, name):            # only reached when normal lookup fails
        if name.startswith("_lazy") or name == "_ctor":
            raise AttributeError(name)
        if self.__dict__.get("_lazy_sd") is not None:
            self._build()
            return getattr(self, name)
        raise AttributeError(name)

    def state_dict(self):
        if self.__dict__.get("_lazy_sd") is not None:
            return OrderedDict((k, v.clone()) for k, v in self._lazy_sd.items())
        return self._impl_cls.state_dict(self)

    def load_state_dict(self, sd, strict=True):
        if self.__dict__.get("_lazy_sd") is not None:
            self._build()
        return self._impl_cls.load_state_dict(self, sd, strict)

    # ---- pickling (class path Net_Restormer.<name>, nn.Module-shaped state)
    def __reduce_ex__(self, protocol):
        sd = self.state_dict()
        return (copyreg._reconstructor, (type(self), object, None),
                _module_like_state(sd, {"_rcot_ctor": dict(self.__dict__.get("_ctor") or {})}))

    def __setstate__(self, state):
        object.__setattr__(self, "_lazy_sd", OrderedDict((k, v.cpu().clone()) for k, v in _flatten_module_state(state).items()))
        kw = dict(state.get("_rcot_ctor") or {})
        if not kw:                              # a reference-made pickle: recover the constructor arguments from the module
            kw = self._ctor_from_reference_state(state)
        object.__setattr__(self, "_lazy_kw", kw)
        object.__setattr__(self, "_ctor", dict(kw))

    @staticmethod
    def _ctor_from_reference_state(state):
        return {}


class T_net(_LazyNet, _nr.T_net):
    """``Net_Restormer.T_net`` (reference Net_Restormer.py:215-434): HIP-backed, picklable."""
    __module__ = "Net_Restormer"            # the class path the reference's checkpoints use
    _impl_cls = _nr.T_net
    _ctor_keys = ("decoder",)

    @staticmethod
    def _ctor_from_reference_state(state):
        return {"decoder": bool(state.get("decoder", True))}          # the reference stores self.decoder (:229)


class F_net(_LazyNet, _nr.F_net):
    """``Net_Restormer.F_net`` (reference Net_Restormer.py:436-522): HIP-backed, picklable."""
    __module__ = "Net_Restormer"
    _impl_cls = _nr.F_net
    _ctor_keys = ("patch_size",)

    @staticmethod
    def _ctor_from_reference_state(state):
        fc = None
        for n, m in (state.get("_modules") or {}).items():
            if n == "fc":
                fc = m
        if fc is not None:                                              # fc: Linear(P*P/2 -> P*P/8), :494
            return {"patch_size": int(round((2 * fc.weight.shape[1]) ** 0.5))}
        return {}


def as_state_dict(obj):
    """state_dict of whatever a checkpoint holds under "Tnet"/"Fnet"/"model"/"discr": a plain dict (round-1 files),
    one of the classes above, or a reference module."""
    return obj if isinstance(obj, dict) else obj.state_dict()


def shim():
    """The top-level ``Net_Restormer`` shim of this repo, importable whatever the working directory is."""
    import importlib
    import os
    import sys
    if "Net_Restormer" in sys.modules:
        return sys.modules["Net_Restormer"]
    try:
        return importlib.import_module("Net_Restormer")
    except ImportError:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        return importlib.import_module("Net_Restormer")


def load_checkpoint(path):
    """torch.load of a checkpoint of either format.  Pure-tensor files (round-1 checkpoints, plain state_dicts) load
    with weights_only=True; module pickles (ours and the reference's) need the full unpickler, with
    ``Net_Restormer`` resolved by the shim — only open such files from sources you trust."""
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception:
        shim()
        return torch.load(path, map_location="cpu", weights_only=False)
