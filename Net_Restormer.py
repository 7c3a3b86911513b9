"""Drop-in module name for checkpoint interchange: ``Net_Restormer.T_net`` / ``Net_Restormer.F_net``.

The reference pickles whole modules into its checkpoints (trainer.py:362-371) and its consumers unpickle them by
class path (tester.py:54, tester_noise.py:63, trainer.py:100-117).  With this file importable as ``Net_Restormer``
those consumers obtain the MI355X-native networks of ``rcot_amd`` (same constructor arguments, ``state_dict`` names and
call semantics; compute = librcot_hip.so).  Nothing is computed here; see rcot_amd/compat.py.

Reference-made checkpoints contain further classes of the reference's file (its block / norm / resampler modules).
They carry no behaviour we need — only parameters — so any such name is materialised on demand as an empty
``nn.Module`` container; ``rcot_amd.compat`` then flattens the tree into a state_dict.
"""
import torch as _torch

from rcot_amd.compat import F_net, T_net  # noqa: F401

_containers = {}


def __getattr__(name):          # PEP 562: called by pickle's find_class for names this module does not define
    if name.startswith("__"):
        raise AttributeError(name)
    if name not in _containers:
        _containers[name] = type(name, (_torch.nn.Module,), {"__module__": __name__})
    return _containers[name]
