"""torch.autograd front end of the explicit schedules: ``nn.Module`` wrappers of ``T_net`` / ``F_net`` on which the reference's
OWN loop body runs unchanged (trainer.py:262-346: ``freeze`` / ``unfreeze``, ``Fnet(x).squeeze()``, ``loss.backward()``,
``torch.optim`` steps, and the gradient penalty's ``torch.autograd.grad(..., create_graph=True)`` at :291-298).

This is the compatibility layer of INTEGRATION.md level 0, not the measured path: ``rcot_amd.trainer.MinimaxStep`` drives the
same kernels without autograd, with fused optimizers and no gradient copies.  Here

* every parameter is an ``nn.Parameter`` that ALIASES its view of the network's flat buffer, so ``torch.optim`` updates land
  where the kernels read them; the K-major weight packs are refreshed at the next forward (the flat buffer's version counter
  tells);
* ``forward`` / ``backward`` are one ``torch.autograd.Function`` per network whose backward runs the explicit backward
  schedule into the (zeroed) flat gradient buffer and hands autograd per-parameter copies — so accumulation, ``zero_grad``
  (to zero or to None), frozen parameters (``requires_grad = False`` -> no weight-gradient kernels) and several applications
  of one network in one graph (``F(target)`` and ``F(fake)``, :266-276) behave as with the reference's modules;
* the double backward of the gradient penalty is a second Function: called with ``create_graph=True`` the critic's backward
  returns dF/dx as a node whose own backward is the linearised sweep ``F_net.gp_param_gradients``.  Second
  derivatives through the LeakyReLU masks are zero almost everywhere, as autograd gives for the reference.

The kernel layer is whatever backend the wrapped network was built on (HIP on the GPU; the fp64 test double on the CPU tier,
where tests/test_autograd_cpu.py runs the loop body against the oracle)."""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import nn

from . import params as P


def _param_grads(net, names: List[str], needs, none_for=()):
    """copies of the per-parameter views of the flat gradient buffer (autograd may keep what it is handed)"""
    out = []
    for n, need in zip(names, needs):
        out.append(net.store.g[n].clone() if (need and n not in none_for and n in net.store.g) else None)
    return out


class _Wrapped(nn.Module):
    """common part: parameters aliasing the flat buffer, reference state_dict, pack refresh"""

    def __init__(self, net):
        super().__init__()
        object.__setattr__(self, "net", net)                       # not a sub-module
        self._names = [n for n, _ in net.store.shapes]
        self.flat_params = nn.ParameterList([nn.Parameter(net.store.p[n]) for n in self._names])
        self._packed_version = None

    # the reference's names, shapes and order (Net_Restormer.py state_dict), not the ParameterList's
    def state_dict(self, *a, **k):
        return self.net.state_dict()

    def load_state_dict(self, sd, strict=True):
        self.net.load_state_dict(sd, strict)
        self._packed_version = None

    def _sync_packs(self):
        v = self.net.store.flat._version
        if v != self._packed_version:
            if hasattr(self.net, "repack"):
                self.net.repack()
            self._packed_version = self.net.store.flat._version


class _TApply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, save, x, *params):
        net = mod.net
        out = net.forward(x.detach(), save=save)
        ctx.mod, ctx.saved = mod, (net._ctx if save else None)
        net._ctx = None
        return out

    @staticmethod
    def backward(ctx, dout):
        mod, net = ctx.mod, ctx.mod.net
        assert ctx.saved is not None, "T_net forward ran without saving activations"
        net._ctx = ctx.saved
        net.zero_grad()
        net.backward(dout.contiguous())
        dead = [n for n in mod._names if P.tnet_is_dead(n)]         # never used upstream: autograd leaves them at None
        return (None, None, None, *_param_grads(net, mod._names, ctx.needs_input_grad[3:], none_for=dead))


class _FInputGrad(torch.autograd.Function):
    """dx = d<gout, F(x)>/dx as a function of the parameters (the node autograd differentiates for the gradient penalty)"""

    @staticmethod
    def forward(ctx, mod, saved, gout, *params):
        net = mod.net
        net._ctx = saved
        gx, lin = net.gp_input_gradient(gout.detach().contiguous())
        net._ctx = None
        ctx.mod, ctx.lin = mod, lin
        return gx

    @staticmethod
    def backward(ctx, u):
        mod, net = ctx.mod, ctx.mod.net
        net.zero_grad()
        net.gp_param_gradients(u.contiguous(), ctx.lin)
        return (None, None, None, *_param_grads(net, mod._names, ctx.needs_input_grad[3:], none_for=("fc2.bias",)))


class _FApply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, save, x, *params):
        net = mod.net
        out = net.forward(x.detach(), save=save)
        ctx.mod, ctx.saved, ctx.params = mod, (net._ctx if save else None), params
        net._ctx = None
        return out

    @staticmethod
    def backward(ctx, dout):
        mod, net = ctx.mod, ctx.mod.net
        assert ctx.saved is not None, "F_net forward ran without saving activations"
        if torch.is_grad_enabled():
            # create_graph=True (trainer.py:291-298): dF/dx must itself be differentiable w.r.t. the parameters.  Only the INPUT
            # gradient is produced on this route — what the reference's gradient penalty asks for (autograd.grad(inputs=interpolates));
            # ctx.needs_input_grad is fixed at forward time (requires_grad of the inputs), so a caller that additionally wanted
            # first-order PARAMETER gradients with create_graph cannot be told apart here and receives None for them: use a
            # plain backward() for those (documented in INTEGRATION.md, level 0).
            dx = _FInputGrad.apply(mod, ctx.saved, dout, *ctx.params)
            return (None, None, dx, *([None] * len(ctx.params)))
        wgrad = any(ctx.needs_input_grad[3:])
        need_dx = ctx.needs_input_grad[2]
        net._ctx = ctx.saved
        net.zero_grad()
        dx = net.backward(dout.contiguous(), wgrad=wgrad, need_dx=need_dx)
        net._ctx = None
        grads = _param_grads(net, mod._names, ctx.needs_input_grad[3:]) if wgrad else [None] * len(ctx.params)
        return (None, None, dx if need_dx else None, *grads)


class TNetModule(_Wrapped):
    """``Net_Restormer.T_net`` call contract (``Tnet(degraded)`` -> restored image) with autograd, over rcot_amd's T_net"""

    def forward(self, inp_img, noise_emb=None):
        self._sync_packs()
        save = torch.is_grad_enabled() and any(p.requires_grad for p in self.flat_params)     # (grad mode is off inside Function.forward)
        return _TApply.apply(self, save, inp_img, *self.flat_params)


class FNetModule(_Wrapped):
    """``Net_Restormer.F_net`` call contract (``Fnet(x)`` -> [B]) with autograd incl. ``create_graph`` on the input"""

    def forward(self, x):
        self._sync_packs()
        save = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.flat_params))
        return _FApply.apply(self, save, x, *self.flat_params)


def as_modules(Tnet, Fnet):
    """(TNetModule, FNetModule) over two rcot_amd networks — what the reference's trainer.py:92-93 would construct"""
    return TNetModule(Tnet), FNetModule(Fnet)
