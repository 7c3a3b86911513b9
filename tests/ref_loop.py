"""The reference's loop body (trainer.py:262-346) restated with the calls it makes on MODULE objects — freeze / unfreeze,
``Fnet(x).squeeze()``, ``loss.backward()``, ``torch.optim`` steps, ``Module.zero_grad()`` and the gradient penalty's
``torch.autograd.grad(..., create_graph=True)`` on the interpolates.  Used by the CPU and GPU tests of rcot_amd/autograd.py."""
import torch

from rcot_amd.trainer import freeze, unfreeze


def reference_style_iteration(Tnet, Fnet, T_optimizer, F_optimizer, degraded, target, de_id, alpha, sigma, Sigma, paired, probe=None, probes=None):
    """trainer.py:262-346 on module objects (alpha injected instead of drawn, :284)"""
    freeze(Tnet)
    unfreeze(Fnet)
    Fnet.zero_grad()                                                                   # :266
    real = Fnet(target).squeeze()
    out_restored = Tnet(degraded)                                                      # T is frozen: no graph
    fake = Fnet(out_restored.data).squeeze()
    F_train_loss = -real.mean() + fake.mean()                                          # :269-276
    F_train_loss.backward()
    if probes is not None:
        probes("F_critic")
    F_optimizer.step()                                                                 # :280
    Fnet.zero_grad()                                                                   # :283
    a = alpha.view(-1, 1, 1, 1).expand_as(target)
    interpolated = (a * target.data + (1 - a) * out_restored.data).requires_grad_(True)
    out = Fnet(interpolated).squeeze()
    grad = torch.autograd.grad(outputs=out, inputs=interpolated, grad_outputs=torch.ones(out.size(), dtype=out.dtype, device=out.device),
                               retain_graph=True, create_graph=True, only_inputs=True)[0]     # :291-298
    grad = grad.view(grad.size(0), -1)
    gp_loss = 10 * torch.mean((torch.sqrt(torch.sum(grad ** 2, dim=1)) - 1) ** 2)     # :300-305
    gp_loss.backward()
    if probe is not None:
        probe(Fnet)
    if probes is not None:
        probes("F_gp")
    F_optimizer.step()                                                                 # :308
    freeze(Fnet)
    unfreeze(Tnet)
    Fnet.zero_grad()
    Tnet.zero_grad()                                                                   # :311-315
    out_restored = Tnet(degraded)
    out_disc = Fnet(out_restored).squeeze()
    res = degraded - out_restored
    mse_loss = torch.mean(res ** 2) ** 0.5
    res_fre = torch.fft.fft2(res)
    penalty = 0
    for i in range(res_fre.shape[0]):                                                  # :325-332 (``**1/2`` is a division by two)
        penalty = penalty + (torch.mean(abs(res_fre[i]) ** 2) ** 1 / 2 if de_id[i] < 3 else torch.mean(abs(res_fre[i])))
    T_train_loss = -out_disc.mean() + sigma * (mse_loss + penalty)
    if paired:
        T_train_loss = T_train_loss + Sigma * torch.mean(abs(out_restored - target))  # :338-340
    T_train_loss.backward()
    if probes is not None:
        probes("T_gen")
    T_optimizer.step()                                                                 # :345-346
    return dict(Loss_F=float(F_train_loss), Loss_T=float(T_train_loss), Loss_mse=float(mse_loss), gp=float(gp_loss))
