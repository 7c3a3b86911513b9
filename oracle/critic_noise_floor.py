"""TEST INFRASTRUCTURE (build container only: imports the reference from /root/reference).  How far are the REFERENCE's own fp32 critic-loss
gradients (trainer.py:268-280: -mean F(y) + mean F(T(x)), backward) from the same modules evaluated in fp64, at BASELINE configs[1]'s full
batch (B = 8, 128x128, the seeded parameters and batch of the `cfg2b8` case of tests/golden/iter_grads.npz)?  The two halves of that loss
nearly cancel at initialisation, so the fixture carries 2e-5 .. 3.5e-5 (1 - cos) of fp32 rounding in features.4 / features.6: the direction
bar of tests/test_iteration_grads_gpu.py for that half-step is set from this measurement.
  python oracle/critic_noise_floor.py"""
import sys, os, types, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import importlib.util
spec = importlib.util.spec_from_file_location("pin", "/root/repo/oracle/pin_against_reference.py"); pin = importlib.util.module_from_spec(spec); spec.loader.exec_module(pin)
pin._stub_modules(); sys.path.insert(0, '/root/reference'); os.chdir('/tmp'); os.makedirs('/tmp/checksample/pin', exist_ok=True)
torch.Tensor.cuda = lambda self, *a, **k: self
import Net_Restormer as NR
from rcot_amd import params as P
from rcot_amd.synth import make_batch
torch.set_num_threads(8)
B, ps, de = 8, 128, [2]*8
to_t = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
Tn, Fn = NR.T_net(decoder=True), NR.F_net(patch_size=ps)
Tn.load_state_dict(to_t(P.seeded_params(P.tnet_param_shapes(), 31, "T"))); Fn.load_state_dict(to_t(P.seeded_params(P.fnet_param_shapes(ps), 32, "F")))
_, deg, clean = make_batch(77, B, ps, de)
with torch.no_grad(): fake = Tn(deg)
def grads(F, fake, clean):
    F.zero_grad()
    loss = -F(clean).mean() + F(fake).mean()
    loss.backward()
    return {k: v.grad.detach().double().clone() for k, v in F.named_parameters() if v.grad is not None}, float(loss)
g32, l32 = grads(Fn, fake, clean)
F64 = NR.F_net(patch_size=ps).double(); F64.load_state_dict({k: v.double() for k, v in Fn.state_dict().items()})
g64, l64 = grads(F64, fake.double(), clean.double())
print("loss", l32, l64)
for k in g32:
    a, b = g32[k].flatten(), g64[k].flatten()
    cos = float((a*b).sum()/max(float(a.norm()*b.norm()), 1e-300))
    if float(b.norm()) > 0:
        print(f"{k:22s} norm rel err {abs(float(a.norm()-b.norm()))/float(b.norm()):.2e}  1-cos {1-cos:.2e}")

# ---- second half-step: the gradient penalty (trainer.py:283-308) is evaluated AFTER the critic's first RMSprop step, which is sign-like
# (g / sqrt(0.01 g^2) = +-10): step the fp32 critic once with its fp32 gradients and once with the fp64 ones (differences = signs of
# gradients that are rounding noise), then evaluate the reference's GP gradients (fp32, double backward) on both
def stepped(g):
    F = NR.F_net(patch_size=ps)
    F.load_state_dict(Fn.state_dict())
    with torch.no_grad():
        for k, v in F.named_parameters():
            if k in g:
                gg = g[k].float()
                v -= 1e-4 * gg / (torch.sqrt(0.01 * gg * gg) + 1e-8)
    return F
def gp_grads(F):
    alpha = pin.seeded_tensor(78, (B,), lo=0.0, hi=1.0).view(B, 1, 1, 1)
    inter = (alpha * clean + (1 - alpha) * fake).requires_grad_(True)
    out = F(inter)
    g = torch.autograd.grad(outputs=out, inputs=inter, grad_outputs=torch.ones_like(out), create_graph=True, retain_graph=True, only_inputs=True)[0]
    gp = 10 * ((g.view(B, -1).norm(2, dim=1) - 1) ** 2).mean()
    F.zero_grad(); gp.backward()
    return {k: v.grad.detach().double().clone() for k, v in F.named_parameters() if v.grad is not None}, float(gp)
ga, gpa = gp_grads(stepped(g32)); gb, gpb = gp_grads(stepped(g64))
print("gp", gpa, gpb)
for k in ga:
    a, b = ga[k].flatten(), gb[k].flatten()
    if float(b.norm()) > 0:
        print(f"GP {k:22s} norm rel diff {abs(float(a.norm()-b.norm()))/float(b.norm()):.2e}  1-cos {1-float((a*b).sum()/(a.norm()*b.norm())):.2e}")
