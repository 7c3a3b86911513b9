import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import relerr, seeded_tensor
from rcot_amd import lib
from rcot_amd.ops import HipBackend
be = HipBackend()
T = lambda s, *sh, scale=1.0: seeded_tensor(s, sh, scale=scale)
for (B, Ci, Co, N) in [(8, 96, 288, 16384), (2, 96, 510, 4096), (4, 192, 510, 1024)]:
    W, dY, X = T(1, Co, Ci, scale=0.1).cuda(), T(2, B, Co, N).cuda(), T(3, B, Ci, N).cuda()
    ref_dX = torch.einsum('oc,bon->bcn', W.double(), dY.double())
    ref_dW = torch.einsum('bon,bcn->oc', dY.double(), X.double())
    for prec in (lib.PREC_FP32, lib.PREC_BF16X3):
        be.prec = prec
        dX = torch.zeros(B, Ci, N, device='cuda'); dW = torch.zeros(Co, Ci, device='cuda')
        be.conv1x1_dgrad(W, dY, dX)
        torch.cuda.synchronize()
        e1 = relerr(dX, ref_dX)
        be.conv1x1_wgrad(dY, X, dW, beta=0.0)
        torch.cuda.synchronize()
        e2 = relerr(dW, ref_dW)
        e3 = relerr(dX, ref_dX)
        mu, rs = torch.zeros(B, N, device='cuda'), torch.zeros(B, N, device='cuda')
        be.ln_stats(X, mu, rs)
        lw, lb = (1 + 0.1 * T(4, Ci)).cuda(), (0.1 * T(5, Ci)).cuda()
        dW2 = torch.zeros(Co, Ci, device='cuda')
        be.conv1x1_wgrad(dY, X, dW2, ln=(mu, rs, lw, lb), beta=0.0)
        torch.cuda.synchronize()
        xn = ((X.double() - mu.double()[:, None]) * rs.double()[:, None] * lw.double()[None, :, None] + lb.double()[None, :, None])
        e4 = relerr(dW2, torch.einsum('bon,bcn->oc', dY.double(), xn))
        print((B, Ci, Co, N), 'prec', prec, 'dgrad', e1, 'wgrad', e2, 'dgrad after wgrad', e3, 'wgrad+ln', e4, flush=True)
