"""Bit-identity and timing of the implicit-GEMM convolution engine between two builds of librcot_hip.so.
  RCOT_LIB=<lib A> python scripts/cmp_conv_engine.py dump /tmp/a.pt ; RCOT_LIB=<lib B> python scripts/cmp_conv_engine.py dump /tmp/b.pt
  python scripts/cmp_conv_engine.py cmp /tmp/a.pt /tmp/b.pt
dump: forward (bias + LeakyReLU), data gradient and weight gradient of every critic layer (Net_Restormer.py:443-487) at B = 16 / 8 / 2,
128x128, and of the transport map's 3x3 convolutions (plain, PixelUnshuffle / PixelShuffle stores); prints the time of the critic's
forward sweep (10 layers, B = 16: the 2B batch of the critic step) from a recorded launch plan."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch


def dump(path):
    from rcot_amd import params as P
    from rcot_amd.ops import HipBackend
    from rcot_amd.plan import LaunchPlan
    be = HipBackend()
    out = {}
    for B in (16, 8, 2):
        H = 128
        for li, (ci, co, k, s, pad, bias) in enumerate(P.FNET_CONVS):
            Ho = H // s
            g = torch.Generator(device="cuda").manual_seed(100 + li)
            X = torch.randn(B, ci, H, H, device="cuda", generator=g)
            Wt = torch.randn(co, ci, k, k, device="cuda", generator=g) * 0.02
            bv = torch.randn(co, device="cuda", generator=g) if bias else None
            Y = torch.empty(B, co, Ho, Ho, device="cuda")
            dZ = torch.randn(B, co, Ho, Ho, device="cuda", generator=g)
            dX, dW = torch.empty_like(X), torch.zeros_like(Wt)
            be.conv2d_fwd(X, Wt, bv, Y, s, pad, 0.2, 0, None)
            be.conv2d_dgrad(dZ, Wt, dX, s, pad, 0.0)
            be.conv2d_wgrad(dZ, X, dW, s, pad, 1.0)
            out[("F", B, li)] = (Y.cpu(), dX.cpu(), dW.cpu())
            H = Ho
    for (ci, co, H, cmap) in ((3, 48, 128, 0), (48, 24, 128, 1), (96, 48, 64, 1), (384, 768, 16, 2), (192, 384, 32, 2), (96, 3, 128, 0)):
        g = torch.Generator(device="cuda").manual_seed(500 + ci)
        X = torch.randn(2, ci, H, H, device="cuda", generator=g)
        Wt = torch.randn(co, ci, 3, 3, device="cuda", generator=g) * 0.05
        shp = (2, 4 * co, H // 2, H // 2) if cmap == 1 else ((2, co // 4, 2 * H, 2 * H) if cmap == 2 else (2, co, H, H))
        Y = torch.empty(*shp, device="cuda")
        be.conv2d_fwd(X, Wt, None, Y, 1, 1, 1.0, cmap, None)
        dZ = torch.randn(2, co, H, H, device="cuda", generator=g)
        dX, dW = torch.empty_like(X), torch.zeros_like(Wt)
        be.conv2d_dgrad(dZ, Wt, dX, 1, 1, 0.0)
        be.conv2d_wgrad(dZ, X, dW, 1, 1, 1.0)
        out[("T", ci, co, H, cmap)] = (Y.cpu(), dX.cpu(), dW.cpu())
    torch.save(out, path)
    # the critic's forward / data-gradient / weight-gradient sweeps at B = 16
    acts = [torch.randn(16, 3, 128, 128, device="cuda")]
    ws = []
    H = 128
    for (ci, co, k, s, pad, bias) in P.FNET_CONVS:
        ws.append((torch.randn(co, ci, k, k, device="cuda") * 0.02, torch.zeros(co, device="cuda") if bias else None, s, pad))
        H //= s
        acts.append(torch.empty(16, co, H, H, device="cuda"))
    grads = [torch.randn_like(a) for a in acts]
    dws = [torch.zeros_like(w[0]) for w in ws]

    def fwd():
        for i, (Wt, bv, s, pad) in enumerate(ws):
            be.conv2d_fwd(acts[i], Wt, bv, acts[i + 1], s, pad, 0.2, 0, None)

    def dgrad():
        for i in reversed(range(1, len(ws))):
            be.conv2d_dgrad(grads[i + 1], ws[i][0], grads[i], ws[i][2], ws[i][3], 0.0)

    def wgrad():
        for i, (Wt, bv, s, pad) in enumerate(ws):
            be.conv2d_wgrad(grads[i + 1], acts[i], dws[i], s, pad, 1.0)
    res = []
    for fn in (fwd, dgrad, wgrad):
        fn()
        torch.cuda.synchronize()
        pl = LaunchPlan(be).record(fn)
        for _ in range(3):
            pl.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            pl.replay()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 20 * 1e6)
    print(f"{os.path.basename(os.environ.get('RCOT_LIB', 'librcot_hip.so'))}: critic sweeps at B=16 (10 layers): forward {res[0]:.0f} us, "
          f"data gradient {res[1]:.0f} us, weight gradient {res[2]:.0f} us")


def cmp(a, b):
    A, Bd = torch.load(a), torch.load(b)
    bad = 0
    for k in A:
        for i, nm in enumerate(("fwd", "dgrad", "wgrad")):
            if not torch.equal(A[k][i], Bd[k][i]):
                bad += 1
                print("DIFFERS", k, nm, float((A[k][i] - Bd[k][i]).abs().max()))
    print(f"{3 * len(A)} outputs compared, {bad} differ")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    dump(sys.argv[2]) if sys.argv[1] == "dump" else cmp(sys.argv[2], sys.argv[3])
