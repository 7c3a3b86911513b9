"""Pin the oracle against the real reference and (re)generate tests/golden/*.npz.

TEST INFRASTRUCTURE.  Runs ONLY in the build container, where /root/reference exists;
nothing of the reference travels (the fixtures hold inputs-by-seed and reference OUTPUTS).

    python oracle/pin_against_reference.py            # check + write fixtures + PINNED.md

What it does
  1. imports the reference's Net_Restormer.py / trainer.py with stub modules for the
     packages this image lacks (torchvision, skimage, cv2, lpips) and Tensor.cuda = identity
     (trainer.py:285,294 call .cuda() unconditionally);
  2. loads the repo's seeded parameters (rcot_amd.params.seeded_params) into the reference
     modules, runs reference and oracle on identical seeded inputs, asserts agreement;
  3. stores the REFERENCE outputs as golden fixtures.
"""
from __future__ import annotations

import argparse
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")

from rcot_amd import params as P          # noqa: E402
from oracle import rcot_oracle as O       # noqa: E402


def _stub_modules():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    captured = {}

    def save_image(t, path, *a, **k):
        captured[os.path.basename(str(path))] = t.detach().clone()
    tv = mod("torchvision")
    tv.utils = mod("torchvision.utils", save_image=save_image)
    ident = lambda *a, **k: (lambda x: x)
    tv.transforms = mod("torchvision.transforms", ToPILImage=ident, Compose=ident, RandomCrop=ident,
                        ToTensor=ident, Grayscale=ident)
    tv.models = mod("torchvision.models")
    sk = mod("skimage")
    sk.metrics = mod("skimage.metrics", peak_signal_noise_ratio=None, structural_similarity=None)
    mod("cv2")
    mod("lpips")
    return captured


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def seeded_tensor(seed, shape, scale=1.0, lo=None, hi=None):
    g = rng(seed)
    a = g.uniform(lo, hi, size=shape) if lo is not None else scale * g.standard_normal(shape)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def to_t(d):
    return {k: torch.from_numpy(v.copy()) for k, v in d.items()}


def relerr(a, b):
    a, b = a.double(), b.double()
    return float(((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).detach())


def strided(t, n=64):
    f = t.detach().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long()
    return f[idx].numpy().astype(np.float32)


def strided64(t, n):
    """strided samples with float64 index arithmetic (tensors of 2^28 elements: F_net(256).fc.weight)"""
    f = t.detach().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel()), dtype=torch.float64).long()
    return f[idx].numpy().astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-train", action="store_true")
    ap.add_argument("--only", default="", help="comma list of sections to (re)generate: blocks,convs,tnet,tnet128,fnet,"
                    "otcost,train,train128,traj,ckpt,itergrads,gpufx,init,mprnet,mprnetfx,data,blocks8 (default: the round-1 set blocks,convs,tnet,fnet,otcost,train)")
    args = ap.parse_args()
    only = set(args.only.split(",")) if args.only else {"blocks", "convs", "tnet", "fnet", "otcost", "train"}
    if args.skip_train:
        only.discard("train")
    captured = _stub_modules()
    sys.path.insert(0, REF)
    os.chdir("/tmp")
    os.makedirs("/tmp/checksample/pin", exist_ok=True)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    import Net_Restormer as NR
    torch.set_num_threads(8)
    report = []
    os.makedirs(GOLD, exist_ok=True)

    # ---------------------------------------------------------------- names / shapes
    refT = NR.T_net(decoder=True)
    ref_shapes = [(k, tuple(v.shape)) for k, v in refT.state_dict().items()]
    assert ref_shapes == P.tnet_param_shapes(), "T_net state_dict contract mismatch"
    for ps in (64, 128, 256):
        refF = NR.F_net(patch_size=ps)
        assert [(k, tuple(v.shape)) for k, v in refF.state_dict().items()] == P.fnet_param_shapes(ps)
    report.append("state_dict names/shapes/order: T_net 816 tensors, F_net(64/128/256) 22 tensors — identical")

    # ---------------------------------------------------------------- F1: blocks
    if "blocks" in only:
        blocks = [(48, 1, 16), (96, 2, 8), (96, 4, 8), (96, 1, 16), (192, 4, 8), (384, 8, 8), (384, 4, 8)]
        fx = {}
        for bi, (C, heads, HW) in enumerate(blocks):
            shapes = P.block_param_shapes("blk", C, heads)
            prm = to_t(P.seeded_params(shapes, 100 + bi, "T"))
            x = seeded_tensor(200 + bi, (2, C, HW, HW))
            gy = seeded_tensor(300 + bi, (2, C, HW, HW))
            m = NR.TransformerBlock(C, heads, 2.66, False, "WithBias")
            m.load_state_dict({k[len("blk."):]: v for k, v in prm.items()})
            xr = x.clone().requires_grad_(True)
            yr = m(xr)
            yr.backward(gy)
            po = {k: v.clone().requires_grad_(True) for k, v in prm.items()}
            xo = x.clone().requires_grad_(True)
            yo = O.transformer_block(xo, po, "blk", heads)
            yo.backward(gy)
            e = [relerr(yo, yr), relerr(xo.grad, xr.grad)]
            for k, v in m.named_parameters():
                e.append(relerr(po["blk." + k].grad, v.grad))
            assert max(e) < 2e-5, (C, heads, e)
            report.append(f"TransformerBlock C={C} heads={heads} {HW}x{HW}: oracle vs reference max rel err {max(e):.2e} (out, dx, 11 param grads)")
            tag = f"blk{bi}"
            fx[tag + "_cfg"] = np.array([C, heads, HW, 100 + bi, 200 + bi, 300 + bi])
            fx[tag + "_y"] = yr.detach().numpy()
            fx[tag + "_dx"] = xr.grad.numpy()
            for k, v in m.named_parameters():
                fx[tag + "_gn_" + k] = np.array(float(v.grad.double().norm()))
                fx[tag + "_gs_" + k] = strided(v.grad, 256)
        np.savez_compressed(os.path.join(GOLD, "blocks.npz"), **fx)

    # ---------------------------------------------------------------- F1b: blocks at the TRAINING batch and planes of the small levels
    # (round 5: B = 8 at 16x16 / 32x32 / 64x64 selects kernels that B = 2 fixtures never reach — the eight-wavefront k-group GEMM, the
    # merged dV/dQ/dK launch, the one-launch attention core, the paired launch; samples and norms instead of whole tensors keep it small)
    if "blocks8" in only:
        blocks = [(384, 8, 16), (384, 4, 16), (192, 4, 32), (96, 2, 64), (96, 1, 128), (48, 1, 128)]      # (the last two: level 1 at the training batch)
        fx = {}
        for bi, (C, heads, HW) in enumerate(blocks):
            shapes = P.block_param_shapes("blk", C, heads)
            prm = to_t(P.seeded_params(shapes, 500 + bi, "T"))
            x = seeded_tensor(600 + bi, (8, C, HW, HW))
            gy = seeded_tensor(700 + bi, (8, C, HW, HW))
            m = NR.TransformerBlock(C, heads, 2.66, False, "WithBias")
            m.load_state_dict({k[len("blk."):]: v for k, v in prm.items()})
            xr = x.clone().requires_grad_(True)
            yr = m(xr)
            yr.backward(gy)
            po = {k: v.clone().requires_grad_(True) for k, v in prm.items()}
            xo = x.clone().requires_grad_(True)
            yo = O.transformer_block(xo, po, "blk", heads)
            yo.backward(gy)
            e = [relerr(yo, yr), relerr(xo.grad, xr.grad)]
            for k, v in m.named_parameters():
                e.append(relerr(po["blk." + k].grad, v.grad))
            assert max(e) < 5e-5, (C, heads, e)
            report.append(f"TransformerBlock B=8 C={C} heads={heads} {HW}x{HW}: oracle vs reference max rel err {max(e):.2e} (out, dx, 11 param grads)")
            tag = f"b8blk{bi}"
            fx[tag + "_cfg"] = np.array([C, heads, HW, 500 + bi, 600 + bi, 700 + bi])
            fx[tag + "_y_s"] = strided(yr.detach(), 8192)
            fx[tag + "_y_n"] = np.array([float(yr.detach().double().norm()), float(yr.detach().abs().max())])
            fx[tag + "_dx_s"] = strided(xr.grad, 8192)
            fx[tag + "_dx_n"] = np.array([float(xr.grad.double().norm()), float(xr.grad.abs().max())])
            for k, v in m.named_parameters():
                fx[tag + "_gn_" + k] = np.array(float(v.grad.double().norm()))
                fx[tag + "_gs_" + k] = strided(v.grad, 256)
        np.savez_compressed(os.path.join(GOLD, "blocks_b8.npz"), **fx)

    # ---------------------------------------------------------------- resamplers / convs
    if "convs" in only:
        fx = {}
        for name, mod_, cin, seed in (("down", NR.Downsample(48), 48, 401), ("up", NR.Upsample(96), 96, 402),
                                      ("embed", NR.OverlapPatchEmbed(3, 48), 3, 403)):
            w = list(mod_.parameters())[0]
            wv = seeded_tensor(seed, tuple(w.shape), scale=0.1)
            w.data.copy_(wv)
            x = seeded_tensor(seed + 10, (2, cin, 16, 16)).requires_grad_(True)
            y = mod_(x)
            gy = seeded_tensor(seed + 20, tuple(y.shape))
            y.backward(gy)
            xo = x.detach().clone().requires_grad_(True)
            wo = wv.clone().requires_grad_(True)
            yo = {"down": O.downsample, "up": O.upsample,
                  "embed": lambda a, b: torch.nn.functional.conv2d(a, b, padding=1)}[name](xo, wo)
            yo.backward(gy)
            e = max(relerr(yo, y), relerr(xo.grad, x.grad), relerr(wo.grad, w.grad))
            assert e < 1e-5
            report.append(f"{name}: oracle vs reference max rel err {e:.2e}")
            fx[name + "_cfg"] = np.array([seed, cin, 16])
            fx[name + "_y"], fx[name + "_dx"], fx[name + "_dw"] = y.detach().numpy(), x.grad.numpy(), w.grad.numpy()
        np.savez_compressed(os.path.join(GOLD, "convs.npz"), **fx)

    # ---------------------------------------------------------------- F2: whole T_net
    if "tnet" in only:
        pT_np = P.seeded_params(P.tnet_param_shapes(), 11, "T")
        refT.load_state_dict(to_t(pT_np))
        fx = {}
        for tag, (B, HW, seed) in {"a": (1, 64, 501), "b": (2, 32, 502)}.items():
            x = seeded_tensor(seed, (B, 3, HW, HW), lo=0.0, hi=1.0)
            r = seeded_tensor(seed + 50, (B, 3, HW, HW))
            refT.zero_grad()
            y = refT(x)
            res_ref = captured["res.png"]
            (y * r).mean().backward()
            po = {k: v.clone().requires_grad_(True) for k, v in to_t(pT_np).items()}
            yo, reso = O.tnet_forward(po, x, True, return_res=True)
            (yo * r).mean().backward()
            e_out, e_res = relerr(yo, y), relerr(reso, res_ref)
            gn_ref, gn_or, dead = [], [], []
            for k, v in refT.named_parameters():
                if v.grad is None:
                    dead.append(k)
                    assert po[k].grad is None, k
                    gn_ref.append(-1.0)
                    continue
                gn_ref.append(float(v.grad.norm()))
                gn_or.append(relerr(po[k].grad, v.grad))
            assert sorted(dead) == sorted(n for n, _ in P.tnet_param_shapes() if P.tnet_is_dead(n)), dead
            assert e_out < 1e-4 and e_res < 1e-4 and max(gn_or) < 2e-3, (e_out, e_res, max(gn_or))
            report.append(f"T_net(decoder=True) B={B} {HW}x{HW}: out rel err {e_out:.2e}, pass-1 res {e_res:.2e}, "
                          f"worst param-grad rel err {max(gn_or):.2e}; 20 dead tensors grad None on both sides")
            fx[tag + "_cfg"] = np.array([B, HW, seed, 11])
            fx[tag + "_y"] = y.detach().numpy()
            fx[tag + "_res"] = res_ref.numpy()
            fx[tag + "_gradnorm"] = np.array(gn_ref, dtype=np.float64)
            for k in ("patch_embed.proj.weight", "output.weight", "latent.3.attn.temperature",
                      "refinement.1.ffn.project_in.weight", "resencoder_level2.0.attn.qkv.weight",
                      "down3_4.body.0.weight", "noise_level1.attn.qkv_dwconv.weight",
                      "decoder_level3.2.norm1.body.weight"):
                fx[tag + "_gs_" + k] = strided(dict(refT.named_parameters())[k].grad)
        np.savez_compressed(os.path.join(GOLD, "tnet.npz"), **fx)

    # ---------------------------------------------------------------- F3: F_net + GP
    if "fnet" in only:
        # Golden values are the reference evaluated in fp64 (module.double()): fp32 CPU runs of the critic are
        # knife-edge sensitive (one LeakyReLU mask of the 64-unit fc1 layer flipping moves fc.weight.grad by >1e-3),
        # so the seed is advanced until every fc1 pre-activation has a safe margin from zero.
        fx = {}
        for ps, seed0 in ((64, 601), (128, 602)):
            pF_np = P.seeded_params(P.fnet_param_shapes(ps), 21, "F")
            refF = NR.F_net(patch_size=ps).double()
            refF.load_state_dict({k: v.double() for k, v in to_t(pF_np).items()})
            seed = seed0
            while True:
                x = seeded_tensor(seed, (2, 3, ps, ps), lo=0.0, hi=1.0)
                with torch.no_grad():
                    z = refF.fc1(refF.fc(refF.features(x.double()).reshape(2, -1)))
                margin = float(z.abs().min() / z.abs().max())
                if margin > 2e-3:
                    break
                seed += 1000
            xr = x.double().requires_grad_(True)
            out = refF(xr)
            (g,) = torch.autograd.grad(out, xr, torch.ones_like(out), create_graph=True)
            gp = 10 * ((g.view(2, -1).pow(2).sum(1).sqrt() - 1) ** 2).mean()
            refF.zero_grad()
            gp.backward()
            gp_grads = {k: (None if v.grad is None else v.grad.clone()) for k, v in refF.named_parameters()}
            refF.zero_grad()
            (-refF(x.double()).mean()).backward()
            cr_grads = {k: v.grad.clone() for k, v in refF.named_parameters()}
            # fp32 oracle vs fp64 reference
            po = {k: v.clone().requires_grad_(True) for k, v in to_t(pF_np).items()}
            oo = O.fnet_forward(po, x)
            gpo = O.gradient_penalty(po, x)
            gpo_g = O._grads(gpo, po)
            pc = {k: v.clone().requires_grad_(True) for k, v in to_t(pF_np).items()}
            (-O.fnet_forward(pc, x).mean()).backward()
            e = [relerr(oo, out), abs(float(gpo) - float(gp)) / abs(float(gp)), max(relerr(pc[k].grad, cr_grads[k]) for k in pc)]
            for k in po:
                if gp_grads[k] is None:
                    assert gpo_g[k] is None
                elif float(gp_grads[k].abs().max()) == 0.0:
                    assert float(gpo_g[k].abs().max()) == 0.0, k
                else:
                    e.append(relerr(gpo_g[k], gp_grads[k]))
            assert max(e) < 2e-3, e
            zero_b = [k for k, v in gp_grads.items() if v is not None and float(v.abs().max()) == 0.0]
            none_b = [k for k, v in gp_grads.items() if v is None]
            report.append(f"F_net(patch={ps}) B=2 (input seed {seed}, fc1 mask margin {margin:.1e}): fp32 oracle vs fp64 reference, "
                          f"max rel err over out/gp/critic-grads/gp-grads {max(e):.2e}; GP grads exact-zero for {len(zero_b)} bias "
                          f"tensors, None for {none_b}")
            t = f"p{ps}"
            fx[t + "_cfg"] = np.array([ps, seed, 21])
            fx[t + "_out"], fx[t + "_dfdx"], fx[t + "_gp"] = out.detach().numpy(), g.detach().numpy().astype(np.float32), np.array(float(gp))
            fx[t + "_gp_gradnorm"] = np.array([-1.0 if v is None else float(v.norm()) for v in gp_grads.values()])
            fx[t + "_cr_gradnorm"] = np.array([float(v.norm()) for v in cr_grads.values()])
            for k in ("features.0.weight", "features.6.weight", "features.18.weight", "fc.weight", "fc1.weight", "fc2.weight"):
                fx[t + "_gp_gs_" + k] = strided(gp_grads[k])
                fx[t + "_cr_gs_" + k] = strided(cr_grads[k])
        np.savez_compressed(os.path.join(GOLD, "fnet.npz"), **fx)

    # ---------------------------------------------------------------- F4: OT cost (the trainer's inline code)
    if "otcost" in only:
        # reference expression evaluated verbatim-in-spirit via the imported torch ops of trainer.py:320-332
        fx = {}
        res = seeded_tensor(701, (4, 3, 32, 32), scale=0.2)
        res[1, 0] = 0.0                      # a plane whose spectrum is exactly zero (|F| = 0 branch)
        res[3, 1, :, :] = 0.25               # constant plane: a single non-zero bin
        de_id = [0, 2, 3, 7]
        rr = res.clone().requires_grad_(True)
        deg = torch.zeros_like(res)
        res_fre = torch.fft.fft2(deg - (-rr))
        pen = 0
        per = []
        for i in range(4):
            sl = res_fre[i, :]
            if de_id[i] < 3:
                t_ = torch.mean(abs(sl) ** 2) ** 1 / 2
            else:
                t_ = torch.mean(abs(sl))
            per.append(float(t_))
            pen = pen + t_
        mse_loss = (torch.mean(rr ** 2)) ** 0.5
        (mse_loss + pen).backward()
        ro = res.clone().requires_grad_(True)
        rm, fo = O.ot_cost(ro, torch.zeros_like(ro), de_id)
        (rm + fo).backward()
        e = max(abs(float(rm) - float(mse_loss)) / float(mse_loss), abs(float(fo) - float(pen)) / float(pen),
                relerr(ro.grad, rr.grad))
        assert e < 1e-5, e
        report.append(f"OT cost (rmse + Fourier penalty, de_id={de_id}): oracle vs trainer.py expression rel err {e:.2e}")
        fx["res"], fx["de_id"] = res.numpy(), np.array(de_id)
        fx["rmse"], fx["per_sample"], fx["dres"] = np.array(float(mse_loss)), np.array(per), rr.grad.numpy()
        np.savez_compressed(os.path.join(GOLD, "otcost.npz"), **fx)

    # ---------------------------------------------------------------- F5: verbatim trainer.train() iteration
    TR = None
    if only & {"train", "train128", "traj", "itergrads", "mprnet"}:
        sys.argv = ["trainer.py"]
        import trainer as TR

    def run_train_variant(fx, tag, B, ps, paired, de, opt_name):
        if True:
            pT_np = P.seeded_params(P.tnet_param_shapes(), 31, "T")
            pF_np = P.seeded_params(P.fnet_param_shapes(ps), 32, "F")
            Tn, Fn = NR.T_net(decoder=True), NR.F_net(patch_size=ps)
            Tn.load_state_dict(to_t(pT_np))
            Fn.load_state_dict(to_t(pF_np))
            lr = 1e-4
            TR.opt = Namespace(cuda=False, lr=lr, step=20, pairnum=(10 ** 7 if paired else 0), batchSize=B,
                               sigma=1.0, Sigma=10000.0, type="pin")
            mk = torch.optim.RMSprop if opt_name == "RMSprop" else torch.optim.Adam
            To, Fo = mk(Tn.parameters(), lr=lr / 2), mk(Fn.parameters(), lr=lr)
            clean = seeded_tensor(801, (B, 3, ps, ps), lo=0.0, hi=1.0)
            deg = (clean + seeded_tensor(802, (B, 3, ps, ps), scale=50 / 255)).clamp(0, 1)
            alpha = seeded_tensor(803, (B, 1, 1, 1), lo=0.0, hi=1.0)
            real_rand = torch.rand
            torch.rand = lambda *a, **k: alpha.clone()
            import io, contextlib
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                TR.train([([["n"] * B, torch.tensor(de)], deg, clean)], To, Fo, Tn, Fn, 1)
            torch.rand = real_rand
            line = [l for l in buf.getvalue().splitlines() if "Loss_F" in l][0]
            # oracle
            pT, pF = to_t(pT_np), to_t(pF_np)
            mko = O.RMSprop if opt_name == "RMSprop" else O.Adam
            oT, oF = mko(pT, lr / 2), mko(pF, lr)
            logs = O.minimax_iteration(pT, pF, oT, oF, deg, clean, de, alpha, 1.0, 10000.0, paired)
            def l2rel(po_, pr_, p0):
                num = den = 0.0
                for k, v in pr_:
                    if k not in po_:
                        continue
                    d0 = torch.from_numpy(p0[k]).double()
                    num += float(((po_[k].detach().double() - d0) - (v.detach().double() - d0)).pow(2).sum())
                    den += float((v.detach().double() - d0).pow(2).sum())
                return (num / den) ** 0.5
            eT = l2rel(pT, list(Tn.named_parameters()), pT_np)
            eF = l2rel(pF, list(Fn.named_parameters()), pF_np)
            report.append(f"verbatim trainer.train() 1 iteration [{tag}, {opt_name}, B={B}, P={ps}, de_id={de}]: "
                          f"'{line.strip()}' ; oracle {logs}; param-UPDATE L2 rel err T {eT:.2e}, F {eF:.2e}")
            # RMSprop/Adam's first step is ~lr*sign(g): elements with g~0 flip between
            # implementations, so the update is compared in L2, not element-wise max.
            assert eF < 5e-2 and eT < 5e-2, (eT, eF)
            fx[tag + "_cfg"] = np.array([B, ps, int(paired), 31, 32, 801, 802, 803] + de)
            fx[tag + "_line"] = np.array(line)
            fx[tag + "_losses"] = np.array([logs["Loss_F"], logs["Loss_T"], logs["Loss_mse"], logs["gp"]])
            fx[tag + "_Tnorm"] = np.array([float(v.detach().double().norm()) for v in Tn.parameters()])
            fx[tag + "_Fnorm"] = np.array([float(v.detach().double().norm()) for v in Fn.parameters()])
            fx[tag + "_Tdelta"] = np.array([float((v.detach() - torch.from_numpy(pT_np[k])).double().norm())
                                            for k, v in Tn.named_parameters()])
            fx[tag + "_Fdelta"] = np.array([float((v.detach() - torch.from_numpy(pF_np[k])).double().norm())
                                            for k, v in Fn.named_parameters()])
    if "train" in only:
        fx = {}
        for tag, cfg in {"unpaired": (2, 64, False, [2, 3], "RMSprop"), "paired": (2, 64, True, [0, 7], "RMSprop"),
                         "adam": (2, 64, True, [4, 1], "Adam")}.items():
            run_train_variant(fx, tag, *cfg)
        np.savez_compressed(os.path.join(GOLD, "train_iter.npz"), **fx)

    # ---------------------------------------------------------------- F5 at the headline patch size (B=4, P=128)
    if "train128" in only:
        fx = {}
        run_train_variant(fx, "p128", 4, 128, True, [2, 3, 0, 4], "RMSprop")
        np.savez_compressed(os.path.join(GOLD, "train_iter128.npz"), **fx)

    # ---------------------------------------------------------------- F2 at 128x128 with gradients
    if "tnet128" in only:
        pT_np = P.seeded_params(P.tnet_param_shapes(), 11, "T")
        refT.load_state_dict(to_t(pT_np))
        fx = {}
        B, HW, seed = 2, 128, 503
        x = seeded_tensor(seed, (B, 3, HW, HW), lo=0.0, hi=1.0)
        r = seeded_tensor(seed + 50, (B, 3, HW, HW))
        refT.zero_grad()
        y = refT(x)
        res_ref = captured["res.png"]
        (y * r).mean().backward()
        po = {k: v.clone().requires_grad_(True) for k, v in to_t(pT_np).items()}
        yo, reso = O.tnet_forward(po, x, True, return_res=True)
        (yo * r).mean().backward()
        e_out, e_res = relerr(yo, y), relerr(reso, res_ref)
        gn_ref, gerr = [], []
        named = dict(refT.named_parameters())
        for k, v in named.items():
            if v.grad is None:
                assert po[k].grad is None, k
                gn_ref.append(-1.0)
                continue
            gn_ref.append(float(v.grad.double().norm()))
            gerr.append(relerr(po[k].grad, v.grad))
        assert e_out < 1e-4 and e_res < 1e-4 and max(gerr) < 5e-3, (e_out, e_res, max(gerr))
        report.append(f"T_net(decoder=True) B={B} {HW}x{HW} (headline patch size): out rel err {e_out:.2e}, pass-1 res {e_res:.2e}, "
                      f"worst param-grad rel err {max(gerr):.2e} over {len(gerr)} live tensors")
        fx["c_cfg"] = np.array([B, HW, seed, 11])
        fx["c_y"] = y.detach().numpy()
        fx["c_res"] = res_ref.numpy()
        fx["c_gradnorm"] = np.array(gn_ref, dtype=np.float64)
        # 12 tap statistics of the output / residual (mean, l2, strided samples) and strided gradient samples of one
        # tensor per kind and level
        fx["c_ystats"] = np.array([float(y.mean()), float(y.double().norm()), float(res_ref.mean()), float(res_ref.double().norm())])
        for k in ("patch_embed.proj.weight", "output.weight", "latent.3.attn.temperature", "latent.7.ffn.project_out.weight",
                  "refinement.1.ffn.project_in.weight", "refinement.3.attn.qkv.weight", "decoder_level1.0.attn.project_out.weight",
                  "encoder_level1.2.ffn.dwconv.weight", "encoder_level2.3.attn.qkv_dwconv.weight", "resencoder_level2.0.attn.qkv.weight",
                  "reslatent.5.norm2.body.bias", "down3_4.body.0.weight", "up2_1.body.0.weight", "noise_level1.attn.qkv_dwconv.weight",
                  "noise_level3.attn.temperature", "reduce_chan_level2.weight", "decoder_level3.2.norm1.body.weight"):
            fx["c_gs_" + k] = strided(named[k].grad, 128)
        np.savez_compressed(os.path.join(GOLD, "tnet128.npz"), **fx)

    # ---------------------------------------------------------------- F6: 10-step trajectory of the verbatim loop
    if "traj" in only:
        from rcot_amd.synth import make_batch
        import io, contextlib
        B, ps, steps, lr = 4, 128, 10, 1e-4
        de = [2, 3, 0, 4]
        pT_np = P.seeded_params(P.tnet_param_shapes(), 31, "T")
        pF_np = P.seeded_params(P.fnet_param_shapes(ps), 32, "F")
        Tn, Fn = NR.T_net(decoder=True), NR.F_net(patch_size=ps)
        Tn.load_state_dict(to_t(pT_np))
        Fn.load_state_dict(to_t(pF_np))
        TR.opt = Namespace(cuda=False, lr=lr, step=20, pairnum=10 ** 7, batchSize=B, sigma=1.0, Sigma=10000.0, type="pin")
        To, Fo = torch.optim.RMSprop(Tn.parameters(), lr=lr / 2), torch.optim.RMSprop(Fn.parameters(), lr=lr)
        _, hx, hy = make_batch(9100, B, ps, de)                         # held-out batch for the PSNR probe
        with torch.no_grad():
            psnr0 = O.psnr(Tn(hx), hy)
        alphas = [seeded_tensor(9300 + i, (B, 1, 1, 1), lo=0.0, hi=1.0) for i in range(steps)]
        # train() prints only at iteration % 10 == 0, so it is called once per step with a single-batch loader (every call
        # is then "iteration 0": paired, printed); the LR schedule depends on the epoch argument only (kept at 1).
        lines = []
        real_rand = torch.rand
        for i in range(steps):
            _, x, yb = make_batch(9200 + i, B, ps, de)
            torch.rand = lambda *a, _al=alphas[i], **k: _al.clone()
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                TR.train([([["n"] * B, torch.tensor(de)], x, yb)], To, Fo, Tn, Fn, 1)
            torch.rand = real_rand
            lines.append([l for l in buf.getvalue().splitlines() if "Loss_F" in l][0].strip())
            print("traj", i, lines[-1], flush=True)
        with torch.no_grad():
            psnr10 = O.psnr(Tn(hx), hy)
        import re as _re
        tri = np.array([[float(v) for v in _re.findall(r"Loss_\w+: ([-+0-9.eE]+|nan)", l)] for l in lines])
        report.append(f"verbatim trainer.train() x{steps} [RMSprop, B={B}, P={ps}, de_id={de}, paired]: losses step 0 {tri[0].tolist()} -> "
                      f"step {steps - 1} {tri[-1].tolist()}; held-out PSNR {psnr0:.4f} dB -> {psnr10:.4f} dB")
        np.savez_compressed(os.path.join(GOLD, "trajectory.npz"), cfg=np.array([B, ps, steps, 31, 32, 9100, 9200, 9300] + de),
                            losses=tri, psnr=np.array([psnr0, psnr10]),
                            Tnorm=np.array([float(v.detach().double().norm()) for v in Tn.parameters()]),
                            Fnorm=np.array([float(v.detach().double().norm()) for v in Fn.parameters()]))

    # ---------------------------------------------------------------- gradients of the real iteration at its three half-steps
    # (VERDICT r2 item 3): the verbatim trainer.train() with optimizers that snapshot every .grad right before they step.  Per
    # case: the reference's printed line (parsed AND compared with the oracle's loss floats), gp (oracle: not printed upstream),
    # per-tensor gradient norms (-1 = None) and strided samples of F after the critic loss, F after the gradient penalty and T
    # after the generator loss, and the per-tensor update norms.  These replace the live oracle runs of the GPU tier.
    if "itergrads" in only:
        import io, contextlib, re as _re
        from rcot_amd.synth import make_batch

        class SnapOpt:
            """optimizer wrapper: packs every .grad (norm, strided samples) right before the step it delegates"""
            def __init__(self, inner, named, log, nsamp):
                self.inner, self.named, self.log, self.nsamp = inner, named, log, nsamp
            @property
            def param_groups(self):
                return self.inner.param_groups
            def step(self):
                gn, gs = [], []
                for k, p in self.named:
                    g = p.grad
                    gn.append(-1.0 if g is None else float(g.double().norm()))
                    if g is not None:
                        gs.append(strided64(g, self.nsamp))
                self.log.append((np.array(gn, dtype=np.float64), np.concatenate(gs).astype(np.float32)))
                self.inner.step()

        cases = {   # tag: (mode, B, ps, paired, unpaired_targets, de, optimizer); mode 0: seeded tensors (801..803), mode 1: synth.make_batch(77)
            "unpaired": (0, 2, 64, False, False, [2, 3], "RMSprop"), "paired": (0, 2, 64, True, False, [0, 7], "RMSprop"),
            "adam": (0, 2, 64, True, False, [4, 1], "Adam"), "p128": (0, 4, 128, True, False, [2, 3, 0, 4], "RMSprop"),
            "cfg3p": (1, 2, 128, True, False, [3, 3], "RMSprop"), "cfg3u": (1, 2, 128, False, False, [3, 3], "RMSprop"),
            "cfg5": (1, 2, 256, False, True, [4, 4], "RMSprop"),
            # round 5: BASELINE configs[1] — the headline workload — at its FULL batch (B = 8, 128x128, denoise_50, paired, RMSprop)
            "cfg2b8": (1, 8, 128, True, False, [2] * 8, "RMSprop")}
        import gc
        out_path = os.path.join(GOLD, "iter_grads.npz")
        fx = dict(np.load(out_path)) if (os.path.isfile(out_path) and os.environ.get("ITERGRADS_KEEP")) else {}
        for tag, (mode, B, ps, paired, unp, de, opt_name) in cases.items():
            if tag + "_cfg" in fx:
                continue
            lr = 1e-4
            pT_np = P.seeded_params(P.tnet_param_shapes(), 31, "T")
            pF_np = P.seeded_params(P.fnet_param_shapes(ps), 32, "F")
            Tn, Fn = NR.T_net(decoder=True), NR.F_net(patch_size=ps)
            Tn.load_state_dict(to_t(pT_np))
            Fn.load_state_dict(to_t(pF_np))
            TR.opt = Namespace(cuda=False, lr=lr, step=20, pairnum=(10 ** 7 if paired else 0), batchSize=B, sigma=1.0, Sigma=10000.0,
                               type="pin")
            mk = torch.optim.RMSprop if opt_name == "RMSprop" else torch.optim.Adam
            logT, logF = [], []
            To = SnapOpt(mk(Tn.parameters(), lr=lr / 2), list(Tn.named_parameters()), logT, 128)
            Fo = SnapOpt(mk(Fn.parameters(), lr=lr), list(Fn.named_parameters()), logF, 512)
            if mode == 0:
                s1, s2, s3 = 801, 802, 803
                clean = seeded_tensor(s1, (B, 3, ps, ps), lo=0.0, hi=1.0)
                deg = (clean + seeded_tensor(s2, (B, 3, ps, ps), scale=50 / 255)).clamp(0, 1)
                alpha = seeded_tensor(s3, (B, 1, 1, 1), lo=0.0, hi=1.0)
            else:
                s1, s2, s3 = 77, 0, 78
                _, deg, clean = make_batch(s1, B, ps, de, unpaired=unp)
                alpha = seeded_tensor(s3, (B,), lo=0.0, hi=1.0).view(B, 1, 1, 1)
            real_rand = torch.rand
            torch.rand = lambda *a, **k: alpha.clone()
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                TR.train([([["n"] * B, torch.tensor(de)], deg, clean)], To, Fo, Tn, Fn, 1)
            torch.rand = real_rand
            line = [l for l in buf.getvalue().splitlines() if "Loss_F" in l][0].strip()
            printed = [float(v) for v in _re.findall(r"Loss_\w+: ([-+0-9.eE]+)", line)]
            assert len(logF) == 2 and len(logT) == 1 and len(printed) == 3
            fx[tag + "_Tdelta"] = np.array([float((v.detach() - torch.from_numpy(pT_np[k])).double().norm()) for k, v in Tn.named_parameters()])
            fx[tag + "_Fdelta"] = np.array([float((v.detach() - torch.from_numpy(pF_np[k])).double().norm()) for k, v in Fn.named_parameters()])
            del Tn, Fn, To, Fo                                            # (F_net(256) is 1.1 GB per copy: free the reference first)
            gc.collect()
            # the oracle on the same iteration: its loss floats must round to what the reference printed (5 significant digits)
            pT, pF = to_t(pT_np), to_t(pF_np)
            del pT_np, pF_np
            mko = O.RMSprop if opt_name == "RMSprop" else O.Adam
            logs = O.minimax_iteration(pT, pF, mko(pT, lr / 2), mko(pF, lr), deg, clean, de, alpha, 1.0, 10000.0, paired)
            del pT, pF
            gc.collect()
            for got, want in zip((logs["Loss_F"], logs["Loss_T"], logs["Loss_mse"]), printed):
                assert abs(got - want) <= 2e-4 * max(abs(want), 1e-4), (tag, got, want, line)
            fx[tag + "_cfg"] = np.array([mode, B, ps, int(paired), int(unp), 31, 32, s1, s2, s3] + de)
            fx[tag + "_line"] = np.array(line)
            fx[tag + "_printed"] = np.array(printed)                       # the REFERENCE's numbers (Loss_F, Loss_T, Loss_mse)
            fx[tag + "_losses"] = np.array([logs["Loss_F"], logs["Loss_T"], logs["Loss_mse"], logs["gp"]])
            for key, (gn, gs) in ((tag + "_Fc", logF[0]), (tag + "_Fg", logF[1]), (tag + "_T", logT[0])):
                fx[key + "_gn"], fx[key + "_gs"] = gn, gs
            report.append(f"iteration gradients [{tag}: {opt_name}, B={B}, P={ps}, de_id={de}, paired={paired}, unpaired targets={unp}]: "
                          f"reference printed '{line}' == oracle floats to 5 digits (gp {logs['gp']:.6g} from the oracle); F grads after "
                          f"critic loss / after GP and T grads after the generator loss stored (norms + strided samples)")
            print(report[-1], flush=True)
            np.savez_compressed(out_path, **fx)                          # (rewritten after every case)

    # ---------------------------------------------------------------- reference outputs for the remaining live-oracle GPU tests
    if "gpufx" in only:
        from rcot_amd.synth import make_batch
        fx = {}
        # (a) north_star forward bar on another 128x128 batch (tests/test_network_gpu.py::test_tnet_vs_oracle_128)
        refT.load_state_dict(to_t(P.seeded_params(P.tnet_param_shapes(), 11, "T")))
        with torch.no_grad():
            fx["fwd128_cfg"] = np.array([2, 128, 900, 11])
            fx["fwd128_y"] = refT(seeded_tensor(900, (2, 3, 128, 128), lo=0.0, hi=1.0)).numpy()
            # (b) whole-image forward at a non-square size (tests/test_pipeline_gpu.py): same parameters, torch generator seed 3
            fx["whole_cfg"] = np.array([1, 96, 160, 3, 11])
            fx["whole_y"] = refT(torch.rand(1, 3, 96, 160, generator=torch.Generator().manual_seed(3))).numpy()
            # (b2) a size whose 1/8-resolution plane has an odd pixel count (5 x 7): the reference accepts it (trainer.py:195-198)
            fx["odd_cfg"] = np.array([1, 40, 56, 4, 11])
            fx["odd_y"] = refT(torch.rand(1, 3, 40, 56, generator=torch.Generator().manual_seed(4))).numpy()
            # (c) evaluate(): PSNR of the reference's output on the two valid validation images of tests/synth_folders.py
            import glob, tempfile
            from PIL import Image
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from synth_folders import dataset_tree
            root = tempfile.mkdtemp()
            dataset_tree(root, 1)
            ps = []
            for d, t in list(zip(sorted(glob.glob(f"{root}/val/input/*")), sorted(glob.glob(f"{root}/val/target/*"))))[:2]:
                x = torch.from_numpy(np.array(Image.open(d).convert("RGB")).transpose(2, 0, 1)).float().div(255).unsqueeze(0)
                yt = torch.from_numpy(np.array(Image.open(t).convert("RGB")).transpose(2, 0, 1)).float().div(255).unsqueeze(0)
                ps.append(O.psnr(refT(x), yt))
            fx["eval_psnr"] = np.array(ps)
        report.append(f"reference forward outputs stored: B=2 128x128 (seed 900), 1x3x96x160 (torch seed 3), PSNR of the two valid validation images {ps}")
        np.savez_compressed(os.path.join(GOLD, "gpu_fixtures.npz"), **fx)

    # ---------------------------------------------------------------- BASELINE configs[0]: the older MPRNet transport map on stock ops
    if "mprnet" in only:
        import io, contextlib, re as _re
        import Net as NM
        from rcot_amd import mprnet as MP
        from rcot_amd.synth import make_batch
        refM = NM.T_net()
        assert [(k, tuple(v.shape)) for k, v in refM.state_dict().items()] == MP.mprnet_param_shapes(), "Net.T_net state_dict contract"
        shapes = MP.mprnet_param_shapes()
        prm = to_t(P.seeded_params([(n, s) for n, s in shapes if not n.endswith("body.1.weight")], 71, "T"))
        for n, _s in shapes:
            if n.endswith("body.1.weight"):
                prm[n] = torch.full((1,), 0.2)
        refM.load_state_dict(prm)
        x = seeded_tensor(72, (2, 3, 64, 64), lo=0.0, hi=1.0)
        r = seeded_tensor(73, (2, 3, 64, 64))
        mine = MP.MPRNetT(seed=0)
        mine.load_state_dict(prm)
        yr = refM(x)
        (yr * r).mean().backward()
        ym = mine(x)
        (ym * r).mean().backward()
        gref = dict(refM.named_parameters())
        e = [relerr(ym, yr)] + [relerr(mine.p[k].grad, v.grad) for k, v in gref.items() if v.grad is not None]
        assert max(e) < 1e-5, max(e)
        # ten verbatim iterations of configs[0]: B=4, 128x128, de_type single (de_id 7), pairnum 0, RMSprop lr 1e-4
        B, ps, steps, lr, de = 4, 128, 10, 1e-4, [7] * 4
        pF = to_t(P.seeded_params(P.fnet_param_shapes(ps), 32, "F"))
        Tn, Fn = NM.T_net(), NR.F_net(patch_size=ps)
        Tn.load_state_dict(prm)
        Fn.load_state_dict(pF)
        TR.opt = Namespace(cuda=False, lr=lr, step=20, pairnum=0, batchSize=B, sigma=1.0, Sigma=10000.0, type="pin")
        To, Fo = torch.optim.RMSprop(Tn.parameters(), lr=lr / 2), torch.optim.RMSprop(Fn.parameters(), lr=lr)
        Tm, Fm = MP.MPRNetT(seed=0), MP.FNetTorch(ps, seed=0)
        Tm.load_state_dict(prm)
        Fm.load_state_dict(pF)
        Tom, Fom = torch.optim.RMSprop(Tm.parameters(), lr=lr / 2), torch.optim.RMSprop(Fm.parameters(), lr=lr)
        lines, mine_l = [], []
        real_rand = torch.rand
        for i in range(steps):
            _, xb, yb = make_batch(7100 + i, B, ps, de)
            al = seeded_tensor(7200 + i, (B, 1, 1, 1), lo=0.0, hi=1.0)
            torch.rand = lambda *a, _al=al, **k: _al.clone()
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                TR.train([([["n"] * B, torch.tensor(de)], xb, yb)], To, Fo, Tn, Fn, 1)
            torch.rand = real_rand
            lines.append([l for l in buf.getvalue().splitlines() if "Loss_F" in l][0].strip())
            s_ = MP.torch_minimax_iteration(Tm, Fm, Tom, Fom, xb, yb, de, al.view(B), 1.0, 10000.0, False)
            mine_l.append([s_["Loss_F"], s_["Loss_T"], s_["Loss_mse"]])
            print("mprnet", i, lines[-1], mine_l[-1], flush=True)
        tri = np.array([[float(v) for v in _re.findall(r"Loss_\w+: ([-+0-9.eE]+|nan)", l)] for l in lines])
        assert np.abs(np.array(mine_l) - tri).max() <= 2e-2 * np.abs(tri).max()
        report.append(f"MPRNet Net.T_net (BASELINE configs[0]): state_dict contract (127 tensors) identical; rcot_amd.mprnet forward + "
                      f"all parameter gradients vs the reference max rel err {max(e):.2e}; ten verbatim trainer.train() iterations "
                      f"[B=4, 128x128, de_id 7, unpaired, RMSprop] losses step 0 {tri[0].tolist()} -> step 9 {tri[-1].tolist()}, the "
                      f"stock-ops loop of rcot_amd.mprnet tracks them within {np.abs(np.array(mine_l) - tri).max() / np.abs(tri).max():.1e}")
        np.savez_compressed(os.path.join(GOLD, "mprnet.npz"), cfg=np.array([2, 64, 71, 72, 73]), y=yr.detach().numpy(),
                            gn=np.array([float(v.grad.double().norm()) if v.grad is not None else -1.0 for v in gref.values()]),
                            traj_cfg=np.array([B, ps, steps, 71, 32, 7100, 7200] + de), traj=tri)

    # ---------------------------------------------------------------- 8(f4): fixtures for the HIP form of the MPRNet transport map
    # (tests/test_mprnet_gpu.py): strided samples of every parameter gradient and of the input gradient of the reference's Net.T_net
    # at 2 x 64 x 64 (same parameters / input / loss as the "mprnet" section), its output on a NON-square whole image (the testers'
    # crop-to-a-multiple-of-4 case, tester.py:77-84), and one CAB / DownSample / SkipUpSample forward + backward on their own
    if "mprnetfx" in only:
        import Net as NM
        from rcot_amd import mprnet as MP
        shapes = MP.mprnet_param_shapes()
        prm = to_t(P.seeded_params([(n, s) for n, s in shapes if not n.endswith("body.1.weight")], 71, "T"))
        for n, _s in shapes:
            if n.endswith("body.1.weight"):
                prm[n] = torch.full((1,), 0.2)
        refM = NM.T_net()
        refM.load_state_dict(prm)
        x = seeded_tensor(72, (2, 3, 64, 64), lo=0.0, hi=1.0).requires_grad_(True)
        r = seeded_tensor(73, (2, 3, 64, 64))
        yr = refM(x)
        (yr * r).mean().backward()
        fx = {"cfg": np.array([2, 64, 71, 72, 73])}
        names = []
        for k, v in refM.named_parameters():                       # distinct tensors, first occurrence (the shared slope once)
            names.append(k)
            if v.grad is not None:
                fx["gs_" + k] = strided(v.grad, 64)
        fx["names"] = np.array(names)
        fx["dx"] = x.grad.numpy()
        xw = seeded_tensor(74, (1, 3, 36, 52), lo=0.0, hi=1.0)
        with torch.no_grad():
            fx["whole_cfg"] = np.array([1, 36, 52, 74])
            fx["whole_y"] = refM(xw).numpy()
        # leaf modules (shared PReLU slope 0.2 as above)
        act = torch.nn.PReLU()
        with torch.no_grad():
            act.weight.fill_(0.2)
        C = 80
        cab = NM.CAB(C, 3, 4, bias=False, act=act)
        cshapes = [(k, tuple(v.shape)) for k, v in cab.state_dict().items() if not k.endswith("body.1.weight")]
        cprm = to_t(P.seeded_params(cshapes, 81, "T"))
        cprm["body.1.weight"] = torch.full((1,), 0.2)
        cab.load_state_dict(cprm)
        cx = seeded_tensor(82, (2, C, 12, 20)).requires_grad_(True)
        cg = seeded_tensor(83, (2, C, 12, 20))
        cy = cab(cx)
        cy.backward(cg)
        fx["cab_cfg"] = np.array([2, C, 12, 20, 81, 82, 83])
        fx["cab_y"], fx["cab_dx"] = cy.detach().numpy(), cx.grad.numpy()
        for k, v in cab.named_parameters():
            fx["cab_g_" + k] = v.grad.numpy()
        dn, up = NM.DownSample(C, 48), NM.SkipUpSample(C, 48)
        dprm = to_t(P.seeded_params([(k, tuple(v.shape)) for k, v in dn.state_dict().items()], 84, "T"))
        uprm = to_t(P.seeded_params([(k, tuple(v.shape)) for k, v in up.state_dict().items()], 85, "T"))
        dn.load_state_dict(dprm)
        up.load_state_dict(uprm)
        rx = seeded_tensor(86, (2, C, 12, 20)).requires_grad_(True)
        dy_ = dn(rx)
        dg = seeded_tensor(87, tuple(dy_.shape))
        dy_.backward(dg)
        fx["down_y"], fx["down_dx"], fx["down_gw"] = dy_.detach().numpy(), rx.grad.numpy(), dn.down[1].weight.grad.numpy()
        ux = seeded_tensor(88, (2, C + 48, 6, 10)).requires_grad_(True)
        us = seeded_tensor(89, (2, C, 12, 20))
        uy = up(ux, us)
        ug = seeded_tensor(90, tuple(uy.shape))
        uy.backward(ug)
        fx["up_y"], fx["up_dx"], fx["up_gw"] = uy.detach().numpy(), ux.grad.numpy(), up.up[1].weight.grad.numpy()
        fx["resample_seeds"] = np.array([84, 85, 86, 87, 88, 89, 90])
        np.savez_compressed(os.path.join(GOLD, "mprnet_hipfx.npz"), **fx)
        report.append(f"MPRNet fixtures for the HIP form (mprnet_hipfx.npz): {sum(1 for k in fx if k.startswith('gs_'))} gradient sample sets + input "
                      f"gradient at 2x64x64, whole-image output at 36x52, CAB / DownSample / SkipUpSample forward + backward at 2x80x12x20")

    # ---------------------------------------------------------------- C6: the constructors' parameter distributions
    if "init" in only:
        torch.manual_seed(123)
        worst = 0.0
        for kind, mod in (("F", NR.F_net(patch_size=128)), ("T", NR.T_net(decoder=True))):
            sd = dict(mod.state_dict())
            for n, t in sd.items():
                t = t.double()
                if n.endswith("body.weight") or n.endswith("temperature"):
                    assert bool((t == 1).all()), n
                elif n.endswith("body.bias"):
                    assert bool((t == 0).all()), n
                elif kind == "F" and n.startswith("features.") and n.endswith(".weight"):
                    if t.numel() >= 20000:
                        worst = max(worst, abs(float(t.std()) / 0.02 - 1))
                        assert abs(float(t.std()) / 0.02 - 1) < 0.02, (n, float(t.std()))
                else:
                    wshape = sd[n[:-len("bias")] + "weight"].shape if n.endswith(".bias") else t.shape
                    bound = 1.0 / float(np.prod(wshape[1:])) ** 0.5
                    assert float(t.abs().max()) <= bound * (1 + 1e-6), (n, float(t.abs().max()), bound)
                    if t.numel() >= 20000:
                        worst = max(worst, abs(float(t.std()) / (bound / 3 ** 0.5) - 1))
                        assert abs(float(t.std()) / (bound / 3 ** 0.5) - 1) < 0.02, n
        report.append(f"C6 init: the reference's constructors follow the rule rcot_amd.net_restormer._reference_init restates (F_net Conv2d "
                      f"weights N(0, 0.02); all other weights / biases U(+-1/sqrt(fan_in of the layer's weight)); LayerNorm 1/0, temperature "
                      f"1): worst std deviation from the rule over tensors >= 20000 elements {worst:.2e}")

    # ---------------------------------------------------------------- checkpoint interchange (trainer.py:362-371, tester.py:54)
    if "ckpt" in only:
        import subprocess
        import tempfile
        pT_np = P.seeded_params(P.tnet_param_shapes(), 11, "T")
        pF_np = P.seeded_params(P.fnet_param_shapes(64), 12, "F")
        tmp = tempfile.mkdtemp()
        ours, theirs = os.path.join(tmp, "ours.pth"), os.path.join(tmp, "ref.pth")
        # (1) our format, written with OUR shim as Net_Restormer (own process: this one has the reference's module loaded)
        code1 = ("import sys, torch; sys.path.insert(0, %r)\n"
                 "import Net_Restormer as N\nfrom rcot_amd import params as P\n"
                 "t = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}\n"
                 "T = N.T_net.from_state_dict(t(P.seeded_params(P.tnet_param_shapes(), 11, 'T')), decoder=True)\n"
                 "F = N.F_net.from_state_dict(t(P.seeded_params(P.fnet_param_shapes(64), 12, 'F')), patch_size=64)\n"
                 "torch.save({'epoch': 3, 'Tnet': T, 'Fnet': F}, %r)\n" % (ROOT, ours))
        subprocess.run([sys.executable, "-c", code1], check=True)
        # (2) opened HERE, where Net_Restormer is the reference's file: the reference's resume path (trainer.py:103-106)
        ck = torch.load(ours, weights_only=False)
        assert isinstance(ck["Tnet"], NR.T_net) and isinstance(ck["Fnet"], NR.F_net)
        T2, F2 = NR.T_net(decoder=True), NR.F_net(patch_size=64)
        T2.load_state_dict(ck["Tnet"].state_dict())
        F2.load_state_dict(ck["Fnet"].state_dict())
        for k, v in T2.state_dict().items():
            assert np.array_equal(v.numpy(), pT_np[k]), k
        torch.save({"epoch": 5, "Tnet": T2, "Fnet": F2}, theirs)              # the reference's own format (:362-369)
        # (3) the reference-made file opened with OUR shim
        code3 = ("import sys, torch, numpy as np; sys.path.insert(0, %r)\n"
                 "from rcot_amd.compat import load_checkpoint\nfrom rcot_amd import params as P\n"
                 "ck = load_checkpoint(%r)\n"
                 "assert ck['epoch'] == 5 and type(ck['Tnet']).__module__ == 'Net_Restormer'\n"
                 "sd, ref = ck['Tnet'].state_dict(), P.seeded_params(P.tnet_param_shapes(), 11, 'T')\n"
                 "assert list(sd) == list(ref) and all(np.array_equal(sd[k].numpy(), ref[k]) for k in ref)\n"
                 "sdf, reff = ck['Fnet'].state_dict(), P.seeded_params(P.fnet_param_shapes(64), 12, 'F')\n"
                 "assert list(sdf) == list(reff) and all(np.array_equal(sdf[k].numpy(), reff[k]) for k in reff)\n"
                 "assert ck['Tnet']._ctor == {'decoder': True} and ck['Fnet']._ctor == {'patch_size': 64}\n" % (ROOT, theirs))
        subprocess.run([sys.executable, "-c", code3], check=True)
        report.append("checkpoint interchange: a checkpoint written by rcot_amd (Net_Restormer.T_net/F_net objects) unpickles under the "
                      "REFERENCE's Net_Restormer.py as reference modules whose state_dict() equals the saved tensors (816 + 22, "
                      "bit-exact), i.e. trainer.py:100-108 resumes from it; a checkpoint written by the reference (whole module trees, "
                      "trainer.py:362-369) unpickles under the repo's shim with identical state_dicts and recovered constructor "
                      "arguments (decoder=True, patch_size=64)")

    # ---------------------------------------------------------------- f2: the data contract either side of the step
    if "data" in only:
        import random as _random
        import tempfile
        from PIL import Image
        from util import image_utils as IU
        from util.degradation_utils import Degradation
        from util import dataset_utils as DU
        from rcot_amd import data as D
        fx = {}
        # (1) data_augmentation, the 8 dihedral modes (util/image_utils.py:133-163) on a non-symmetric uint8 patch
        patch = rng(501).integers(0, 256, size=(12, 12, 3), dtype=np.uint8)
        fx["aug_in"] = patch
        fx["aug_out"] = np.stack([np.ascontiguousarray(IU.data_augmentation(torch.from_numpy(patch) if m == 0 else patch, m))
                                  for m in range(8)])
        # (2) crop_img (util/image_utils.py:59-64): centre crop to multiples of `base`
        shapes = [(70, 93), (64, 64), (81, 50), (321, 481), (17, 33)]
        crops = []
        for (h, w) in shapes:
            im = np.arange(h * w * 3, dtype=np.int64).reshape(h, w, 3)
            c = IU.crop_img(im, base=16)
            crops.append([h, w, c.shape[0], c.shape[1], int(c[0, 0, 0]), int(c[-1, -1, 2])])
            mine = D.crop_to_multiple(im, 16)
            assert mine.shape == c.shape and np.array_equal(mine, c)
        fx["crop"] = np.array(crops, dtype=np.int64)
        # (3) the synthetic-noise degradation (util/degradation_utils.py:21-27) for a FIXED noise field: clip, then truncation to uint8
        noise = rng(502).standard_normal((12, 12, 3))
        keep = np.random.randn
        np.random.randn = lambda *shape: noise
        try:
            deg = np.stack([Degradation(Namespace(patch_size=12))._degrade_by_type(patch, t)[0] for t in (0, 1, 2)])
        finally:
            np.random.randn = keep
        fx["noise"], fx["noise_out"] = noise, deg
        # (4) sample-list rules (util/dataset_utils.py:63-228) on a miniature of the dataset layout
        r = tempfile.mkdtemp()

        def png(path, h, w, seed):
            os.makedirs(os.path.dirname(path), exist_ok=True)
            Image.fromarray(rng(seed).integers(0, 256, size=(h, w, 3), dtype=np.uint8)).save(path)
        den = [f"img{i}.png" for i in range(3)]
        for i, n in enumerate(den):
            png(f"{r}/Denoise/{n}", 70 + i, 90 + 2 * i, 10 + i)
        png(f"{r}/Denoise/not_listed.png", 64, 64, 99)
        for d_ in ("noisy", "rainy", "hazy"):
            os.makedirs(f"{r}/lists/{d_}")
        open(f"{r}/lists/noisy/denoise.txt", "w").write("\n".join(den) + "\n")
        open(f"{r}/lists/rainy/rainTrain.txt", "w").write("rainy/rain-1.png\nrainy/rain-2.png\n")
        for i in (1, 2):
            png(f"{r}/Derain/rainy/rain-{i}.png", 80, 96, 20 + i)
            png(f"{r}/Derain/gt/norain-{i}.png", 80, 96, 30 + i)
        open(f"{r}/lists/hazy/hazy_outside.txt", "w").write("synthetic/part1/0025_0.8_0.04.png\n")
        png(f"{r}/Dehaze/synthetic/part1/0025_0.8_0.04.png", 72, 72, 41)
        png(f"{r}/Dehaze/original/0025.png", 72, 72, 42)
        for n in ("a.png", "b.png"):
            png(f"{r}/Single/degraded/{n}", 48, 48, 50)
            png(f"{r}/Single/target/{n}", 48, 48, 51)
        a = Namespace(de_type=["denoise_15", "denoise_50", "derain", "dehaze", "single"], data_file_dir=f"{r}/lists/",
                      denoise_dir=f"{r}/Denoise/", derain_dir=f"{r}/Derain/", dehaze_dir=f"{r}/Dehaze/", single_dir=f"{r}/Single/",
                      patch_size=32)
        _random.seed(0)
        ds = DU.TrainDataset(Namespace(**vars(a), ))
        ref_ids = sorted((os.path.relpath(os.path.join(a.single_dir, "degraded/", s["clean_id"]) if s["de_type"] == 7 else s["clean_id"], r),
                          int(s["de_type"])) for s in ds.sample_ids)
        a2 = Namespace(**vars(a))
        a2.de_type = ["denoise_15", "denoise_50", "derain", "dehaze", "single"]      # (the reference shuffles args.de_type in place)
        mine = sorted((os.path.relpath(s["file"], r), int(s["de"])) for s in D.build_sample_ids(a2))
        assert mine == ref_ids, "sample lists differ from the reference's"
        fx["ids_files"] = np.array([f for f, _ in ref_ids])
        fx["ids_de"] = np.array([d for _, d in ref_ids], dtype=np.int64)
        names = ["/d/Derain/rainy/rain-100.png", "x/rainy/rain-7.jpg", "/d/Dehaze/synthetic/part1/0025_0.8_0.04.png", "h/synthetic/12_1_0.2.jpg"]
        fx["gt_in"] = np.array(names)
        fx["gt_rain"] = np.array([ds._get_raingt_name(n) for n in names[:2]])
        fx["gt_hazy"] = np.array([ds._get_nonhazy_name(n) for n in names[2:]])
        assert [D.rain_gt_name(n) for n in names[:2]] == list(fx["gt_rain"]) and [D.nonhazy_name(n) for n in names[2:]] == list(fx["gt_hazy"])
        # (5) one verbatim __getitem__ per paired task with the random draws recorded: crop origin and augmentation mode -> patch pair
        items = []
        for idx, s in enumerate(ds.sample_ids):
            if s["de_type"] in (3, 4) and not any(it[0] == s["de_type"] for it in items):
                _random.seed(77 + s["de_type"])
                (nm, de_id), dpatch, cpatch = ds[idx]                     # ToTensor is the identity stub: HWC uint8 arrays come back
                _random.seed(77 + s["de_type"])
                full = IU.crop_img(np.array(Image.open(s["clean_id"]).convert("RGB")), base=16)
                y0 = _random.randint(0, full.shape[0] - a.patch_size)
                x0 = _random.randint(0, full.shape[1] - a.patch_size)
                mode_ = _random.randint(1, 7)
                items.append((int(s["de_type"]), os.path.relpath(s["clean_id"], r), y0, x0, mode_, np.ascontiguousarray(dpatch),
                              np.ascontiguousarray(cpatch), str(nm)))
        assert len(items) == 2
        fx["item_meta"] = np.array([[it[0], it[2], it[3], it[4]] for it in items], dtype=np.int64)
        fx["item_file"] = np.array([it[1] for it in items])
        fx["item_name"] = np.array([it[7] for it in items])
        fx["item_deg"] = np.stack([it[5] for it in items])
        fx["item_clean"] = np.stack([it[6] for it in items])
        fx["tree_seeds"] = np.array([10, 11, 12, 99, 21, 22, 31, 32, 41, 42, 50, 51], dtype=np.int64)
        np.savez_compressed(os.path.join(GOLD, "data_contract.npz"), **fx)
        report.append("f2 data contract (tests/golden/data_contract.npz, made by the REFERENCE's util/image_utils.py, "
                      "util/degradation_utils.py and util/dataset_utils.py on a generated miniature dataset): data_augmentation's 8 "
                      "dihedral modes; crop_img on 5 sizes (== rcot_amd.data.crop_to_multiple, asserted here); the denoise degradation "
                      "for a fixed noise field (clip then truncate to uint8, sigma 15/25/50); TrainDataset's merged sample list "
                      f"({len(ref_ids)} ids incl. the x5 / x360 / x5 replication and the unlisted-file rule == build_sample_ids as a "
                      "multiset, asserted here); the rain / haze ground-truth naming rules; one verbatim __getitem__ per paired task "
                      "(derain, dehaze) with its crop origin and augmentation mode recorded")

    mode = "a" if args.only else "w"
    with open(os.path.join(ROOT, "oracle", "PINNED.md"), mode) as f:
        if args.only:
            f.write("\n## added by `--only %s`\n\n" % args.only)
        else:
            f.write("# Oracle pin report\n\nGenerated by `python oracle/pin_against_reference.py` in the build container "
                "(torch %s CPU, reference imported from /root/reference).\n"
                "The reference ships no tests for this path (SURVEY.md section 4); the pins are direct agreement "
                "with the imported reference (below) and the reference-produced fixtures in tests/golden/.\n\n" % torch.__version__)
        for r in report:
            f.write("* " + r + "\n")
    print("\n".join(report))


if __name__ == "__main__":
    main()
