"""Whole-image inference behind the CLI of the reference's testers (tester.py, tester_noise.py), on the HIP kernels.

    python -m rcot_amd.tester --model checkpoint/model_X__N_S.pth --degset dir/ --tarset dir/ --save OUT/ --savetar TAR/ --saveres RES/
                              [--noise_sigma 50] [--tile 512 --overlap 32]

Same walk as the reference (tester.py:56-113, tester_noise.py:65-115): sorted ``glob(degset + "*")`` / ``glob(tarset + "*")`` pairs,
RGB, [0, 1] floats, pairs of different shapes skipped; ``tester.py`` crops rows / columns from the END until H and W are multiples
of 4 (:77-84), ``tester_noise.py`` — chosen with ``--noise_sigma`` — drops the FIRST row and column when either is not (:84-86,
kept as is: an odd size stays unusable, and is skipped here with a message where the reference would fail inside the network) and
adds N(0, (sigma / 255)^2) noise to the degraded image (:93-101, numpy's global generator there; ``--seed`` here).  The network is what
the checkpoint holds: a pickled ``Net_Restormer.T_net`` object (the reference's and this package's Restormer checkpoints,
rcot_amd/compat.py) or the state_dict form of ``--backbone mprnet`` runs (``{"backbone": "mprnet"}`` -> ``MPRNetHip``).  Outputs: the
restored image, the target and the scaled residual (x2 resp. x3 with noise, :109 / :111) as PNGs under the three folders, then
PSNR / SSIM over the two folders as ``evaluate.calculate_evaluation_floder`` computes them (evaluate.py:43-106; cv2 and skimage are
not in this image, so both metrics are restated in numpy from their definitions: PSNR = skimage's for uint8 images, SSIM = the
script's own 2 x 2 box-window form).  FID (tester.py:115-118) needs a pretrained Inception network and is not computed.

Superset: ``--tile T`` processes the image as overlapping T x T tiles (``--overlap`` pixels, averaged where tiles overlap) for sizes
one does not want to hold whole; the default is the reference's whole-image call.  Restormer takes H, W multiples of 8 (its three
PixelUnshuffle stages; the reference raises on other sizes), MPRNet multiples of 4.
"""
from __future__ import annotations

import argparse
import glob
import math
import os

import numpy as np
import torch

parser = argparse.ArgumentParser(description="RCOT evaluation on MI355X (tester.py / tester_noise.py flags)")
parser.add_argument("--cuda", action="store_true", help="accepted for CLI compatibility (the HIP path always runs on the GPU)")
parser.add_argument("--model", default="./checkpoint/model_Dehazing__99_10.0.pth", type=str, help="model path")
parser.add_argument("--degset", default="./datasets/Dehazing/outdoor/hazy/", type=str, help="degraded data")
parser.add_argument("--tarset", default="./datasets/Dehazing/outdoor/gt/", type=str, help="target data")
parser.add_argument("--saveres", default="./results/Dehazing/RES/", type=str, help="savepath, Default: residual")
parser.add_argument("--save", default="./results/Dehazing/OUT/", type=str, help="savepath, Default: results")
parser.add_argument("--savetar", default="./results/Dehazing/TAR/", type=str, help="savepath, Default: targets")
parser.add_argument("--gpus", default="0", type=str, help="gpu ids (accepted, ignored: one process, current device)")
parser.add_argument("--noise_sigma", type=float, default=None, help="tester_noise.py: add N(0, sigma^2) (8-bit scale) to the degraded image")
parser.add_argument("--seed", type=int, default=0, help="seed of the added noise")
parser.add_argument("--tile", type=int, default=0, help="superset: tile size (0 = whole image, the reference's behaviour)")
parser.add_argument("--overlap", type=int, default=32, help="superset: tile overlap in pixels")


# ------------------------------------------------------------------------------- metrics (evaluate.py)
def psnr_uint8(im1: np.ndarray, im2: np.ndarray) -> float:
    """skimage.metrics.peak_signal_noise_ratio for uint8 images (data range 255), as evaluate.py:84 calls it"""
    err = float(np.mean((im1.astype(np.float64) - im2.astype(np.float64)) ** 2))
    return float("inf") if err == 0.0 else 10.0 * math.log10(255.0 * 255.0 / err)


def _box2(a: np.ndarray) -> np.ndarray:
    """cv2.filter2D with the 2 x 2 window of cv2.getGaussianKernel(2, 1) (both taps 0.5; anchor at the window centre (1, 1): the taps
    sit at offsets -1 and 0), cropped [5:-5] as evaluate.py:54-60 does — the border handling never reaches the crop"""
    s = 0.25 * (a[:-1, :-1] + a[:-1, 1:] + a[1:, :-1] + a[1:, 1:])          # s[y-1, x-1] = window ending at (y, x), y, x >= 1
    return s[4:-5, 4:-5]


def ssim_plane(img1: np.ndarray, img2: np.ndarray) -> float:
    """evaluate.ssim (:43-63) for one 2-D plane in [0, 255]"""
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    a, b = img1.astype(np.float64), img2.astype(np.float64)
    mu1, mu2 = _box2(a), _box2(b)
    s1, s2, s12 = _box2(a * a) - mu1 * mu1, _box2(b * b) - mu2 * mu2, _box2(a * b) - mu1 * mu2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))
    return float(m.mean())


def ssim_image(im1: np.ndarray, im2: np.ndarray) -> float:
    """``ssim(im1, im2)`` as evaluate.py:86 calls its own function on H x W x 3 arrays: filter2D works per channel, the mean runs
    over everything"""
    return float(np.mean([ssim_plane(im1[:, :, c], im2[:, :, c]) for c in range(im1.shape[2])]))


def evaluate_folders(path1: str, path2: str):
    """evaluate.calculate_evaluation_floder (:65-106): mean / best / worst PSNR and SSIM over the sorted file pairs"""
    from PIL import Image
    a, b = sorted(os.listdir(path1)), sorted(os.listdir(path2))
    ps, ss = [], []
    for n1, n2 in zip(a, b):
        i1 = np.array(Image.open(os.path.join(path1, n1)).convert("RGB"))
        i2 = np.array(Image.open(os.path.join(path2, n2)).convert("RGB"))
        ps.append(psnr_uint8(i1, i2))
        ss.append(ssim_image(i1, i2))
    if not ps:
        nan = float("nan")
        return nan, nan, nan, nan, nan, nan
    return sum(ps) / len(b), sum(ss) / len(b), max(ps), max(ss), min(ps), min(ss)


# ------------------------------------------------------------------------------- the network of a checkpoint
def load_network(path: str):
    """-> (callable network on the GPU, multiple its input sizes must have)"""
    from .compat import load_checkpoint
    ck = load_checkpoint(path)
    if isinstance(ck, dict) and ck.get("backbone") == "mprnet":
        from .mprnet_hip import MPRNetHip
        net = MPRNetHip(seed=0)
        net.load_state_dict(ck["Tnet"])
        return net, 4
    tn = ck["Tnet"]
    if isinstance(tn, dict):                                          # a plain state_dict
        from .compat import shim
        tn = shim().T_net.from_state_dict(tn, decoder=True)
    return tn, 8


def restore(net, x: torch.Tensor, tile: int = 0, overlap: int = 32, mult: int = 8) -> torch.Tensor:
    """``net(x)`` whole (tile 0), or as overlapping tiles averaged where they overlap"""
    _, _, H, W = x.shape
    if not tile or (tile >= H and tile >= W):
        return net(x)
    tile = max(mult, tile // mult * mult)
    step = max(mult, (tile - overlap) // mult * mult)
    acc, cnt = torch.zeros_like(x), torch.zeros(1, 1, H, W, device=x.device)
    ys = sorted({min(y, max(H - tile, 0)) for y in range(0, H, step)})
    xs = sorted({min(c, max(W - tile, 0)) for c in range(0, W, step)})
    for y0 in ys:
        for x0 in xs:
            y1, x1 = min(y0 + tile, H), min(x0 + tile, W)
            acc[:, :, y0:y1, x0:x1] += net(x[:, :, y0:y1, x0:x1].contiguous())
            cnt[:, :, y0:y1, x0:x1] += 1
    return acc / cnt


def main(argv=None):
    from PIL import Image
    from .trainer import save_image
    opt = parser.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("No GPU found: rcot_amd.tester runs the HIP path only")
    for d in (opt.save, opt.savetar, opt.saveres):
        os.makedirs(d, exist_ok=True)
    net, mult = load_network(opt.model)
    deg_list, tar_list = sorted(glob.glob(opt.degset + "*")), sorted(glob.glob(opt.tarset + "*"))
    rng = np.random.default_rng(opt.seed)
    noisy = opt.noise_sigma is not None
    done = 0
    for deg_name, tar_name in zip(deg_list, tar_list):
        name = os.path.basename(tar_name)
        print("Processing ", deg_name)
        deg, tar = np.array(Image.open(deg_name).convert("RGB")), np.array(Image.open(tar_name).convert("RGB"))
        shape1, shape2 = deg.shape, tar.shape
        h, w = deg.shape[:2]
        if noisy:
            if (h % 4) or (w % 4):                                    # tester_noise.py:84-86
                deg, tar = deg[1:h, 1:w], tar[1:h, 1:w]
        else:
            deg, tar = deg[:h - h % 4, :w - w % 4], tar[:h - h % 4, :w - w % 4]      # tester.py:77-84
        if shape1 != shape2:
            continue
        h, w = deg.shape[:2]
        if h % mult or w % mult or h == 0 or w == 0:
            print(f"  skipped: {h} x {w} is not a multiple of {mult} (the network's resampling levels)")
            continue
        x = torch.from_numpy(np.ascontiguousarray(deg.transpose(2, 0, 1))).float().div(255).unsqueeze(0)
        if noisy:
            x = x + torch.from_numpy(rng.normal(size=tar.transpose(2, 0, 1).shape) * opt.noise_sigma / 255.0).float()
        gt = torch.from_numpy(np.ascontiguousarray(tar.transpose(2, 0, 1))).float().div(255).unsqueeze(0)
        xd = x.cuda()
        out = restore(net, xd, opt.tile, opt.overlap, mult)
        res = (xd - out).cpu()
        save_image(res * (3 if noisy else 2), os.path.join(opt.saveres, name))
        save_image(out.cpu(), os.path.join(opt.save, name))
        save_image(gt, os.path.join(opt.savetar, name))
        done += 1
    print("FID value: not computed (needs a pretrained Inception network; tester.py:115-118)")
    psnr, ssim, pmax, smax, pmin, smin = evaluate_folders(opt.savetar, opt.save)
    print("PSNR: Averyge {:.5f},   best {:.5f},   worst {:.5f}".format(psnr, pmax, pmin))
    print("SSIM: Averyge {:.5f},   best {:.5f},   worst {:.5f}".format(ssim, smax, smin))
    return dict(images=done, psnr=psnr, ssim=ssim)


if __name__ == "__main__":
    main()
