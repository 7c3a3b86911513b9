"""CPU tier: the folder-driven data contract (rcot_amd/data.py) on a generated miniature of the reference's dataset
layout — sample lists with the reference's replication factors and naming rules (util/dataset_utils.py:63-228), the
batch tuple, shuffling / sharding by global sample position, reproducibility.  Kernel layer = numpy test double."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from host_double import TorchDouble


def _png(path, h, w, seed):
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    a = np.random.Generator(np.random.PCG64(seed)).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    Image.fromarray(a).save(path)
    return a


@pytest.fixture()
def tree(tmp_path):
    r = str(tmp_path)
    den = [f"img{i}.png" for i in range(3)]
    imgs = {}
    for i, n in enumerate(den):
        imgs[n] = _png(f"{r}/Denoise/{n}", 70 + i, 90 + 2 * i, 10 + i)
    _png(f"{r}/Denoise/not_listed.png", 64, 64, 99)
    os.makedirs(f"{r}/lists/noisy"); os.makedirs(f"{r}/lists/rainy"); os.makedirs(f"{r}/lists/hazy")
    open(f"{r}/lists/noisy/denoise.txt", "w").write("\n".join(den) + "\n")
    rain = ["rainy/rain-1.png", "rainy/rain-2.png"]
    open(f"{r}/lists/rainy/rainTrain.txt", "w").write("\n".join(rain) + "\n")
    for i in (1, 2):
        _png(f"{r}/Derain/rainy/rain-{i}.png", 80, 96, 20 + i)
        _png(f"{r}/Derain/gt/norain-{i}.png", 80, 96, 30 + i)
    open(f"{r}/lists/hazy/hazy_outside.txt", "w").write("synthetic/part1/0025_0.8_0.04.png\n")
    _png(f"{r}/Dehaze/synthetic/part1/0025_0.8_0.04.png", 72, 72, 41)
    _png(f"{r}/Dehaze/original/0025.png", 72, 72, 42)
    return Namespace(de_type=["denoise_15", "denoise_50", "derain", "dehaze"], data_file_dir=f"{r}/lists/", denoise_dir=f"{r}/Denoise/",
                     derain_dir=f"{r}/Derain/", dehaze_dir=f"{r}/Dehaze/", patch_size=32), imgs


def test_sample_lists_follow_the_reference(tree):
    from rcot_amd import data as D
    args, _ = tree
    ids = D.build_sample_ids(args)
    by = lambda lab: [s for s in ids if s["de"] == lab]
    assert len(by(0)) == 15 and len(by(2)) == 15 and len(by(1)) == 0          # 3 listed files x5 per sigma; unlisted file ignored
    assert len(by(3)) == 2 * 360 and len(by(4)) == 1
    assert by(3)[0]["gt"].endswith("Derain/gt/norain-1.png") and by(4)[0]["gt"].endswith("Dehaze/original/0025.png")
    assert D.rain_gt_name("a/rainy/rain-100.png") == "a/gt/norain-100.png"
    assert D.nonhazy_name("d/synthetic/part1/0025_0.8_0.04.jpg") == "d/original/0025.jpg"
    img = np.zeros((70, 93, 3), np.uint8)
    assert D.crop_to_multiple(img, 16).shape == (64, 80, 3)


def test_loader_contract_sharding_and_reproducibility(tree):
    from rcot_amd import data as D
    args, imgs = tree
    args.de_type = ["denoise_15", "dehaze"]
    be = TorchDouble(torch.float32)
    one = D.FolderLoader(args, 4, seed=7, rank=0, world=1, backend=be)
    assert len(one) == 4                                                    # 16 samples / 4, ragged tails kept at world 1
    batches = list(one)
    ([names, de_id], deg, clean) = batches[0]
    assert tuple(deg.shape) == (4, 3, 32, 32) and deg.dtype == torch.float32 and len(names) == 4 and de_id.dtype == torch.int64
    assert float(clean.min()) >= 0 and float(clean.max()) <= 1
    q = (clean * 255).round()
    assert float((q / 255 - clean).abs().max()) < 1e-6                       # uint8-quantised values / 255 (ToTensor)
    for b in range(4):
        d = (deg[b] - clean[b]) * 255
        if int(de_id[b]) == 4:
            assert float(d.abs().max()) > 1                                   # a different (paired) image
        else:
            assert 5 < float(d.std()) < 25                                    # sigma 15 noise, clipped and quantised
    # two ranks see disjoint halves of the same global batches; a second loader with the same seed repeats them
    r0 = list(D.FolderLoader(args, 2, seed=7, rank=0, world=2, backend=be))
    r1 = list(D.FolderLoader(args, 2, seed=7, rank=1, world=2, backend=be))
    assert len(r0) == len(r1) == 4
    assert r0[0][0][0] + r1[0][0][0] == names
    assert torch.equal(torch.cat([r0[0][2], r1[0][2]]), clean)
    again = list(D.FolderLoader(args, 4, seed=7, backend=be))
    assert torch.equal(again[0][2], clean) and torch.equal(again[0][1], deg)
    other = list(D.FolderLoader(args, 4, seed=8, backend=be))
    assert not torch.equal(other[0][2], clean)


# ---- the same contract against fixtures made by the REFERENCE's own util/ code (oracle/pin_against_reference.py --only data)
def _mini_tree(r):
    """the miniature dataset the fixture was made on (same seeds, same files)"""
    for i in range(3):
        _png(f"{r}/Denoise/img{i}.png", 70 + i, 90 + 2 * i, 10 + i)
    _png(f"{r}/Denoise/not_listed.png", 64, 64, 99)
    for d_ in ("noisy", "rainy", "hazy"):
        os.makedirs(f"{r}/lists/{d_}")
    open(f"{r}/lists/noisy/denoise.txt", "w").write("\n".join(f"img{i}.png" for i in range(3)) + "\n")
    open(f"{r}/lists/rainy/rainTrain.txt", "w").write("rainy/rain-1.png\nrainy/rain-2.png\n")
    for i in (1, 2):
        _png(f"{r}/Derain/rainy/rain-{i}.png", 80, 96, 20 + i)
        _png(f"{r}/Derain/gt/norain-{i}.png", 80, 96, 30 + i)
    open(f"{r}/lists/hazy/hazy_outside.txt", "w").write("synthetic/part1/0025_0.8_0.04.png\n")
    _png(f"{r}/Dehaze/synthetic/part1/0025_0.8_0.04.png", 72, 72, 41)
    _png(f"{r}/Dehaze/original/0025.png", 72, 72, 42)
    for n in ("a.png", "b.png"):
        _png(f"{r}/Single/degraded/{n}", 48, 48, 50)
        _png(f"{r}/Single/target/{n}", 48, 48, 51)
    return Namespace(de_type=["denoise_15", "denoise_50", "derain", "dehaze", "single"], data_file_dir=f"{r}/lists/",
                     denoise_dir=f"{r}/Denoise/", derain_dir=f"{r}/Derain/", dehaze_dir=f"{r}/Dehaze/", single_dir=f"{r}/Single/",
                     patch_size=32)


def test_augmentation_crop_and_noise_rule_vs_reference_fixture(gold):
    from rcot_amd import data as D
    fx = gold("data_contract.npz")
    patch = torch.from_numpy(fx["aug_in"])
    be = TorchDouble(torch.float32)
    Pz = patch.shape[0]
    for mode in range(8):                                               # data_augmentation, util/image_utils.py:133-163
        d, c = torch.empty(3, Pz, Pz), torch.empty(3, Pz, Pz)
        be.patch_prep(patch, patch, 0, 0, Pz, mode, 0.0, 1, d, c)
        want = torch.from_numpy(fx["aug_out"][mode]).permute(2, 0, 1).float() / 255.0
        assert torch.equal(c, want) and torch.equal(d, want), mode
    for h, w, ch, cw, first, last in fx["crop"]:                         # crop_img, util/image_utils.py:59-64
        im = np.arange(h * w * 3, dtype=np.int64).reshape(h, w, 3)
        c = D.crop_to_multiple(im, 16)
        assert c.shape[:2] == (ch, cw) and int(c[0, 0, 0]) == first and int(c[-1, -1, 2]) == last
    for k, sigma in enumerate((15.0, 25.0, 50.0)):                       # util/degradation_utils.py:21-27 for a fixed noise field
        mine = np.clip(fx["aug_in"] + fx["noise"] * sigma, 0, 255).astype(np.uint8)
        assert np.array_equal(mine, fx["noise_out"][k])
        assert D.NOISE_SIGMA[k] == sigma


def test_sample_lists_and_getitem_vs_reference_fixture(gold, tmp_path):
    from rcot_amd import data as D
    fx = gold("data_contract.npz")
    r = str(tmp_path)
    args = _mini_tree(r)
    mine = sorted((os.path.relpath(s["file"], r), int(s["de"])) for s in D.build_sample_ids(args))
    assert [m[0] for m in mine] == [str(f) for f in fx["ids_files"]] and [m[1] for m in mine] == fx["ids_de"].tolist()
    assert [D.rain_gt_name(str(n)) for n in fx["gt_in"][:2]] == [str(v) for v in fx["gt_rain"]]
    assert [D.nonhazy_name(str(n)) for n in fx["gt_in"][2:]] == [str(v) for v in fx["gt_hazy"]]
    # the reference's __getitem__ of one derain and one dehaze sample, with ITS crop origin and augmentation mode
    be = TorchDouble(torch.float32)
    ids = D.build_sample_ids(args)
    for k, (de, y0, x0, mode) in enumerate(fx["item_meta"].tolist()):
        sid = next(s for s in ids if os.path.relpath(s["file"], r) == str(fx["item_file"][k]))
        assert sid["de"] == de
        img, gt = D.FolderLoader._decode(sid)
        d, c = torch.empty(3, 32, 32), torch.empty(3, 32, 32)
        be.patch_prep(torch.from_numpy(gt), torch.from_numpy(img), y0, x0, 32, mode, 0.0, 1, d, c)
        assert torch.equal(d, torch.from_numpy(fx["item_deg"][k]).permute(2, 0, 1).float() / 255.0)
        assert torch.equal(c, torch.from_numpy(fx["item_clean"][k]).permute(2, 0, 1).float() / 255.0)
