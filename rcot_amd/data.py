"""Folder-driven training data with the contract of the reference's ``TrainDataset`` + ``DataLoader``
(util/dataset_utils.py:27-281, trainer.py:132-135): batches ``([names, de_id], degraded, clean)`` of fp32 CHW patches in
[0, 1].  The host keeps what needs a file system and an image decoder — the sample lists (:63-169, same list files,
same replication factors, same clean/ground-truth naming rules :198-213) and PIL decoding — and hands every decoded uint8
image to ONE device kernel (``rcot_patch_prep``) that crops, applies the dihedral augmentation, adds the Gaussian noise of
the denoise_* tasks and converts to CHW float (rcot_amd/csrc/dataprep.hip).  The reference does those steps with
PIL/numpy on the host at ``num_workers=0`` (trainer.py:32,134).

Randomness: the reference leaves python's ``random`` and numpy unseeded; here every draw (epoch
shuffle, crop origin, augmentation mode 1..7, noise seed) comes from one ``random.Random(seed, epoch)`` stream indexed by
the GLOBAL sample position, so a run is reproducible and the union of the ranks' shards does not depend on the world size.
"""
from __future__ import annotations

import os
import random
from typing import List, Sequence

import numpy as np
import torch

DE_DICT = {"denoise_15": 0, "denoise_25": 1, "denoise_50": 2, "derain": 3, "dehaze": 4, "deblur": 5, "lowlight": 6,
           "single": 7}                                   # util/dataset_utils.py:40
NOISE_SIGMA = {0: 15.0, 1: 25.0, 2: 50.0}                  # util/degradation_utils.py:29-40


def crop_to_multiple(img: np.ndarray, base: int = 16) -> np.ndarray:
    """util/image_utils.py:59-64 crop_img: centre-crop H and W to multiples of ``base``."""
    h, w = img.shape[0], img.shape[1]
    ch, cw = h % base, w % base
    return img[ch // 2:h - ch + ch // 2, cw // 2:w - cw + cw // 2, :]


def rain_gt_name(rainy_name: str) -> str:
    """util/dataset_utils.py:198-200."""
    return rainy_name.split("rainy")[0] + "gt/norain-" + rainy_name.split("rain-")[-1]


def nonhazy_name(hazy_name: str) -> str:
    """util/dataset_utils.py:202-207."""
    dir_name = hazy_name.split("synthetic")[0] + "original/"
    name = hazy_name.split("/")[-1].split("_")[0]
    return dir_name + name + "." + hazy_name.split(".")[-1]


def build_sample_ids(args) -> List[dict]:
    """The reference's ``_init_ids`` + ``_merge_ids`` (util/dataset_utils.py:63-228): one dict per sample with the file of
    the (degraded or clean) image, the task label and, for paired tasks, the ground-truth file."""
    de_type = list(args.de_type)
    ids: List[dict] = []
    if any(t in de_type for t in ("denoise_15", "denoise_25", "denoise_50")):
        ref = os.path.join(args.data_file_dir, "noisy/denoise.txt")
        listed = set(l.strip() for l in open(ref))
        clean = [args.denoise_dir + n for n in sorted(os.listdir(args.denoise_dir)) if n.strip() in listed]
        for t, lab in (("denoise_15", 0), ("denoise_25", 1), ("denoise_50", 2)):
            if t in de_type:
                ids += [{"file": c, "de": lab, "gt": None} for c in clean] * 5          # :87-101 (x5)
    if "derain" in de_type:
        rs = os.path.join(args.data_file_dir, "rainy/rainTrain.txt")
        files = [args.derain_dir + l.strip() for l in open(rs)]
        ids += [{"file": f, "de": 3, "gt": rain_gt_name(f)} for f in files] * 360       # :122-127 (x360)
    if "dehaze" in de_type:
        hz = os.path.join(args.data_file_dir, "hazy/hazy_outside.txt")
        files = [args.dehaze_dir + l.strip() for l in open(hz)]
        ids += [{"file": f, "de": 4, "gt": nonhazy_name(f)} for f in files]             # :106-116
    for t, lab, attr, sub_d, sub_c, rep in (("deblur", 5, "deblur_dir", "blur/", "sharp/", 5),
                                             ("lowlight", 6, "lowlight_dir", "low/", "high/", 20),
                                             ("single", 7, "single_dir", "degraded/", "target/", 5)):
        if t in de_type:
            root = getattr(args, attr, None)
            if root is None:
                raise SystemExit(f"--de_type {t} needs --{attr} (the reference never defines that flag either: "
                                 f"util/dataset_utils.py:134-169 reads args.{attr})")
            names = sorted(os.listdir(os.path.join(root, sub_c if t == "deblur" else sub_d)))
            ids += [{"file": os.path.join(root, sub_d, n), "de": lab, "gt": os.path.join(root, sub_c, n)} for n in names] * rep
    return ids


def _read_rgb(path: str) -> np.ndarray:
    from PIL import Image
    return np.array(Image.open(path).convert("RGB"))


class FolderLoader:
    """Iterable over one epoch of shuffled batches (``len`` = batches per epoch of the GLOBAL batch size), sharded over
    ranks by global sample position.  Yields device tensors."""

    def __init__(self, args, local_batch: int, seed: int = 0, rank: int = 0, world: int = 1, backend=None, threads: int = 0):
        """``threads`` (the reference's --threads / DataLoader num_workers, trainer.py:32,134): decode workers.  The files of
        the NEXT batches are read and decoded in a thread pool while the current iteration runs (PIL releases the GIL while it
        decodes); 0 still prefetches with one worker — decoding never sits on the training thread's critical path."""
        self.args, self.B, self.P = args, local_batch, args.patch_size
        self.threads = max(1, int(threads or 0))
        self.depth = 2                                                      # batches decoded ahead
        self.seed, self.rank, self.world = seed, rank, world
        self.ids = build_sample_ids(args)
        if not self.ids:
            raise SystemExit("no training samples found: check --de_type and the *_dir / --data_file_dir flags")
        if rank == 0:
            print(f"...total sample ids: {len(self.ids)}")                 # util/dataset_utils.py:228
        if backend is None:
            from .ops import default_backend
            backend = default_backend()
        self.be = backend
        self.epoch = 0

    def __len__(self):
        g = self.B * self.world
        if self.world > 1:
            return len(self.ids) // g           # every rank must take part in every all-reduce: the ragged tail is dropped
        return (len(self.ids) + g - 1) // g                                  # DataLoader(drop_last=False)

    def set_epoch(self, epoch: int):
        """the next __iter__ yields epoch ``epoch`` (1-based): shuffle order, crops, augmentations and noise seeds are functions of
        (seed, epoch), so a run resumed at --start_epoch continues the stream instead of replaying epoch 1"""
        self.epoch = int(epoch) - 1

    @staticmethod
    def _decode(sid: dict):
        """worker thread: read + decode + crop to a multiple of 16 (util/image_utils.py:59-64)"""
        img = crop_to_multiple(_read_rgb(sid["file"]), 16)
        gt = crop_to_multiple(_read_rgb(sid["gt"]), 16) if sid["gt"] is not None else None
        return np.ascontiguousarray(img), (None if gt is None else np.ascontiguousarray(gt))

    def _sample(self, rng: random.Random, sid: dict, deg_out, clean_out, decoded=None):
        P = self.P
        img, gt = decoded if decoded is not None else self._decode(sid)
        H, W = img.shape[0], img.shape[1]
        if H < P or W < P or (gt is not None and gt.shape != img.shape):
            raise ValueError(f"{sid['file']}: {H}x{W} is smaller than the {P}x{P} patch or differs from its ground truth")
        y0, x0 = rng.randint(0, H - P), rng.randint(0, W - P)
        mode = rng.randint(1, 7)                                            # random_augmentation: always 1..7
        nseed = rng.getrandbits(63)
        dev = self.be.device
        a = torch.from_numpy(img).to(dev, non_blocking=True)
        if gt is None:        # denoise_*: the file IS the clean image, the degradation is synthetic noise
            self.be.patch_prep(a, None, y0, x0, P, mode, NOISE_SIGMA[sid["de"]], nseed, deg_out, clean_out)
        else:
            g = torch.from_numpy(gt).to(dev, non_blocking=True)
            self.be.patch_prep(g, a, y0, x0, P, mode, 0.0, nseed, deg_out, clean_out)

    def __iter__(self):
        from concurrent.futures import ThreadPoolExecutor
        self.epoch += 1
        order = list(range(len(self.ids)))
        random.Random(self.seed * 1_000_003 + self.epoch).shuffle(order)      # DataLoader(shuffle=True)
        g = self.B * self.world
        dev = self.be.device
        nb = len(self)
        batch_idx = lambda it: order[it * g + self.rank * self.B:it * g + self.rank * self.B + self.B]
        with ThreadPoolExecutor(max_workers=self.threads) as pool:
            pending = {}                                                    # batch -> futures of its decoded files

            def submit(it):
                if it < nb and it not in pending:
                    pending[it] = [pool.submit(self._decode, self.ids[k]) for k in batch_idx(it)]
            for it in range(min(self.depth, nb)):
                submit(it)
            for it in range(nb):
                lo = it * g + self.rank * self.B
                idx = batch_idx(it)
                if not idx:
                    break
                submit(it)
                futs = pending.pop(it)
                submit(it + self.depth)                                     # keep the pool busy while this batch is prepared and trained
                n = len(idx)
                deg = torch.empty(n, 3, self.P, self.P, dtype=torch.float32, device=dev)
                clean = torch.empty_like(deg)
                names, labels = [], []
                for j, k in enumerate(idx):
                    sid = self.ids[k]
                    rng = random.Random((self.seed * 1_000_003 + self.epoch) * 2_147_483_659 + lo + j)
                    self._sample(rng, sid, deg[j], clean[j], futs[j].result())
                    names.append(os.path.basename(sid["gt"] or sid["file"]).split(".")[0])
                    labels.append(sid["de"])
                yield ([names, torch.tensor(labels)], deg, clean)
