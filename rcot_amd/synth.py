"""Seeded synthetic patch generator reproducing the *output contract* of the reference's
TrainDataset (util/dataset_utils.py:215-281): batches ``([names, de_id], degraded, target)`` of fp32
NCHW tensors in [0,1] obtained by /255 of uint8-quantised images (ToTensor, :264-265), with the
degradations of util/degradation_utils.py:21-40 (Gaussian noise sigma 15/25/50 for de_id 0/1/2,
``clip(clean + randn*sigma, 0, 255).astype(uint8)``) and Rain100L-/SOTS-shaped stand-ins for
de_id 3/4.  No dataset folders are shipped with the reference.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

DE_IDS = {"denoise_15": 0, "denoise_25": 1, "denoise_50": 2, "derain": 3, "dehaze": 4, "deblur": 5,
          "lowlight": 6, "single": 7}
_SIGMA = {0: 15.0, 1: 25.0, 2: 50.0}


def _clean_patch(rng: np.random.Generator, P: int) -> np.ndarray:
    """Smooth random field (a few low-frequency cosines per channel) + mild texture, uint8 HWC."""
    yy, xx = np.meshgrid(np.arange(P, dtype=np.float64), np.arange(P, dtype=np.float64), indexing="ij")
    img = np.zeros((P, P, 3))
    for c in range(3):
        acc = np.full((P, P), rng.uniform(80, 170))
        for _ in range(4):
            fy, fx = rng.uniform(0.2, 3.0, size=2) * 2 * np.pi / P
            acc += rng.uniform(10, 40) * np.cos(fy * yy + fx * xx + rng.uniform(0, 2 * np.pi))
        img[..., c] = acc
    img += rng.uniform(-6, 6, size=img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def _degrade(rng: np.random.Generator, clean: np.ndarray, de_id: int) -> np.ndarray:
    c = clean.astype(np.float64)
    P = clean.shape[0]
    if de_id in _SIGMA:                                   # util/degradation_utils.py:21-27
        return np.clip(c + rng.standard_normal(c.shape) * _SIGMA[de_id], 0, 255).astype(np.uint8)
    if de_id == 3:                                        # sparse bright anisotropic streaks
        rain = np.zeros((P, P))
        for _ in range(max(4, P // 6)):
            x0, y0, ln = rng.integers(0, P), rng.integers(0, P), rng.integers(P // 8, P // 3)
            for t in range(ln):
                y, x = y0 + t, x0 + t // 3
                if y < P and x < P:
                    rain[y, x] = max(rain[y, x], rng.uniform(60, 140))
        return np.clip(c + rain[..., None], 0, 255).astype(np.uint8)
    if de_id == 4:                                        # hazy = clean*t + A*(1-t), smooth t in [0.3,0.9]
        yy = np.linspace(0, 1, P)[:, None]
        t = np.clip(0.3 + 0.6 * (0.5 + 0.5 * np.cos(2 * np.pi * (yy * rng.uniform(0.3, 1.0) + rng.uniform()))), 0.3, 0.9)
        A = rng.uniform(200, 250)
        return np.clip(c * t[..., None] + A * (1 - t[..., None]), 0, 255).astype(np.uint8)
    if de_id == 6:
        return np.clip(c * 0.3, 0, 255).astype(np.uint8)
    return np.clip(c + rng.standard_normal(c.shape) * 25.0, 0, 255).astype(np.uint8)


def make_batch(seed: int, B: int, P: int, de_ids: Sequence[int], unpaired: bool = False):
    """Deterministic batch: returns (de_id list, degraded [B,3,P,P], target [B,3,P,P]) CPU float32."""
    rng = np.random.Generator(np.random.PCG64(seed))
    clean = [_clean_patch(rng, P) for _ in range(B)]
    deg = [_degrade(rng, clean[i], int(de_ids[i])) for i in range(B)]
    tgt = clean
    if unpaired:                                           # target = a different set of clean patches
        tgt = [_clean_patch(rng, P) for _ in range(B)]
    to_t = lambda lst: torch.from_numpy(np.stack(lst).transpose(0, 3, 1, 2).astype(np.float32) / 255.0).contiguous()
    return [int(d) for d in de_ids], to_t(deg), to_t(tgt)


class SyntheticLoader:
    """Iterable with ``len``; each rank draws its own shard of the global batch (global sample index
    determines the seed, so the union over ranks does not depend on the world size)."""

    def __init__(self, de_types: Sequence[str], local_batch: int, patch: int, iters: int, seed: int = 0,
                 rank: int = 0, world: int = 1, unpaired: bool = False):
        self.ids = [DE_IDS[t] for t in de_types]
        self.B, self.P, self.iters, self.seed, self.rank, self.world, self.unpaired = local_batch, patch, iters, seed, rank, world, unpaired
        self.epoch = 0

    def __len__(self):
        return self.iters

    def set_epoch(self, epoch: int):
        """the next __iter__ yields epoch ``epoch`` (1-based): a resumed run continues the stream"""
        self.epoch = int(epoch) - 1

    def __iter__(self):
        self.epoch += 1
        for it in range(self.iters):
            g0 = ((self.epoch * self.iters + it) * self.world + self.rank) * self.B
            de = [self.ids[(g0 + i) % len(self.ids)] for i in range(self.B)]
            # one generator stream PER SAMPLE, seeded by its global index: the union of the ranks' shards is the batch a
            # single process draws, whatever the world size
            parts = [make_batch(self.seed * 1_000_003 + g0 + i, 1, self.P, [de[i]], self.unpaired) for i in range(self.B)]
            x, y = torch.cat([p[1] for p in parts]), torch.cat([p[2] for p in parts])
            yield ([["synthetic"] * self.B, torch.tensor(de)], x, y)
