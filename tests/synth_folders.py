"""Synthetic dataset folders for the CLI / evaluate() tests (test infrastructure; also used by oracle/pin_against_reference.py
to produce the reference PSNR of the same images)."""
import os

import numpy as np


def write_png(path, arr):
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(arr).save(path)


def dataset_tree(root, n_den=3):
    g = np.random.Generator(np.random.PCG64(1))
    smooth = lambda h, w: np.clip(128 + 60 * np.sin(np.linspace(0, 6, h))[:, None, None] * np.cos(np.linspace(0, 5, w))[None, :, None]
                                  + g.normal(0, 4, (h, w, 3)), 0, 255).astype(np.uint8)
    names = [f"c{i}.png" for i in range(n_den)]
    for n in names:
        write_png(f"{root}/Denoise/{n}", smooth(96, 112))
    os.makedirs(f"{root}/lists/noisy", exist_ok=True)
    open(f"{root}/lists/noisy/denoise.txt", "w").write("\n".join(names) + "\n")
    for i, (h, w) in enumerate(((64, 96), (48, 160), (70, 90))):               # the last is skipped by evaluate (not multiples of 8)
        t = smooth(h, w)
        write_png(f"{root}/val/target/{i}.png", t)
        write_png(f"{root}/val/input/{i}.png", np.clip(t + g.normal(0, 25, t.shape), 0, 255).astype(np.uint8))
