"""Parameter inventory of the RCOT transport map (T_net) and potential (F_net).

The names and shapes reproduce the reference's ``state_dict`` contract so that
checkpoints interchange (reference: Net_Restormer.py:216-326 for T_net,
Net_Restormer.py:437-496 for F_net).  Nothing here
touches the GPU; it is the single source of truth for

* the (name, shape) table, in the reference's registration order,
* which tensors are never reached by ``forward`` (reference quirk: 20 dead
  tensors, Net_Restormer.py:232,237-241,252,263,272,287-292),
* the flat-buffer layout the HIP optimizer / gradient all-reduce use,
* a platform-independent seeded initialiser (numpy PCG64) used for golden
  fixtures, and the reference-distribution initialiser used for training.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

DIM = 48
NUM_BLOCKS = (4, 6, 6, 8)
NUM_REFINEMENT = 4
HEADS = (1, 2, 4, 8)
FFN_FACTOR = 2.66


def ffn_hidden(dim: int) -> int:
    # Net_Restormer.py:71  hidden_features = int(dim * ffn_expansion_factor)
    return int(dim * FFN_FACTOR)


@dataclass(frozen=True)
class BlockSpec:
    """One TransformerBlock (Net_Restormer.py:201-214)."""
    prefix: str
    dim: int
    heads: int

    @property
    def hidden(self) -> int:
        return ffn_hidden(self.dim)


def block_param_shapes(prefix: str, dim: int, heads: int) -> List[Tuple[str, Tuple[int, ...]]]:
    hid = ffn_hidden(dim)
    return [
        (f"{prefix}.norm1.body.weight", (dim,)),
        (f"{prefix}.norm1.body.bias", (dim,)),
        (f"{prefix}.attn.temperature", (heads, 1, 1)),
        (f"{prefix}.attn.qkv.weight", (3 * dim, dim, 1, 1)),
        (f"{prefix}.attn.qkv_dwconv.weight", (3 * dim, 1, 3, 3)),
        (f"{prefix}.attn.project_out.weight", (dim, dim, 1, 1)),
        (f"{prefix}.norm2.body.weight", (dim,)),
        (f"{prefix}.norm2.body.bias", (dim,)),
        (f"{prefix}.ffn.project_in.weight", (2 * hid, dim, 1, 1)),
        (f"{prefix}.ffn.dwconv.weight", (2 * hid, 1, 3, 3)),
        (f"{prefix}.ffn.project_out.weight", (dim, hid, 1, 1)),
    ]


def _stage(prefix: str, n: int, dim: int, heads: int):
    out = []
    for i in range(n):
        out += block_param_shapes(f"{prefix}.{i}", dim, heads)
    return out


def tnet_param_shapes() -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) in the reference's registration order (816 tensors)."""
    d = DIM
    nb = NUM_BLOCKS
    h = HEADS
    p: List[Tuple[str, Tuple[int, ...]]] = []
    p.append(("patch_embed.proj.weight", (d, 3, 3, 3)))
    p.append(("res_patch_embed.proj.weight", (d, 3, 3, 3)))
    p.append(("chnl_reduce1.weight", (64, 64, 1, 1)))
    p.append(("chnl_reduce2.weight", (128, 128, 1, 1)))
    p.append(("chnl_reduce3.weight", (256, 320, 1, 1)))
    p.append(("reduce_noise_channel_1.weight", (d, d + 64, 1, 1)))
    p += _stage("encoder_level1", nb[0], d, h[0])
    p += _stage("resencoder_level1", nb[0], d, h[0])
    p.append(("down1_2.body.0.weight", (d // 2, d, 3, 3)))
    p.append(("resdown1_2.body.0.weight", (d // 2, d, 3, 3)))
    p.append(("reduce_noise_channel_2.weight", (2 * d, 2 * d + 128, 1, 1)))
    p += _stage("encoder_level2", nb[1], 2 * d, h[1])
    p += _stage("resencoder_level2", nb[1], 2 * d, h[1])
    p.append(("down2_3.body.0.weight", (d, 2 * d, 3, 3)))
    p.append(("resdown2_3.body.0.weight", (d, 2 * d, 3, 3)))
    p.append(("reduce_noise_channel_3.weight", (4 * d, 4 * d + 256, 1, 1)))
    p += _stage("encoder_level3", nb[2], 4 * d, h[2])
    p += _stage("resencoder_level3", nb[2], 4 * d, h[2])
    p.append(("down3_4.body.0.weight", (2 * d, 4 * d, 3, 3)))
    p.append(("resdown3_4.body.0.weight", (2 * d, 4 * d, 3, 3)))
    p += _stage("latent", nb[3], 8 * d, h[3])
    p += _stage("reslatent", nb[3], 8 * d, h[3])
    p.append(("up4_3.body.0.weight", (8 * d, 4 * d, 3, 3)))
    p.append(("reduce_chan_level3.weight", (4 * d, 2 * d + 192, 1, 1)))
    p += block_param_shapes("noise_level3", 4 * d + 192, h[2])
    p += block_param_shapes("resnoise_level3", 4 * d + 192, h[2])
    p.append(("reduce_noise_level3.weight", (4 * d, 4 * d + 192, 1, 1)))
    p.append(("resreduce_noise_level3.weight", (4 * d, 4 * d + 192, 1, 1)))
    p += _stage("decoder_level3", nb[2], 4 * d, h[2])
    p.append(("up3_2.body.0.weight", (8 * d, 4 * d, 3, 3)))
    p.append(("reduce_chan_level2.weight", (2 * d, 4 * d, 1, 1)))
    p += block_param_shapes("noise_level2", 4 * d, h[2])
    p.append(("reduce_noise_level2.weight", (4 * d, 4 * d, 1, 1)))
    p += _stage("decoder_level2", nb[1], 2 * d, h[1])
    p.append(("up2_1.body.0.weight", (4 * d, 2 * d, 3, 3)))
    p += block_param_shapes("noise_level1", 2 * d, h[2])
    p.append(("reduce_noise_level1.weight", (2 * d, 2 * d, 1, 1)))
    p += _stage("decoder_level1", nb[0], 2 * d, h[0])
    p += _stage("refinement", NUM_REFINEMENT, 2 * d, h[0])
    p.append(("output.weight", (3, 2 * d, 3, 3)))
    return p


#: prefixes of tensors that exist in the state_dict but are never used by
#: T_net.forward: they receive grad None upstream, so the
#: optimizer must never touch them.
TNET_DEAD_PREFIXES = (
    "res_patch_embed.", "chnl_reduce1.", "chnl_reduce2.", "chnl_reduce3.",
    "reduce_noise_channel_1.", "reduce_noise_channel_2.", "reduce_noise_channel_3.",
    "resdown3_4.", "resnoise_level3.", "resreduce_noise_level3.",
)


def tnet_is_dead(name: str) -> bool:
    return name.startswith(TNET_DEAD_PREFIXES)


# (cin, cout, k, stride, pad, bias) for F_net.features (Net_Restormer.py:440-490)
FNET_CONVS = (
    (3, 64, 5, 1, 2, True),
    (64, 64, 4, 2, 1, True),
    (64, 128, 3, 1, 1, True),
    (128, 128, 4, 2, 1, True),
    (128, 256, 3, 1, 1, True),
    (256, 256, 4, 2, 1, True),
    (256, 512, 3, 1, 1, False),
    (512, 512, 4, 2, 1, False),
    (512, 512, 3, 1, 1, False),
    (512, 512, 4, 2, 1, False),
)


def fnet_param_shapes(patch_size: int) -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) in registration order (22 tensors; Net_Restormer.py:440-496)."""
    p: List[Tuple[str, Tuple[int, ...]]] = []
    for i, (cin, cout, k, _s, _p, bias) in enumerate(FNET_CONVS):
        p.append((f"features.{2 * i}.weight", (cout, cin, k, k)))
        if bias:
            p.append((f"features.{2 * i}.bias", (cout,)))
    num_fea = int(patch_size * patch_size / 2)     # Net_Restormer.py:493
    p.append(("fc.weight", (int(num_fea / 4), num_fea)))
    p.append(("fc.bias", (int(num_fea / 4),)))
    p.append(("fc1.weight", (64, int(num_fea / 4))))
    p.append(("fc1.bias", (64,)))
    p.append(("fc2.weight", (1, 64)))
    p.append(("fc2.bias", (1,)))
    return p


def _fan_in(shape: Tuple[int, ...]) -> int:
    if len(shape) < 2:
        return shape[0]
    rf = 1
    for s in shape[2:]:
        rf *= s
    return shape[1] * rf


def seeded_params(shapes, seed: int, kind: str) -> Dict[str, np.ndarray]:
    """Deterministic, platform-independent parameters (numpy PCG64).

    Used for golden fixtures: only outputs are stored, weights are regenerated.
    Distribution mirrors the reference's (PyTorch default kaiming-uniform(a=sqrt(5))
    == U(+-1/sqrt(fan_in)) for conv/linear weights and biases; LayerNorm 1/0 with a
    small perturbation so fixtures exercise the affine path; temperature near 1;
    F_net conv weights N(0, 0.02), Net_Restormer.py:501-503).  ``kind`` is 'T' or 'F'.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    out: Dict[str, np.ndarray] = {}
    bound_prev = None
    for name, shape in shapes:
        if name.endswith("norm1.body.weight") or name.endswith("norm2.body.weight"):
            a = 1.0 + 0.1 * rng.standard_normal(shape)
        elif name.endswith("body.bias"):
            a = 0.1 * rng.standard_normal(shape)
        elif name.endswith("temperature"):
            a = 1.0 + 0.25 * rng.standard_normal(shape)
        elif name.endswith(".bias"):
            a = rng.uniform(-bound_prev, bound_prev, size=shape)
        elif kind == "F" and name.startswith("features."):
            a = 0.02 * rng.standard_normal(shape)
            bound_prev = 1.0 / math.sqrt(_fan_in(shape))
        else:
            bound_prev = 1.0 / math.sqrt(_fan_in(shape))
            a = rng.uniform(-bound_prev, bound_prev, size=shape)
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
    return out


@dataclass(frozen=True)
class FlatLayout:
    """Flat fp32 buffer layout for parameters / gradients / optimizer state.

    ``order`` lists tensor names in buffer order, ``offset`` their element offsets
    (each 64-element aligned so vector accesses and RCCL buckets stay aligned),
    ``n_live`` the number of leading elements the optimizer steps over.
    """
    order: Tuple[str, ...]
    offset: Dict[str, int]
    shape: Dict[str, Tuple[int, ...]]
    n_live: int
    n_total: int


def _align(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


def make_layout(shapes, live_order: List[str], dead: List[str]) -> FlatLayout:
    shp = dict(shapes)
    off: Dict[str, int] = {}
    cur = 0
    for n in live_order:
        off[n] = cur
        cur = _align(cur + int(np.prod(shp[n])))
    n_live = cur
    for n in dead:
        off[n] = cur
        cur = _align(cur + int(np.prod(shp[n])))
    return FlatLayout(tuple(live_order) + tuple(dead), off, shp, n_live, cur)
