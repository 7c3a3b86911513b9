"""ctypes binding of librcot_hip.so (C ABI declared in include/rcot_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is
absent, importing/using the kernels raises immediately (``RcotLibraryError``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 25     # == RCOT_ABI_VERSION in include/rcot_hip.h (checked by tests/test_abi.py and at load time)
PREC_FP32, PREC_BF16X3, PREC_BF16X6, PREC_BF16X1 = 0, 1, 2, 3     # RCOT_PREC_* of include/rcot_hip.h
LIB_PATH = os.environ.get("RCOT_LIB") or os.path.join(_HERE, "librcot_hip.so")   # RCOT_LIB: A/B builds while tuning


class RcotLibraryError(RuntimeError):
    pass


class RcotKernelError(RuntimeError):
    pass


_f = C.c_void_p          # device pointers travel as integers
_l = C.c_long
_i = C.c_int
_fl = C.c_float
_d = C.c_double
_sz = C.c_size_t

class KmajorDesc(C.Structure):
    """rcot_kmajor_desc of include/rcot_hip.h (one product of rcot_gemm_kmajor_multi)"""
    _fields_ = [("At", _f), ("lda", _l), ("sAo", _l), ("sAi", _l), ("a_rows", _i),
                ("Bm", _f), ("ldb", _l), ("sBo", _l), ("sBi", _l),
                ("C", _f), ("ldc", _l), ("sCo", _l), ("sCi", _l),
                ("R", _f), ("ldr", _l), ("sRo", _l), ("sRi", _l),
                ("rowscale", _f), ("sSo", _l), ("sSi", _l),
                ("Zo", _i), ("Zi", _i), ("M", _i), ("K", _i)]


# name -> argtypes, mirroring include/rcot_hip.h exactly (order matters)
SIGNATURES = {
    "rcot_abi_version": [],
    "rcot_last_kernel": [_f, _i],            # char* out: a ctypes string buffer
    "rcot_debug_nt_coop": [_i],              # host-side test hook (no launch)
    "rcot_profile_begin": [],
    "rcot_profile_end": [_f, _i],            # char* out: a ctypes string buffer
    "rcot_conv1x1_fwd": [_f, _l, _f, _l, _f, _l, _i, _i, _i, _i, _f, _f, _f, _f, _f, _l, _fl, _f],
    "rcot_conv1x1_dgrad": [_f, _l, _f, _l, _f, _l, _i, _i, _i, _i, _fl, _f],
    "rcot_conv1x1_wgrad": [_f, _l, _f, _l, _f, _l, _i, _i, _i, _i, _f, _f, _f, _f, _fl, _f, _sz, _i, _f],
    "rcot_conv1x1_wgrad_slabs": [_f, _l, _f, _l, _i, _i, _i, _i, _f, _f, _f, _f, _f, _sz, _i, _f, _f, _f],   # int* S, int* ldws: HOST
    "rcot_conv1x1_dgrad_wgrad_slabs": [_f, _l, _f, _f, _l, _f, _l, _f, _l, _i, _i, _i, _i, _f, _f, _f, _f, _f, _sz, _f, _sz, _i, _f, _f, _f],   # int* S, int* ldws: HOST
    "rcot_bmm_nn": [_f, _l, _l, _l, _i, _f, _l, _l, _l, _f, _l, _l, _l, _f, _l, _l, _l, _f, _l, _l,
                    _i, _i, _i, _i, _i, _fl, _f],
    "rcot_bmm_nt": [_f, _l, _l, _l, _f, _l, _l, _l, _f, _l, _l, _l, _i, _i, _i, _i, _i, _f, _sz, _i, _f],
    "rcot_bmm_nt_slabs": [_f, _l, _l, _l, _f, _l, _l, _l, _i, _i, _i, _i, _i, _f, _sz, _i, _f, _f, _f],   # int* S, int* ldws: HOST
    "rcot_gemm_kmajor": [_f, _l, _l, _l, _i, _f, _l, _l, _l, _f, _l, _l, _l, _f, _l, _l, _l, _f, _l, _l,
                         _f, _f, _l, _i, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _fl, _f, _sz, _i, _f],
    "rcot_gemm_kmajor_stats": [_f, _l, _l, _l, _i, _f, _l, _l, _l, _f, _l, _l, _l, _f, _l, _l, _l, _f, _f, _l, _i, _i, _i, _i, _i, _f],
    "rcot_kmajor_desc_size": [],
    "rcot_gemm_kmajor_multi": [_f, _i, _i, _i, _f],         # rcot_kmajor_desc* d: a HOST ctypes array of KmajorDesc
    "rcot_pack_weight": [_f, _l, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f],
    "rcot_pack_weights": [_f, _f, _i, _f],
    "rcot_linear_fwd": [_f, _f, _f, _f, _i, _i, _i, _fl, _f, _sz, _f],
    "rcot_linear_dgrad": [_f, _f, _f, _i, _i, _i, _f, _sz, _f],
    "rcot_linear_wgrad": [_f, _f, _f, _i, _i, _i, _fl, _f],
    "rcot_conv2d_fwd": [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fl, _i, _f, _f, _fl, _f, _sz, _f],
    "rcot_conv2d_dgrad": [_f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fl, _f, _fl, _f, _sz, _f],
    "rcot_conv2d_wgrad": [_f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fl, _f, _sz, _f],
    "rcot_conv_pcm_prep": [_f, _f, _l, _i, _i, _i, _i, _i, _f],
    "rcot_conv_pcm_wgrad": [_f, _f, _l, _i, _i, _i, _i, _f, _fl, _f, _sz, _i, _f],
    "rcot_conv_pcm_merge": [_f, _l, _f, _i, _i, _i, _i, _f],
    "rcot_conv_pcm_pack": [_f, _f, _f, _i, _i, _f, _f],
    "rcot_conv_pcm": [_f, _i, _i, _f, _l, _i, _f, _i, _f, _fl, _f, _f, _l, _f, _sz, _f],   # tapoff: HOST int array
    "rcot_pixel_shuffle": [_f, _f, _l, _i, _i, _i, _f],
    "rcot_ln_stats": [_f, _f, _f, _i, _i, _i, _f],
    "rcot_ln_bwd": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _f, _l, _f],
    "rcot_ln_bwd_rows": [_i, _i, _i],
    "rcot_block_param_reduce": [_f, _f, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _f, _i, _f],   # slab_sets: HOST array
    "rcot_dwconv3x3": [_f, _f, _f, _i, _i, _i, _i, _i, _f],
    "rcot_gdfn_gate_fwd": [_f, _f, _f, _i, _i, _i, _i, _f],
    "rcot_gdfn_gate_bwd": [_f, _f, _f, _f, _f, _i, _i, _i, _i, _f],
    "rcot_gdfn_bwd": [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _f],
    "rcot_dwconv3x3_wgrad": [_f, _f, _f, _i, _i, _i, _i, _f],
    "rcot_dwconv3x3_bwd": [_f, _f, _f, _f, _f, _i, _i, _i, _i, _f],
    "rcot_row_sumsq": [_f, _f, _i, _i, _i, _l, _f],
    "rcot_attn_softmax": [_f, _i, _i, _f, _f, _f, _f, _i, _i, _i, _f],
    "rcot_attn_bwd_small": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _f],
    "rcot_attn_bwd_fused": [_f] * 13 + [_i, _i, _i, _f, _l, _f],
    "rcot_attn_core_fwd": [_f, _l, _f, _f, _l, _f, _f, _f, _f, _l, _l, _i, _i, _i, _i, _f, _sz, _f],
    "rcot_attn_core_bwd": [_f, _i, _i] + [_f] * 12 + [_i, _i, _i, _f],
    "rcot_batch_reduce": [_f, _f, _i, _l, _fl, _f],
    "rcot_lrelu_bwd": [_f, _f, _f, _l, _fl, _f],
    "rcot_bias_grad": [_f, _f, _i, _i, _i, _f],
    "rcot_axpby2d": [_f, _l, _f, _l, _f, _l, _l, _l, _fl, _fl, _f],
    "rcot_fill": [_f, _l, _fl, _f],
    "rcot_lerp": [_f, _f, _f, _f, _i, _l, _f],
    "rcot_gp_penalty": [_f, _f, _f, _f, _i, _l, _fl, _f],
    "rcot_ot_reduce": [_f, _f, _f, _f, _i, _l, _f],
    "rcot_ot_spectrum": [_f, _f, _f, _f, _f, _f, _sz, _i, _i, _i, _f],
    "rcot_ot_grad": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _l, _fl, _fl, _l, _f],
    "rcot_patch_prep": [_f, _f, _i, _i, _i, _i, _i, _i, _fl, C.c_ulonglong, _f, _f, _f],
    "rcot_rmsprop_step": [_f, _f, _f, _l, _d, _d, _d, _d, _f],
    "rcot_adam_step": [_f, _f, _f, _f, _l, _d, _d, _d, _d, _i, _d, _f],
    # the MPRNet backbone's pointwise pieces (csrc/mprnet_ops.hip)
    "rcot_prelu_fwd": [_f, _f, _f, _l, _f],
    "rcot_prelu_bwd": [_f, _f, _f, _f, _f, _l, _f, _sz, _f],
    "rcot_row_dot": [_f, _f, _f, _l, _i, _fl, _f],
    "rcot_row_scale_add": [_f, _f, _f, _f, _fl, _f, _l, _i, _f],
    "rcot_ca_gate_fwd": [_f, _f, _f, _f, _f, _i, _i, _i, _f],
    "rcot_ca_gate_bwd": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _f, _sz, _f],
    "rcot_bilinear_down2": [_f, _f, _l, _i, _i, _f],
    "rcot_bilinear_down2_bwd": [_f, _f, _l, _i, _i, _fl, _f],
    "rcot_bilinear_up2": [_f, _f, _f, _l, _i, _i, _f],
    "rcot_bilinear_up2_bwd": [_f, _f, _l, _i, _i, _f],
    "rcot_conv_weight_flip": [_f, _f, _f, _i, _i, _i, _i, _i, _f],
}

_lib = None


def load():
    """Load librcot_hip.so once.  ``import torch`` happens first so the HIP runtime
    (libamdhip64.so.7) already mapped by torch is the one our library binds to."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (maps libamdhip64 / librccl before we dlopen)
    if not os.path.isfile(LIB_PATH):
        raise RcotLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C rcot_amd/csrc`. There is no CPU/PyTorch fallback for the RCOT hot path.")
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise RcotLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RcotLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = C.c_int
    if lib.rcot_abi_version() != ABI_VERSION:
        raise RcotLibraryError("librcot_hip.so ABI version mismatch")
    _lib = lib
    return lib


EUNSUPPORTED = -3     # RCOT_EUNSUPPORTED: the entry point has no kernel for the shape (callers fall back to the general one)


def check(rc: int, what: str):
    if rc == 0:
        return
    if rc == EUNSUPPORTED:
        raise RcotKernelError(f"{what}: no kernel for this shape")
    if rc == -1:
        raise RcotKernelError(f"{what}: invalid argument (shape/alignment/null)")
    if rc == -2:
        raise RcotKernelError(f"{what}: workspace too small")
    raise RcotKernelError(f"{what}: HIP error {rc}")
