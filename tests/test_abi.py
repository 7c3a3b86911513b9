"""CPU tier: the C-ABI shared library loads and exports exactly the symbols include/rcot_hip.h declares
(no compute is launched without a GPU), and the product refuses to run without a HIP device."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from rcot_amd import lib


def _declared():
    src = open(os.path.join(ROOT, "include", "rcot_hip.h")).read()
    return sorted(set(re.findall(r"^int (rcot_\w+)\(", src, flags=re.M)))


def test_header_bindings_and_exports_agree():
    decl = _declared()
    assert decl == sorted(lib.SIGNATURES), set(decl) ^ set(lib.SIGNATURES)
    L = lib.load()
    raw = ctypes.CDLL(lib.LIB_PATH)
    for name in decl:
        assert hasattr(raw, name), name
    src = open(os.path.join(ROOT, "include", "rcot_hip.h")).read()
    header_version = int(re.search(r"^#define RCOT_ABI_VERSION (\d+)", src, flags=re.M).group(1))
    assert header_version == lib.ABI_VERSION == L.rcot_abi_version()      # one constant: header -> .so -> binding
    assert L.rcot_kmajor_desc_size() == ctypes.sizeof(lib.KmajorDesc)     # the one struct of the ABI: same layout on both sides


def test_signature_arity_matches_header():
    src = open(os.path.join(ROOT, "include", "rcot_hip.h")).read()
    for name, args in lib.SIGNATURES.items():
        m = re.search(r"^int %s\((.*?)\);" % name, src, flags=re.M | re.S)
        assert m, name
        params = [p for p in m.group(1).replace("\n", " ").split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == len(args), (name, len(params), len(args))
        for p, a in zip(params, args):
            p = p.strip()
            if "*" in p:
                assert a is ctypes.c_void_p, (name, p)
            elif p.startswith("double"):
                assert a is ctypes.c_double, (name, p)
            elif p.startswith("float"):
                assert a is ctypes.c_float, (name, p)
            elif p.startswith("long"):
                assert a is ctypes.c_long, (name, p)
            elif p.startswith("size_t"):
                assert a is ctypes.c_size_t, (name, p)
            elif p.startswith("int"):
                assert a is ctypes.c_int, (name, p)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rcot_amd.ops import HipBackend
    with pytest.raises(lib.RcotLibraryError):
        HipBackend()
    from rcot_amd.net_restormer import T_net
    with pytest.raises(lib.RcotLibraryError):
        T_net(decoder=True)
