import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_DEFAULT_THREADS = torch.get_num_threads()


@pytest.fixture(autouse=True)
def _restore_thread_count():
    """tests that change torch's intra-op thread count (the 2-rank gloo runs) must not leak it into fixtures that were made at
    the default count (MKL-DNN reductions round differently per thread count: tests/test_mprnet_cpu.py)"""
    torch.set_num_threads(_DEFAULT_THREADS)
    yield
    torch.set_num_threads(_DEFAULT_THREADS)


def seeded_tensor(seed, shape, scale=1.0, lo=None, hi=None, dtype=torch.float32):
    """Same generator as oracle/pin_against_reference.py so fixtures' inputs can be regenerated."""
    g = np.random.Generator(np.random.PCG64(seed))
    a = g.uniform(lo, hi, size=shape) if lo is not None else scale * g.standard_normal(shape)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dtype)


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(scope="session")
def gold():
    return lambda name: np.load(os.path.join(GOLD, name), allow_pickle=False)
