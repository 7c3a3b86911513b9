import os
import sys

# Before torch (and with it libgomp / MKL) is loaded: a modest OpenMP team.  The GPU boxes are 256-thread hosts shared between pods;
# torch's default there is a 128-thread team, and every small fp64 test double becomes a 128-way fork/join that crawls as soon as
# the host is busy (round 4: 3.5 s per test on the driver's box, 1200 s limit hit).  8 threads is also the width the committed
# fixtures were made at in the build container.  (A passive wait policy — OMP_WAIT_POLICY=PASSIVE, GOMP_SPINCOUNT=0 — was tried and
# is WRONG here: the many tiny parallel regions then wake their team through futexes, 22 of 25 minutes of the CPU tier in the kernel.)
# Round 6: half the visible CPUs, at most 8.  On the 8-vCPU build container (a shared microVM: the same CPU tier took 414, 502 and 868 s at 8
# threads within one afternoon, depending on the neighbours) every fp64 double was an 8-way fork/join on 8 contended vCPUs; at 4 threads the tier
# runs in 373 s with the same results (48 passed; the thread-dependent fixtures of tests/test_mprnet_cpu.py hold their bars).  The 256-thread
# GPU hosts keep 8.
_DEFAULT_TEST_THREADS = str(min(8, max(2, (os.cpu_count() or 8) // 2)))
os.environ.setdefault("OMP_NUM_THREADS", os.environ.get("RCOT_TEST_THREADS", _DEFAULT_TEST_THREADS))
os.environ.setdefault("MKL_NUM_THREADS", os.environ["OMP_NUM_THREADS"])

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
# The fp64 host doubles (tests/host_double.py) work on small tensors: on a 256-thread GPU host torch's default intra-op width makes
# every one of them a 128-way fork/join.  Cap it ONCE per session (bench.py caps its CPU leg the same way; the environment above
# already does it unless the caller set OMP_NUM_THREADS).  Fixtures that depend on the thread count (MKL-DNN reductions:
# tests/test_mprnet_cpu.py) were made in the 8-thread build container and are CPU-tier.
# Measured on an MI355X box (gpurun_out r05a, profiles/r05_gpu_suite_tail.txt): tests/test_kernels_gpu.py 38.5 s / 9 min 14 s of CPU
# time with round 4's per-test set_num_threads(128) fixture, 10.4 s / 52 s without it at 16 threads; whole GPU tier 203 s.
_SESSION_THREADS = min(_DEFAULT_THREADS, int(os.environ.get("RCOT_TEST_THREADS", _DEFAULT_TEST_THREADS)))
if os.environ.get("RCOT_TEST_OLD_THREADS") != "1":
    torch.set_num_threads(_SESSION_THREADS)


@pytest.fixture
def restore_thread_count():
    """for the tests that change torch's intra-op thread count themselves (the 2-rank gloo run): do not leak it into tests whose
    fixtures were made at the session's count.  (Round 4 had this autouse: two torch.set_num_threads calls — thread pool torn down
    and rebuilt at full host width, mkl_set_dynamic(false) — around every one of 472 GPU tests.)"""
    n = torch.get_num_threads()
    yield
    torch.set_num_threads(n)


if os.environ.get("RCOT_TEST_OLD_THREADS") == "1":      # round 4's behaviour, kept switchable to MEASURE what it cost
    @pytest.fixture(autouse=True)
    def _restore_thread_count_r4():
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
