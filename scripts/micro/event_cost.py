"""What a cross-stream hand-over costs the stream that records it (round 6).  A chain of N kernels on stream A, after each of
which stream B is made to wait for A (then runs one kernel of its own), timed against the same chain with no hand-over:
  torch        : B.wait_stream(A)  — torch.cuda.Event from torch's pool (hipEventDisableTiming)
  nofence      : raw HIP events created with hipEventDisableTiming | hipEventDisableSystemFence
  todevice     : ... | hipEventReleaseToDevice
  ext          : no record at all — the event (nofence flags) rides on the kernel's own dispatch as its stop event
                 (rcot_completion_event -> hipExtLaunchKernelGGL), B waits for it
The kernels are rcot_fill launches on a ~64 KiB / ~64 MiB tensor (short / long)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rcot_amd.ops import HipBackend
be = HipBackend()
hip = ctypes.CDLL("libamdhip64.so")
hipEventDisableTiming, hipEventReleaseToDevice, hipEventDisableSystemFence = 0x2, 0x40000000, 0x20000000
A, B = torch.cuda.Stream(), torch.cuda.Stream()
def mk(flags, n):
    evs = []
    for _ in range(n):
        e = ctypes.c_void_p()
        assert hip.hipEventCreateWithFlags(ctypes.byref(e), ctypes.c_uint(flags)) == 0
        evs.append(e)
    return evs
def run(mode, n, ta, tb, evs=None):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream(A):
            if mode == "ext":
                be.L.rcot_completion_event(evs[i])      # (prototype entry point of round 6, not in the library: see profiles/r06_event_cost.txt)
            be.fill(ta, 1.0)
            if mode == "ext":
                be.L.rcot_completion_event(None)
        if mode == "ext":
            hip.hipStreamWaitEvent(ctypes.c_void_p(B.cuda_stream), evs[i], 0)
        elif mode == "torch":
            B.wait_stream(A)
        elif mode != "none":
            e = evs[i]
            hip.hipEventRecord(e, ctypes.c_void_p(A.cuda_stream))
            hip.hipStreamWaitEvent(ctypes.c_void_p(B.cuda_stream), e, 0)
        if mode != "none":
            with torch.cuda.stream(B):
                be.fill(tb, 2.0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
N = 400
for label, nel in (("256 MiB fills (~45 us each: the host stays ahead)", 64 << 20), ("512 MiB fills", 128 << 20)):
    ta, tb = torch.empty(nel, device="cuda"), torch.empty(16384, device="cuda")
    sets = {"nofence": mk(hipEventDisableTiming | hipEventDisableSystemFence, N), "todevice": mk(hipEventDisableTiming | hipEventReleaseToDevice, N),
            "plain": mk(hipEventDisableTiming, N), "ext": mk(hipEventDisableTiming | hipEventDisableSystemFence, N)}
    print(label)
    for rep in range(2):
        base = run("none", N, ta, tb)
        line = [f"  chain alone {base:6.2f} us/kernel;  + hand-over per kernel:"]
        for mode in ("torch", "plain", "nofence", "todevice") + (("ext",) if hasattr(be.L, "rcot_completion_event") else ()):
            t = run(mode, N, ta, tb, sets.get(mode))
            line.append(f"{mode} {t - base:+6.2f}")
        print(" ".join(line), flush=True)
