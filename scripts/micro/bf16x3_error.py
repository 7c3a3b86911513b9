"""Numerical error of a bf16x3 split product (x = hi + lo, hi*hi + hi*lo + lo*hi, fp32 accumulation) on the shapes of the
transport map's projections, against fp64 and against plain fp32 accumulation.  CPU emulation (numpy)."""
import numpy as np
def bf16(x):                      # round-to-nearest-even to bfloat16, returned as float32
    u = x.astype(np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).astype(np.uint32).view(np.float32)
rng = np.random.default_rng(0)
for (M, K, N) in ((96, 96, 4096), (510, 96, 4096), (96, 510, 4096), (1020, 192, 1024), (96, 16384, 96)):
    A = (rng.standard_normal((M, K)) * 0.1).astype(np.float32)
    B = rng.standard_normal((K, N)).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    f32 = A @ B
    Ah, Bh = bf16(A), bf16(B)
    Al, Bl = bf16(A - Ah), bf16(B - Bh)
    x3 = Ah @ Bh + (Ah @ Bl + Al @ Bh)
    x1 = Ah @ Bh
    sc = np.abs(ref).max()
    print(f"M={M:5d} K={K:6d} N={N:5d}: max|err|/max|C|  fp32 {np.abs(f32 - ref).max() / sc:.2e}   bf16x3 {np.abs(x3 - ref).max() / sc:.2e}   plain bf16 {np.abs(x1 - ref).max() / sc:.2e}")
