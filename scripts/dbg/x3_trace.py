"""Phase timeline of the bf16x3 K-major kernels (debug builds with -DX3_TRACE of gemm_x3w.hip, or of gemm_x3.hip with X3_OLD=1;
RCOT_LIB=build_variants/librcot_trace*.so): per
workgroup time stamps (100 MHz) at start / ring primed / loop end / stores issued / stores drained, relative to the first
workgroup's start.  X3_SHAPES picks rows of bench_x3.SHAPES."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rcot_amd import lib
from rcot_amd.ops import HipBackend
be = HipBackend()
be.prec = lib.PREC_BF16X3
L = lib.load()
_set = L.rcot_x3_set_trace if os.environ.get("X3_OLD") else L.rcot_x3w_set_trace      # which kernel file was built with -DX3_TRACE
_set.argtypes = [ctypes.c_void_p]
SH = [(8, 1024, 576, 192, True, False), (8, 1024, 192, 510, False, True), (8, 256, 384, 1152, False, False),
      (8, 4096, 510, 96, True, False), (8, 16384, 510, 96, True, False), (8, 16384, 96, 510, False, False)]
if os.environ.get("X3_SHAPES"):
    SH = [SH[int(i)] for i in os.environ["X3_SHAPES"].split(",")]
tr = torch.zeros(4096 * 64, dtype=torch.int64, device="cuda")
for (B, N, Co, Ci, ln, res) in SH:
    W = torch.randn(Co, Ci, device="cuda") * 0.1
    st, sp = be.pack_shapes(Co, Ci)
    WT, WP = torch.zeros(*st, device="cuda"), torch.zeros(*sp, device="cuda")
    lw, lb = torch.ones(Ci, device="cuda"), torch.zeros(Ci, device="cuda")
    WTf, c12 = (torch.zeros(*s_, device="cuda") for s_ in be.fold_shapes(Co, Ci))
    sp3 = None if os.environ.get("X3_OLD") else tuple(torch.zeros(*be.split_shapes(Co, Ci)[i], device="cuda") for i in (0, 1, 0))
    be.pack_weight(W, WT, WP, (lw, lb, WTf, c12), sp3)
    sets = []
    for _ in range(6):
        X = torch.randn(B, Ci, N, device="cuda"); Y = torch.empty(B, Co, N, device="cuda")
        R = torch.randn(B, Co, N, device="cuda") if res else None
        mu, rs = torch.zeros(B, N, device="cuda"), torch.ones(B, N, device="cuda")
        sets.append((X, Y, R, mu, rs))
    def run(i):
        X, Y, R, mu, rs = sets[i % len(sets)]
        be.conv1x1_fwd(W, X, Y, ln=(mu, rs, lw, lb) if ln else None, R=R, packed=(WT, WP, (WTf, c12), sp3))
    _set(None)
    for i in range(12): run(i)
    torch.cuda.synchronize()
    tr.zero_()
    _set(tr.data_ptr())
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); run(3); e.record()
    torch.cuda.synchronize()
    _set(None)
    t = tr.view(-1, 64).cpu()
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    def q(col):
        v = (t[:, col][t[:, col] > 0] - t0).float() / 100.0
        return f"{v.min():6.2f}/{v.median():6.2f}/{v.max():6.2f}" if len(v) else "   -"
    print(f"B={B} N={N} M={Co} K={Ci} ln={int(ln)} res={int(res)}: event {s.elapsed_time(e)*1e3:.1f} us, {len(t)} workgroups;"
          f" us since first start, min/med/max:\n   start {q(0)} | primed {q(1)} | loop end {q(2)} | next landed {q(3)} | stores issued {q(4)} | drained {q(5)}")
    if os.environ.get("X3_HWID"):
        hw, xcc = t[:, 6], t[:, 7] & 0xf
        cu, sh, se, simd = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 4) & 3
        key = (xcc * 8 + se) * 32 + sh * 16 + cu
        import collections
        groups = collections.defaultdict(list)
        for b in range(len(t)):
            groups[int(key[b])].append(b)
        print("   distinct CUs:", len(groups), " workgroups per CU:", sorted(collections.Counter(len(v) for v in groups.values()).items()))
        print("   first CUs -> block ids:", [v for _, v in sorted(groups.items())[:6]])
        print("   xcc of blocks 0..15:", [int(x) for x in xcc[:16]])
    if (t[:, 23] > 0).any():
        row = t[len(t) // 2]
        print("   a middle workgroup, (loop end, stores issued) per tile, us since its start:", " ".join(f"{(int(row[23 + k] - row[0])) / 100.0:.1f}" for k in range(40) if row[23 + k] > 0))
    if (t[:, 8] > 0).any():
        row = t[0]
        sl = [(int(row[8 + k] - row[0])) / 100.0 for k in range(56) if row[8 + k] > 0]
        print("   workgroup 0 stamps 8.. (us since its start):", " ".join(f"{x:.2f}" for x in sl))
