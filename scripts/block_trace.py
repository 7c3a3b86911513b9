"""Per-dispatch time line of ONE TransformerBlock forward+backward at every level of T_net (B=8, 128x128 input):
run under `rocprofv3 --kernel-trace --output-format csv`, then `python scripts/block_trace.py report <csv>` prints, per
level, every launch of the last iteration with its duration and the gap to the previous launch's end.

  rocprofv3 --kernel-trace --output-format csv -d OUT -o run -- python scripts/block_trace.py run
  python scripts/block_trace.py report OUT/**/run_kernel_trace.csv > gpurun_out/block_trace.txt

A marker launch (rcot_axpby2d on a 7-element tensor: grid of one workgroup) precedes every forward; TWO markers in a row
(an empty segment between them) separate a forward from its backward, so a segment is labelled by what surrounds it and not
by its position in the file (round 3's position-based labels came out swapped on every other level).
"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LEVELS = [("enc1 C48 h1 128x128", "enc1", 48, 128), ("dec1 C96 h1 128x128", "dec1", 96, 128), ("enc2 C96 h2 64x64", "enc2", 96, 64),
          ("enc3 C192 h4 32x32", "enc3", 192, 32), ("latent C384 h8 16x16", "latent", 384, 16),
          ("noise3 C384 h4 16x16", "noise3", 384, 16)]
ITERS = 4


def run():
    import torch
    from rcot_amd import lib
    from rcot_amd.net_restormer import T_net
    from rcot_amd.ops import default_backend
    be = default_backend()
    be.prec = {"fp32": lib.PREC_FP32, "bf16x3": lib.PREC_BF16X3, "bf16x6": lib.PREC_BF16X6}[os.environ.get("RCOT_GEMM_PREC", "bf16x3")]
    be.x6_packs = True
    Tn = T_net(decoder=True, seed=1234)
    B = int(os.environ.get("BT_BATCH", "8"))
    m7 = be.zeros(7)
    for label, attr, C, H in LEVELS:
        blk = getattr(Tn, attr)
        blk = blk[0] if isinstance(blk, list) else blk
        x = torch.randn(B, C, H, H, device="cuda")
        d = torch.randn(B, C, H, H, device="cuda")
        for _ in range(ITERS):
            be.axpby(m7, None, m7, 1.0, 0.0)
            y, ctx = blk.forward(x, True)
            be.axpby(m7, None, m7, 1.0, 0.0)
            be.axpby(m7, None, m7, 1.0, 0.0)
            blk.backward(ctx, d)
        be.axpby(m7, None, m7, 1.0, 0.0)
        torch.cuda.synchronize()


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    for pre in ("rcot_x3w::", "rcot_x3::", "rcot_nt::", "rcot::"):
        n = n.replace(pre, "")
    p = n.find("(")
    return (n[:p] if p > 0 else n)[:60]


def report(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # split at the markers
    segs, cur = [], []
    for r in rows:
        nm = r["Kernel_Name"]
        gx = int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0)
        wx = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", "1")) or 1)
        if "axpby2d" in nm and gx <= wx:
            segs.append(cur)
            cur = []
        else:
            cur.append(r)
    segs.append(cur)
    # (forward, backward) pairs: the segments on either side of an EMPTY segment (the double marker)
    pairs = [(segs[i - 1], segs[i + 1]) for i in range(1, len(segs) - 1) if not segs[i] and segs[i - 1] and segs[i + 1]]
    for li, (label, *_r) in enumerate(LEVELS):
        mine = pairs[li * ITERS:(li + 1) * ITERS]
        if len(mine) < ITERS:
            break
        for phase, seg in (("forward", mine[-1][0]), ("backward", mine[-1][1])):
            t0 = int(seg[0]["Start_Timestamp"])
            prev_end = t0
            busy = 0
            print(f"== {label} {phase}: {len(seg)} launches")
            for r in seg:
                s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                busy += e - s
                gx = int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0)
                wx = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", "1")) or 1)
                print(f"  +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  wgs {gx // max(wx, 1):5d}x{wx:<4d} {short(r['Kernel_Name'])}")
                prev_end = max(prev_end, e)
            span = (prev_end - t0) / 1e3
            print(f"   span {span:.1f} us, kernel time {busy / 1e3:.1f} us, gaps {span - busy / 1e3:.1f} us")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2])
