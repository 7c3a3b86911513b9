"""Prints the two measurement tables of DESIGN.md section 6 from the committed profile files of a round (profiles/<round>_*):
per-level block times (small_levels_*.txt) and the largest kernels of an exact-fp32 iteration (kernel_stats_fp32.txt).
  python scripts/design_tables.py r05"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r05"
GB = {"enc1": 2005 * 8e-3, "dec1": 8037 * 8e-3, "enc2": 3265 * 8e-3, "enc3": 1633 * 8e-3, "l16": 817 * 8e-3}     # SURVEY 8d, MB per image x B = 8
N = {"enc1": 8, "dec1": 16, "enc2": 26, "enc3": 26, "latent": 16, "noise3": 10}


def levels(prec):
    out = {}
    for ln in open(os.path.join(ROOT, "profiles", f"{R}_small_levels_{prec}.txt")):
        m = re.match(r"(\w+)\s.*fwd\s+([\d.]+) us.*bwd\s+([\d.]+) us", ln)
        if m:
            out[m.group(1)] = (float(m.group(2)), float(m.group(3)))
    return out


L = {p: levels(p) for p in ("fp32", "bf16x6", "bf16x3")}
ms = lambda p, k: sum(L[p][k]) * N[k] / 1e3
print("| level (blocks per unit) | algorithmic GB | fp32: fwd + bwd µs per block → ms | of the HBM roof | bf16x6 ms | bf16x3 ms | of the roof (bf16x3) |")
print("|---|---|---|---|---|---|---|")
rows = [("48 ch @ 128² (×8)", "enc1", ["enc1"]), ("96 ch @ 128² (×16)", "dec1", ["dec1"]), ("96 ch @ 64² (×26)", "enc2", ["enc2"]),
        ("192 ch @ 32² (×26)", "enc3", ["enc3"]), ("384 ch @ 16² (×16, and ×10 with 4 heads)", "l16", ["latent", "noise3"])]
tot = {p: 0.0 for p in L}
for label, g, keys in rows:
    t = {p: sum(ms(p, k) for k in keys) for p in L}
    for p in L:
        tot[p] += t[p]
    fb = ", ".join(f"{L['fp32'][k][0]:.0f} + {L['fp32'][k][1]:.0f}" for k in keys)
    print(f"| {label} | {GB[g]:.1f} | {fb} → {t['fp32']:.1f} | {GB[g] / t['fp32'] / 8:.2f} | {t['bf16x6']:.1f} | {t['bf16x3']:.1f} | {GB[g] / t['bf16x3'] / 8:.2f} |")
G = sum(GB.values())
print(f"| sum over the 94 block applications | {G:.1f} | {tot['fp32']:.1f} | {G / tot['fp32'] / 8:.3f} | {tot['bf16x6']:.1f} | {tot['bf16x3']:.1f} | {G / tot['bf16x3'] / 8:.2f} |")
print()
lines = open(os.path.join(ROOT, "profiles", f"{R}_kernel_stats_fp32.txt")).read().splitlines()
print(lines[0][:420])
for ln in lines[1:16]:
    name, n, t, avg, pct = [x.strip() for x in ln.rsplit("|", 4)]
    name = re.sub(r"\(anonymous namespace\)::|rcot_nt::|rcot::|^void ", "", name)
    name = name[:name.find("(")] if "(" in name else name
    print(f"| `{name}` | {float(n):.0f} | {float(t):.2f} | {float(avg):.1f} |")
