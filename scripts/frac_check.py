"""Does the bench line's roofline.frac follow from the rocprofv3 statistics of the same call?  
  python scripts/frac_check.py <bench.json> <kstats.txt>
frac_rocprof = algorithmic work of the dominant symbol per step / (its ms per step in the replay-only kernel statistics) / peak."""
import json
import sys

line = json.load(open(sys.argv[1]))
r = line["roofline"]
sym = r["kernel"].replace(" ", "")
ms = None
for ln in open(sys.argv[2]):
    if ln.startswith("#"):
        continue
    name, n, tot, avg, pct = [x.strip() for x in ln.rsplit("|", 4)]
    if sym in name.replace(" ", ""):
        ms, calls = float(tot), float(n)
        break
assert ms is not None, sym
work_s = (r["algorithmic_gbytes_per_step"] * 1e9 / (r["peak"] * 1e9)) if r["bound"] == "hbm" else (r["algorithmic_tflop_per_step"] / r["mfma_peak_tflops"])
frac_prof = work_s / (ms * 1e-3)
print(f"dominant symbol {r['kernel']}: bench.py {r['launches_per_step']} launches, kernel_ms_per_step {r['kernel_ms_per_step']}, frac {r['frac']} "
      f"(with launch brackets: {r['frac_in_situ_brackets']});  rocprofv3 {calls:.1f} launches per step, {ms:.3f} ms per step -> frac {frac_prof:.4f};  "
      f"relative difference {abs(r['frac'] - frac_prof) / frac_prof * 100:.1f} %")
