"""Per-kernel averages of the counters in a rocprofv3 --pmc sqlite database (any counter set).
usage: python scripts/pmc_report.py <dir-with-db> [substring-of-kernel-name ...]"""
import glob, sqlite3, sys
from collections import defaultdict
db = sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[-1]
pats = sys.argv[2:]
c = sqlite3.connect(db)
cols = lambda t: [r[1] for r in c.execute(f"pragma table_info({t})")]
views = [r[0] for r in c.execute("select name from sqlite_master where type='view'")]
rows = None
if "counters_collection" in views:
    cc = cols("counters_collection")
    kn = "kernel_name" if "kernel_name" in cc else [x for x in cc if "name" in x and "kernel" in x][0]
    cn = "counter_name" if "counter_name" in cc else "name"
    vn = "value" if "value" in cc else "counter_value"
    did = "dispatch_id" if "dispatch_id" in cc else cc[0]
    rows = c.execute(f"select {kn}, {did}, {cn}, {vn} from counters_collection").fetchall()
else:
    pe, ip, kd, ks = cols("rocpd_pmc_event"), cols("rocpd_info_pmc"), cols("rocpd_kernel_dispatch"), cols("rocpd_info_kernel_symbol")
    print("schema:", pe, ip, kd, ks, file=sys.stderr)
    kname = "kernel_name" if "kernel_name" in ks else [x for x in ks if "name" in x][0]
    rows = c.execute(f"""select s.{kname}, d.id, p.name, e.value from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
                         join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id""").fetchall()
agg = defaultdict(lambda: defaultdict(float))
ndisp = defaultdict(set)
for k, d, n, v in rows:
    if pats and not any(p in k for p in pats):
        continue
    agg[k][n] += float(v)
    ndisp[k].add(d)
for k, cs in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("GRBM_GUI_ACTIVE", 0))):
    n = len(ndisp[k])
    print(f"\n{k[:150]}   ({n} dispatches; per-dispatch averages)")
    for name, v in sorted(cs.items()):
        print(f"   {name:32s} {v / n:16.1f}")
    wc = cs.get("SQ_WAVE_CYCLES")
    if wc:
        for nm in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_INST_CYCLES_VMEM"):
            if nm in cs:
                print(f"   {nm + ' / WAVE_CYCLES':32s} {cs[nm] / wc:16.3f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs:
        # rocprofv3 sums a counter over its instances: GRBM_GUI_ACTIVE comes as the sum over the 8 XCDs (2.48 M "cycles" for a 130 us kernel
        # = 8 x 310 k), SQ_VALU_MFMA_BUSY_CYCLES as the sum over all SIMDs.  (Rounds 2-4 divided by the 8-fold GUI_ACTIVE: their
        # "MFMA pipe busy" figures are 8 x too small.)
        print(f"   MFMA pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD x 256 CU x GRBM_GUI_ACTIVE / 8 XCD)) {cs['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cs['GRBM_GUI_ACTIVE'] / 8):.3f}")
