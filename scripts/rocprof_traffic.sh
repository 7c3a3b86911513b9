# HBM traffic per launch: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (TCC slots), --kernel-trace only, over a TINY
# script (PMC_SCRIPT, default pmc_stencil.py; counters serialise every dispatch).  -> gpurun_out/pmc_traffic${TAG}.txt
# Units: KiB per dispatch as rocprofv3 reports them; on gfx950 FETCH_SIZE counts wide coalesced reads at half their bytes
# (MI355X_MICROARCH.md, HBM section): the report prints the corrected value 2 x FETCH_SIZE next to the raw one.
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_$c
  timeout 240 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o run -- python $GRAFT_REPO_ROOT/scripts/${PMC_SCRIPT:-pmc_stencil.py} > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
TAG=$TAG python - <<'PY'
import sqlite3, glob, os, re, collections
tag = os.environ.get("TAG", "")
agg = collections.defaultdict(lambda: {"n": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "dur": 0.0})
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    db = sorted(glob.glob(f"gpurun_out/pmc_{cname}/**/*.db", recursive=True))[-1]
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, kernel_name, sum(value), max(duration) from counters_collection where counter_name=? "
                     "group by dispatch_id", (cname,)).fetchall()
    for did, kn, v, dur in rows:
        if "at::native" in kn or "elementwise" in kn:
            continue
        k = re.sub(r".anonymous namespace.::|^void ", "", kn)[:110]
        a = agg[k]
        a[cname] += v
        if cname == "FETCH_SIZE":
            a["n"] += 1
            a["dur"] += dur
with open(f"gpurun_out/pmc_traffic{tag}.txt", "w") as f:
    f.write("# kernel | launches | avg us (under the counter pass) | FETCH_SIZE KiB/launch | corrected read MiB/launch (2 x FETCH) | WRITE_SIZE KiB/launch\n")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["dur"]):
        n = max(a["n"], 1)
        f.write(f"{k} | {a['n']} | {a['dur'] / n / 1e3:.1f} | {a['FETCH_SIZE'] / n:.1f} | {2 * a['FETCH_SIZE'] / n / 1024:.2f} | {a['WRITE_SIZE'] / n:.1f}\n")
print(open(f"gpurun_out/pmc_traffic{tag}.txt").read()[:3000])
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
