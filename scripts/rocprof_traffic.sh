# HBM traffic of the dominant launch (LN-fused 510x96 weight gradient, B=8 128x128): FETCH_SIZE and WRITE_SIZE in SEPARATE
# passes (TCC slots), kernel-trace only.  Tiny script on purpose: counters serialise every dispatch.
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_$c
  timeout 240 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o run -- python $GRAFT_REPO_ROOT/scripts/pmc_gemm.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3, glob, re, collections
out = []
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    db = sorted(glob.glob(f"gpurun_out/pmc_{cname}/**/*.db", recursive=True))[-1]
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, substr(kernel_name,1,80), sum(value), max(duration) from counters_collection "
                     "where counter_name=? and (kernel_name like '%gemm_nt_kernel%' or kernel_name like '%nt_reduce%' or kernel_name like '%gemm_xx_kernel%') "
                     "group by dispatch_id order by dispatch_id", (cname,)).fetchall()
    for did, kn, v, dur in rows:
        out.append(f"{cname} dispatch {did:4d} {dur/1e3:8.1f} us  raw={v:14.0f}  {re.sub(r'.anonymous namespace.::|^void ', '', kn)[:60]}")
open("gpurun_out/pmc_traffic_raw.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
PY
