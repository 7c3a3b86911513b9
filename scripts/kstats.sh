# rocprofv3 --kernel-trace --stats of "$@" -> prints per-kernel calls / avg us (top 25)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_prof
rocprofv3 --kernel-trace --stats -d /tmp/ks_prof -o run -- "$@" > /tmp/ks_run.log 2>&1
python - <<'PY'
import sqlite3, glob
db = sorted(glob.glob("/tmp/ks_prof/**/*.db", recursive=True))[-1]
c = sqlite3.connect(db)
for r in c.execute("select name, total_calls, total_duration, average from top_kernels").fetchall()[:25]:
    print(f"{r[0][:90]:90s} | {r[1]:6d} | {r[2]/1e3:9.2f} ms | {r[3]:8.2f} us")
PY
cd $GRAFT_REPO_ROOT
