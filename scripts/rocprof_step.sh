cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3, glob
db = sorted(glob.glob("gpurun_out/prof/**/*.db", recursive=True))[-1]
c = sqlite3.connect(db)
rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
tot = sum(r[2] for r in rows)
with open("gpurun_out/kstats.txt", "w") as f:
    f.write(f"# total kernel time {tot/1e6/6:.1f} ms/step (6 iterations profiled)\n")
    for r in rows[:60]:
        f.write(f"{r[0][:170]} | {r[1]} | {r[2]/1e3:.1f} | {r[3]/1e3:.2f} | {r[4]:.2f}\n")
print(open("gpurun_out/kstats.txt").read()[:200])
PY
