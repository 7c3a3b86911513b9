# rocprofv3 --kernel-trace --stats of `bench.py --steps 4 --warmup 1 $BENCH_ARGS` -> gpurun_out/kstats${TAG}.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof${TAG}
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof${TAG} -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline $BENCH_ARGS > $GRAFT_REPO_ROOT/gpurun_out/prof_bench${TAG}.log 2>&1
cd $GRAFT_REPO_ROOT
TAG=$TAG python - <<'PY'
import sqlite3, glob, os
tag = os.environ.get("TAG", "")
db = sorted(glob.glob(f"gpurun_out/prof{tag}/**/*.db", recursive=True))[-1]
c = sqlite3.connect(db)
rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
tot = sum(r[2] for r in rows)
with open(f"gpurun_out/kstats{tag}.txt", "w") as f:
    # top_kernels reports durations in microseconds on this rocprofv3 (7.2)
    f.write(f"# total kernel time {tot/1e3/7:.1f} ms/step (7 iterations profiled: the eager warm-up pass and the recording pass of the launch plan, 4 timed replays, 1 host-enqueue probe; durations of kernels that overlap on the side stream both count); columns: kernel | calls | total ms | avg us | %\n")
    for r in rows[:70]:
        f.write(f"{r[0][:170]} | {r[1]} | {r[2]/1e3:.1f} | {r[3]:.2f} | {r[4]:.2f}\n")
print(open(f"gpurun_out/kstats{tag}.txt").read()[:300])
PY
rm -rf gpurun_out/prof${TAG}
