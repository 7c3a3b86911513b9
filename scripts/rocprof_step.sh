# rocprofv3 --kernel-trace --stats of `bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline $BENCH_ARGS` -> gpurun_out/kstats${TAG}.txt
# Round 5: the statistics cover the REPLAYED iterations only.  The process runs 2 + 1 + 8 + 1 iterations: the eager warm-up pass and
# the recording pass of the launch plan (both inside the first warm-up call), one warm-up replay, 8 timed replays, one host-enqueue
# probe.  An iteration ends with its third optimizer launch (rmsprop_kernel: critic, gradient penalty, generator); every dispatch that
# starts after the sixth one is a replayed launch.  Per-step figures = totals / replayed iterations.
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof${TAG}
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof${TAG} -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline $BENCH_ARGS > $GRAFT_REPO_ROOT/gpurun_out/prof_bench${TAG}.log 2>&1
cd $GRAFT_REPO_ROOT
TAG=$TAG BENCH_ARGS="$BENCH_ARGS" python - <<'PY'
import sqlite3, glob, os, collections
tag = os.environ.get("TAG", "")
db = sorted(glob.glob(f"gpurun_out/prof{tag}/**/*.db", recursive=True))[-1]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
opt = [r for r in rows if "rmsprop_kernel" in r[0] or "adam_kernel" in r[0]]
assert len(opt) >= 9 and len(opt) % 3 == 0, len(opt)
t0 = opt[5][2]                                   # end of the recording pass's last optimizer launch
iters = len(opt) // 3 - 2
agg = collections.defaultdict(lambda: [0, 0.0])
for name, s, e in rows:
    if s <= t0:
        continue
    a = agg[name]
    a[0] += 1
    a[1] += (e - s) * 1e-3                       # ns -> us
tot = sum(a[1] for a in agg.values())
span = (rows[-1][2] - t0) * 1e-6 / iters
with open(f"gpurun_out/kstats{tag}.txt", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace of bench.py --steps 8 --warmup 2 {os.environ.get('BENCH_ARGS', '')}: the {iters} REPLAYED iterations only (launch-plan replays; the eager "
            f"warm-up pass and the recording pass are cut off at the sixth optimizer launch).  Kernel time {tot / 1e3 / iters:.1f} ms per step — kernels that overlap "
            f"on the side stream both count — over a GPU span of {span:.1f} ms per step.  columns: kernel | launches per step | ms per step | avg us | % of kernel time\n")
    for name, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:80]:
        f.write(f"{name[:170]} | {n / iters:.1f} | {us / 1e3 / iters:.3f} | {us / n:.2f} | {100 * us / tot:.2f}\n")
print(open(f"gpurun_out/kstats{tag}.txt").read()[:600])
PY
rm -rf gpurun_out/prof${TAG}
