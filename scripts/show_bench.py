"""print the essentials of a bench.py JSON line (last line of the file given)"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], d["unit"], "ms/step", d["ms_per_step"], "|", d["dtype"], "| host", d["extra"]["host_enqueue_ms_per_step"], d["extra"]["launch_mode"])
pick = lambda q: {k: q[k] for k in ("kernel", "launches_per_step", "ms_per_step", "share_of_gpu_time", "achieved", "unit", "frac", "bound")}
print(" roofline:", pick(r))
print(" path:", {k: r["path"][k] for k in ("ms", "ms_eager", "ms_plan_replay", "plan_host_enqueue_ms", "launches", "hbm_frac", "mfma_frac")})
for row in r["per_shape"][:6]:
    print("   ", row)
for o in (d["extra"].get("other_prec") or {}).values():
    print("other:", o["prec"], o["ms_per_step"], "ms/step", o["patches_per_s"], "patches/s")
    print(" roofline:", pick(o["roofline"]))
    print(" path:", {k: o["roofline"]["path"][k] for k in ("ms", "ms_eager", "ms_plan_replay", "hbm_frac")})
    for row in o["roofline"]["per_shape"][:8]:
        print("   ", row)
    for row in o["roofline"]["next_symbols"][:6]:
        print("   next:", row)
print("comm:", d.get("comm"))
