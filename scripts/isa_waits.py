"""Static look at the compiled kernels: per kernel, how many global loads, how many s_waitcnt vmcnt and how many of those are
vmcnt(0).  A straight-line kernel whose vmcnt(0) count is close to its load count pays one memory round trip per load (loads
inside `if` blocks followed by a cross-lane use are the usual cause); a pipelined one waits with vmcnt(N > 0).
usage: python scripts/isa_waits.py [file.hip ...]   (default: every csrc/*.hip; needs hipcc, no GPU)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "rcot_amd", "csrc")
files = sys.argv[1:] or sorted(f for f in os.listdir(CS) if f.endswith(".hip"))
for f in files:
    out = f"/tmp/isa_{os.path.basename(f)}.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S",
                    "--cuda-device-only", os.path.join(CS, os.path.basename(f)), "-o", out], stderr=subprocess.DEVNULL, check=True)
    name, stats = None, {}
    for ln in open(out):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name = m.group(1)
            stats[name] = [0, 0, 0, 0, 0]
            continue
        if name is None:
            continue
        st = stats[name]
        st[4] += ln.startswith("\t") and not ln.startswith("\t.") and not ln.startswith("\t;")
        st[0] += "global_load" in ln or "buffer_load" in ln
        if "s_waitcnt" in ln and "vmcnt" in ln:
            st[1] += 1
            st[2] += "vmcnt(0)" in ln
        st[3] += "global_store" in ln
        if "s_endpgm" in ln:
            name = None
    for k, v in stats.items():
        if v[0] >= 4:
            d = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            d = re.sub(r"\(anonymous namespace\)::|^void ", "", d)[:90]
            print(f"{os.path.basename(f):18s} instr {v[4]:5d} loads {v[0]:4d} stores {v[3]:4d} waits {v[1]:4d} of which vmcnt(0) {v[2]:4d}  {d}")
