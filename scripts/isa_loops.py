"""Static instruction mix of the loops that contain MFMAs in the compiled kernels (no GPU): per loop the number of MFMA, VALU, SALU,
LDS, global/buffer instructions, waits and barriers, and the most frequent VALU opcodes.
  python scripts/isa_loops.py rcot_amd/csrc/conv_ops.hip 'conv_fwd_lean' [extra hipcc flags]
On gfx950 the VALU instructions of a SIMD do not overlap its MFMAs (scripts/micro/lds_mfma_loop.hip): a slab loop's ceiling is
MFMA cycles / (MFMA cycles + 4 x VALU instructions); `s_waitcnt vmcnt(0)` inside such a loop means its loads have no lookahead."""
import os, re, subprocess, sys, tempfile
from collections import Counter


def loops(asm, pat):
    lines = asm.split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z[A-Za-z0-9_]+:\s*(;.*)?$", l) and re.search(pat, l)]
    for st in starts:
        end = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
        seg = lines[st:end]
        labels = {m.group(1): i for i, l in enumerate(seg) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        print(lines[st].split(":")[0])
        for i, l in enumerate(seg):
            m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
            if not (m and m.group(1) in labels and labels[m.group(1)] < i):
                continue
            body = seg[labels[m.group(1)]:i + 1]
            n = Counter()
            for b in body:
                mm = re.match(r"\s+([a-z_0-9]+)", b)
                if not mm:
                    continue
                op = mm.group(1)
                key = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "wait" if op.startswith("s_waitcnt") else
                       "barrier" if op.startswith("s_barrier") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else
                       "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "other")
                n[key] += 1
            if n["mfma"]:
                valu = Counter(re.match(r"\s+([a-z_0-9]+)", b).group(1) for b in body if re.match(r"\s+v_(?!mfma)", b))
                waits = sorted({re.search(r"vmcnt\((\d+)\)", b).group(1) for b in body if "vmcnt(" in b}, key=int)
                print(f"   loop of {len(body)} lines: {dict(n)}  vmcnt waits: {waits}")
                print(f"      VALU: {valu.most_common(8)}")


def main():
    src, pat, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--cuda-device-only", "-S",
               os.path.abspath(src), "-o", out] + extra
        subprocess.run(cmd, check=True, cwd=os.path.dirname(os.path.abspath(src)) or ".")
        loops(open(out).read(), pat)


if __name__ == "__main__":
    main()
