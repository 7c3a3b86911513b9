#!/usr/bin/env python
"""Headline benchmark: 128x128 patches/sec of the full RCOT minimax iteration (critic step +
gradient-penalty step + generator step, 3 optimizer steps) on the hand-written HIP path.

  python bench.py --gpus N --steps K --warmup W [--config 2|3|5] [--prec fp32|bf16x3]

N > 1: when the process was not started by torch.distributed.run (no WORLD_SIZE in the environment) bench.py starts its own
ranks — it re-executes itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` on
a free port and passes the ranks' output through — so both `python bench.py --gpus 8` and the torchrun form work.

A "step" is one minimax iteration over one synthetic batch already resident in HBM.
Workloads (BASELINE.json `configs`, 0-based index in brackets):
  --config 2 [1] (default, the headline): Restormer T_net(decoder=True) + F_net(128), denoise_50 (de_id 2, Parseval
             branch of the Fourier OT cost), B=8 per GPU, 128x128, RMSprop, paired L1 term on (README recipe)
  --config 3 [2]: derain (de_id 3, L1-spectrum branch: FFT in LDS), B=16 per GPU, 128x128, paired
  --config 5 [4]: dehaze (de_id 4), 256x256, F_net(256), unpaired OT (pairnum=0), B=4 per GPU
Weak scaling: every rank processes its own batch, gradients are SUM all-reduced over RCCL.  Rank 0 prints ONE JSON line.
`--prec` selects the arithmetic of the big MFMA products (include/rcot_hip.h RCOT_PREC_*): operands and results are fp32
in HBM either way.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_SYMBOLS = ("gemm_kernel<", "gemm_xx_kernel", "conv2d", "linear", "conv_few", "wgrad_few")   # exact fp32 whatever --prec says
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3       # dense fp32 MFMA (v_mfma_f32_32x32x2_f32)
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16 MFMA; a bf16x3 split product costs three of them per fp32 product
TNET_FWDBWD_BYTES_PER_PATCH = 15.946e9   # SURVEY.md 8(d) stage model, fp32, 128x128 (x4 at 256x256)
TNET_FWD_FLOP_PER_PATCH = 166.5e9

DTYPE = {"fp32": "fp32 (exact fp32 MFMA everywhere)",
         "bf16x6": "fp32-class: fp32 storage and accumulation; 1x1 weight projections as six bf16 partial products of a three-term "
                   "split (bf16x6, max error vs fp64 equal to the exact-fp32 MFMA kernel's, every parity test at the fp32 bars), "
                   "all other products exact fp32 MFMA",
         "bf16x3": "fp32 storage/accumulate, bf16x3 split-MFMA products (two-term split, ~2^-16 per product) in the 1x1 / Gram GEMMs"}

CONFIGS = {      # batch per GPU, patch, de_id, paired, unpaired targets, BASELINE.json configs[] index, label
    2: dict(B=8, P=128, de=2, paired=True, unpaired=False, idx=1, label="denoise_50, RMSprop, paired"),
    3: dict(B=16, P=128, de=3, paired=True, unpaired=False, idx=2, label="derain (L1-spectrum OT cost), RMSprop, paired"),
    5: dict(B=4, P=256, de=4, paired=False, unpaired=True, idx=4, label="dehaze, unpaired OT (pairnum=0), RMSprop"),
}


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


CPU_SNIPPET = r"""
import sys, time, json, os, torch
sys.path.insert(0, %(root)r)
from oracle import rcot_oracle as O
from rcot_amd import params as PP
from rcot_amd.synth import make_batch
threads, P, cb, lr, de, paired, unp, timed, budget = %(threads)d, %(P)d, %(cb)d, %(lr)r, %(de)d, %(paired)r, %(unp)r, %(timed)d, %(budget)r
torch.set_num_threads(threads)
pT = {k: torch.from_numpy(v) for k, v in PP.seeded_params(PP.tnet_param_shapes(), 31, "T").items()}
pF = {k: torch.from_numpy(v) for k, v in PP.seeded_params(PP.fnet_param_shapes(P), 32, "F").items()}
oT, oF = O.RMSprop(pT, lr / 2), O.RMSprop(pF, lr)
t0 = time.perf_counter()
secs = []
for it in range(1 + timed):                      # 1 warm-up + `timed` timed iterations (BASELINE.md section 4)
    _, xd, yd = make_batch(5 + it, cb, P, [de] * cb, unpaired=unp)
    t1 = time.perf_counter()
    O.minimax_iteration(pT, pF, oT, oF, xd, yd, [de] * cb, torch.full((cb, 1, 1, 1), 0.5), 1.0, 10000.0, paired)
    dt = time.perf_counter() - t1
    if it > 0:
        secs.append(dt)
    print(json.dumps({"it": it, "seconds": dt}), flush=True)
    if it > 0 and time.perf_counter() - t0 > budget:
        break
# SURVEY 8(d): the generator alone, forward and forward + backward (autograd), one warm call + one timed call each, same batch
if time.perf_counter() - t0 < budget:
    _, xd, _ = make_batch(3, cb, P, [de] * cb, unpaired=unp)
    with torch.no_grad():
        O.tnet_forward(pT, xd)
        t1 = time.perf_counter(); O.tnet_forward(pT, xd); fwd = time.perf_counter() - t1
    q = {k: v.clone().requires_grad_(True) for k, v in pT.items()}
    t1 = time.perf_counter(); O.tnet_forward(q, xd).square().mean().backward(); fb = time.perf_counter() - t1
    print(json.dumps({"gen_fwd_seconds": fwd, "gen_fwd_bwd_seconds": fb}), flush=True)
"""


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(cfg, lr, budget_s=200):
    """The oracle's full minimax iteration (critic + GP + generator steps, RMSprop) on the host cores, in its own process:
    1 warm-up + up to 3 timed iterations at B=4 (B=2 at 256x256), bounded by a wall-clock budget (at least one timed
    iteration is always kept).  threads = min(host cores, 32): MKL-DNN stops scaling beyond that on this model size;
    the count actually used and the CPU model are reported."""
    import subprocess
    cores = os.cpu_count() or 1
    threads = min(cores, 32)
    P = cfg["P"]
    cb = 4 if P <= 128 else 2
    code = CPU_SNIPPET % dict(root=ROOT, threads=threads, P=P, cb=cb, lr=lr, de=cfg["de"], paired=cfg["paired"],
                              unp=cfg["unpaired"], timed=3, budget=float(budget_s) * 0.5)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="")
    out = ""
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=budget_s, env=env)
        out = r.stdout
    except subprocess.TimeoutExpired as e:      # keep what finished
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
    recs = [json.loads(l) for l in out.strip().splitlines() if l.startswith("{")]
    its = [d for d in recs if "it" in d]
    gen = next((d for d in recs if "gen_fwd_seconds" in d), None)
    timed = [d["seconds"] for d in its if d["it"] > 0] or [d["seconds"] for d in its]
    what = (f"oracle minimax iteration (critic+GP+generator, RMSprop), B={cb}, {P}x{P}, de_id={cfg['de']}, torch "
            f"{torch.__version__} CPU fp32, {threads} threads of {cores} host cores ({_cpu_model()})")
    if not timed:
        log("cpu baseline did not finish one iteration")
        return {"value": None, "unit": "patches/s", "cores": threads, "kind": "port",
                "sample": f"{what}: no iteration finished within {budget_s}s"}
    secs = sum(timed) / len(timed)
    warm = "1 warm-up + " if any(d["it"] > 0 for d in its) else "cold, "
    res = {"value": round(cb / secs, 4), "unit": "patches/s", "cores": threads, "kind": "port",
           "sample": f"{warm}{len(timed)} timed {what}; {secs:.1f} s per iteration"}
    if gen is not None:       # SURVEY 8(d) asks for the generator alone beside the full iteration (one timed call each, same threads)
        res["generator_fwd_patches_per_s"] = round(cb / gen["gen_fwd_seconds"], 4)
        res["generator_fwd_bwd_patches_per_s"] = round(cb / gen["gen_fwd_bwd_seconds"], 4)
    return res


def self_launch(n, argv):
    """Start ``n`` ranks of this script on this node (one per GPU) under torch.distributed.run and relay their output; returns
    the launcher's exit code.  Used when ``--gpus N`` (N > 1, or --spawn) was given to a bare ``python bench.py``.  The rendezvous
    is torchrun's own (`--standalone`: a c10d store on a port the launcher binds itself and keeps), so there is no window between
    choosing a free port and using it."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL needs it on this driver
    env["RCOT_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", os.path.abspath(__file__)] + [a for a in argv if a != "--spawn"]
    log(f"--gpus {n} without WORLD_SIZE: starting {n} ranks: {' '.join(cmd[1:8])} ...")
    return subprocess.call(cmd, env=env)


def launch_check(args, world, rank):
    """--launch-check: the launcher / rendezvous plumbing without kernels (runs on CPU with gloo, so the N>1 spawn path has a
    test where there is no GPU): every rank joins the group, all-reduces a rank-stamped vector and a gradient-sized flat
    bucket, and rank 0 prints the JSON line with the rank count it saw."""
    backend = args.backend or ("nccl" if torch.cuda.is_available() else "gloo")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    dev = "cpu"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dev = "cuda"
    dist.init_process_group(backend, rank=rank, world_size=world)
    v = torch.zeros(world, device=dev)
    v[rank] = rank + 1.0
    dist.all_reduce(v)
    bucket = torch.ones(8 << 20, device=dev)
    dist.barrier()
    t0 = time.perf_counter()
    dist.all_reduce(bucket)
    if dev == "cuda":
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = bool((v.cpu() == torch.arange(1, world + 1, dtype=torch.float32)).all()) and float(bucket[0]) == world
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "launch-check", "n_gpus": world, "ranks_seen": int((v > 0).sum()), "backend": backend, "ok": ok,
                          "self_launched": os.environ.get("RCOT_BENCH_SELF_LAUNCHED") == "1",
                          "allreduce_32MiB_ms": round(dt * 1e3, 3)}), flush=True)
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE workload (see module docstring)")
    ap.add_argument("--prec", default=os.environ.get("RCOT_GEMM_PREC", "fp32"), choices=["fp32", "bf16x6", "bf16x3"],
                    help="arithmetic of the top-level value: fp32 (default, the product's default and the reference's arithmetic: exact fp32 "
                         "MFMA), bf16x6 (fp32-class results from the bf16 pipe) or bf16x3; the other two are timed in the same run with the "
                         "same number of steps and reported under extra.other_prec")
    ap.add_argument("--batch", type=int, default=0, help="override patches per GPU")
    ap.add_argument("--patch", type=int, default=0, help="override patch size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--spawn", action="store_true", help="start the ranks through the self-launcher even for --gpus 1")
    ap.add_argument("--launch-check", action="store_true", help="rendezvous + one all-reduce only (no kernels; works on CPU with gloo)")
    ap.add_argument("--backend", default=None, choices=["nccl", "gloo"], help="process-group backend of --launch-check")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.launch_check:
        sys.exit(launch_check(args, world, rank))
    torch.cuda.set_device(local)
    if world > 1 or os.environ.get("RCOT_FORCE_REDUCER") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from rcot_amd import lib
    from rcot_amd.net_restormer import F_net, T_net
    from rcot_amd.ops import default_backend
    from rcot_amd.profiling import GEMM_OPS, OpTimer
    from rcot_amd.synth import make_batch
    from rcot_amd.trainer import FlatOptimizer, MinimaxStep

    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["B"] = args.batch
    if args.patch:
        cfg["P"] = args.patch
    B, P = cfg["B"], cfg["P"]
    be = default_backend()
    PREC = {"fp32": lib.PREC_FP32, "bf16x3": lib.PREC_BF16X3, "bf16x6": lib.PREC_BF16X6}
    be.prec = PREC[args.prec]
    be.x6_packs = True                                   # all three arithmetics are timed in this process
    Tn, Fn = T_net(decoder=True, seed=1234), F_net(patch_size=P, seed=1235)     # same init on every rank
    lr = 1e-4
    st = MinimaxStep(Tn, Fn, FlatOptimizer(Tn, "RMSprop", lr / 2), FlatOptimizer(Fn, "RMSprop", lr), 1.0, 10000.0)
    de = [cfg["de"]] * B
    st.set_de_ids(de)
    de_dev = torch.tensor(de, dtype=torch.int32, device="cuda")
    nb = 4
    batches = []
    for i in range(nb):                                 # distinct shards per rank, resident in HBM
        _, x, y = make_batch((1000 + args.config) * 1000 + (i * world + rank), B, P, de, unpaired=cfg["unpaired"])
        batches.append((x.cuda(), y.cuda()))
    gen = torch.Generator().manual_seed(77 + rank)
    alphas = [torch.rand(B, generator=gen).cuda() for _ in range(nb)]

    def step(i, eager=False):
        x, y = batches[i % nb]
        (st.iteration if eager else st.run)(x, y, de_dev, alphas[i % nb], cfg["paired"])

    def timed_run(steps, first):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(first + i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        return dt

    log(f"rank {rank}/{world}: config {args.config} (B={B}, {P}x{P}), prec {args.prec}, warmup {args.warmup}")
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    log("warmup done")
    dt = timed_run(args.steps, args.warmup)
    losses = st.scalars()
    log(f"timed {args.steps} steps: {dt / args.steps * 1e3:.1f} ms/step")
    torch.cuda.synchronize()
    th = time.perf_counter()
    step(args.warmup + args.steps)
    host_ms = (time.perf_counter() - th) * 1e3          # time to ENQUEUE one step (GPU runs behind)
    torch.cuda.synchronize()
    plans = st.planned is not None and st.planned.enabled
    mode = "eager launches from a recorded launch plan" if plans else "eager launches, Python schedule"
    log(f"host enqueue time of one step: {host_ms:.1f} ms ({mode})")

    # ---- per-kernel timing pass (HIP events on the launch stream) -> roofline of the dominant KERNEL SYMBOL, in both arithmetics
    roof, extra = None, {"host_enqueue_ms_per_step": round(host_ms, 1), "launch_mode": mode}
    if plans:
        extra["plan_launches"] = [e["plan"].n_launches for e in st.planned.cache.values()]
    scale = (P / 128.0) ** 2
    ARITH = {"fp32": "fp32 MFMA 32x32x2 (exact: the reference's arithmetic)",
             "bf16x6": "bf16x6: weight projections as six bf16 partial products of a three-term split (fp32-class: as close to fp64 as the "
                       "fp32 MFMA kernel), everything else exact fp32 MFMA; peak quoted = fp32 MFMA 157.3 TFLOP/s",
             "bf16x3": "bf16x3 split MFMA 32x32x16 (3 products per fp32 product, fp32 accumulate); peak = 2500/3 TFLOP/s fp32-equivalent"}

    def measure_roofline(prec):
        """One eagerly launched iteration with HIP events around every entry point (in situ: the neighbours, the side stream and
        the caches are those of the real step), grouped by the kernel symbol the dispatcher chose (rcot_last_kernel): the symbol
        with the largest summed time is the roofline object, with its per-shape table, so that `frac` can be recomputed from
        profiles/r04_kernel_stats_*.txt (total time of the symbol) and the algorithmic bytes listed here.  Then the north_star
        unit (two-pass T_net forward + backward) timed two ways."""
        mfma_peak = MFMA_BF16_PEAK_TF / 3.0 if prec == "bf16x3" else MFMA_F32_PEAK_TF
        tm = OpTimer(Tn.be)
        step(args.warmup + args.steps, eager=True)       # per-op events need the eager launch path
        summ = tm.summary()
        syms = tm.by_symbol()
        tm.remove()
        if os.environ.get("RCOT_BENCH_SHAPES"):
            for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
                log(f"  op {k:18s} {v['ms']:8.2f} ms  x{v['calls']}")
            n_rows = int(os.environ["RCOT_BENCH_SHAPES"]) if os.environ["RCOT_BENCH_SHAPES"].isdigit() else 40
            for row in tm.by_shape(n_rows):
                log(f"  {row[2]:9.3f} ms  x{row[1]:<4d} {row[3]:>12s}  {row[0]}")
            for k, v in sorted(syms.items(), key=lambda kv: -kv[1]["ms"])[:25]:
                log(f"  sym {v['ms']:8.2f} ms x{v['calls']:<4d} {v['bytes'] / max(v['ms'], 1e-9) / 1e6:7.0f} GB/s  {k}")
        tot_ms = sum(v["ms"] for v in summ.values())
        launches = sum(v["calls"] for v in summ.values())
        g_ms = sum(v["ms"] for k, v in summ.items() if k in GEMM_OPS)
        g_fl = sum(v["flops"] for k, v in summ.items() if k in GEMM_OPS)
        g_calls = sum(v["calls"] for k, v in summ.items() if k in GEMM_OPS)

        def entry(sym, d):
            """the binding roof is the larger of (algorithmic bytes / HBM peak) and (flops / MFMA peak of the arithmetic THE SYMBOL
            computes in: the general engine, the exact-fp32 K-major kernel and the thin convolutions are fp32 MFMA / FMA in every
            mode) over ALL launches of the symbol in the step; achieved / frac are step-averaged (sum of work / sum of in-situ time)"""
            mfma_peak = MFMA_F32_PEAK_TF if (prec != "bf16x3" or sym.startswith(FP32_SYMBOLS)) else MFMA_BF16_PEAK_TF / 3.0
            t = d["ms"] * 1e-3
            gbs, tfs = d["bytes"] / t / 1e9, d["flops"] / t / 1e12
            hbm_bound = d["bytes"] / (HBM_PEAK_GBS * 1e9) >= d["flops"] / (mfma_peak * 1e12)
            rows = []
            for key, (n, ms, by, fl) in sorted(d["shapes"].items(), key=lambda kv: -kv[1][1]):
                rows.append({"launch": key, "launches": n, "us_per_launch": round(ms / n * 1e3, 2), "algorithmic_mbytes_per_launch": round(by / n / 1e6, 2),
                             "GBs": round(by / (ms * 1e-3) / 1e9, 1), "gflop_per_launch": round(fl / n / 1e9, 3)})
            return {"bound": "hbm" if hbm_bound else "mfma", "kernel": sym, "entry_points": sorted(d["entry_points"]),
                    "launches_per_step": d["calls"], "ms_per_step": round(d["ms"], 3), "share_of_gpu_time": round(d["ms"] / tot_ms, 4),
                    "achieved": round(gbs if hbm_bound else tfs, 2), "peak": HBM_PEAK_GBS if hbm_bound else round(mfma_peak, 1),
                    "unit": "GB/s" if hbm_bound else "TFLOP/s",
                    "frac": round((gbs / HBM_PEAK_GBS) if hbm_bound else (tfs / mfma_peak), 4), "traffic": None,
                    "algorithmic_gbytes_per_step": round(d["bytes"] / 1e9, 3), "algorithmic_tflop_per_step": round(d["flops"] / 1e12, 4),
                    "hbm_frac": round(gbs / HBM_PEAK_GBS, 4), "mfma_frac": round(tfs / mfma_peak, 4) if d["flops"] else None,
                    "mfma_peak_tflops": round(mfma_peak, 1),
                    "timing": "in situ: HIP events around every launch of the symbol inside one iteration (incl. the split-K reduce "
                              "launch behind it where there is one); step-averaged = sum of algorithmic work / sum of time",
                    "per_shape": rows[:12]}
        top = sorted(syms.items(), key=lambda kv: -kv[1]["ms"])
        # WHICH symbol dominates, and its `frac`, come from DEVICE time stamps: one replayed iteration runs with
        # rcot_profile_begin() on — every launch of the library then carries a start and a stop event of its own (hipExtLaunchKernelGGL:
        # the dispatch's begin / end, nothing inserted between the kernels) — and rcot_profile_end() returns launches and total
        # milliseconds per kernel symbol: the in-situ durations `rocprofv3 --kernel-trace --stats` lists, side stream included
        # (scripts/frac_check.py holds the line against that tool's output of the same call).  The in-situ BRACKETS of the pass above
        # (`frac_in_situ_brackets`, `ms_per_step`, `per_shape`) carry start event -> dispatch -> end event and, wherever the eager timing
        # pass is host-bound, the host's enqueue time: ~13 us per launch; `kernel_ms_back_to_back` re-issues the symbol's recorded launches
        # alone (rcot_amd.plan.time_symbol): what the symbol costs when nothing runs next to it.
        import ctypes as _C
        import re as _re

        def _norm(name):
            name = _re.sub(r"\(anonymous namespace\)::|rcot_nt::|rcot_x3w::|rcot_x3::|rcot::|^void ", "", name.strip())
            depth, out = 0, []
            for ch in name:                              # cut the argument list: the first '(' outside the template brackets
                if ch == "<":
                    depth += 1
                elif ch == ">":
                    depth -= 1
                elif ch == "(" and depth == 0:
                    break
                out.append(ch)
            return "".join(out).replace(" ", "")
        prof, dev_info = {}, None
        if world == 1 and hasattr(Tn.be.L, "rcot_profile_begin"):
            buf = _C.create_string_buffer(1 << 18)
            torch.cuda.synchronize()
            Tn.be.L.rcot_profile_begin()
            step(args.warmup + args.steps + 1)            # a replayed iteration (launch plan) when plans are on
            torch.cuda.synchronize()
            n_prof = Tn.be.L.rcot_profile_end(buf, 1 << 18)
            if os.environ.get("RCOT_BENCH_DUMP_PROFILE"):
                open(os.environ["RCOT_BENCH_DUMP_PROFILE"] + "." + prec, "w").write(buf.value.decode(errors="replace"))
            for ln_ in buf.value.decode(errors="replace").splitlines():
                parts = ln_.rsplit("|", 2)
                if len(parts) == 3 and not parts[0].startswith("#"):
                    k_ = _norm(parts[0])
                    c0, t0 = prof.get(k_, (0, 0.0))
                    prof[k_] = (c0 + int(parts[1]), t0 + float(parts[2]))
            dev_info = {"launches": n_prof, "kernel_ms": round(sum(v[1] for v in prof.values()), 2),
                        "note": "one replayed iteration, per-launch device time stamps; overlapping kernels of the two streams both count"}

        def prof_of(sym):
            """(launches, in-situ device ms) of an OpTimer symbol: entry points with one kernel behind them are named by the entry
            point and have no row (None)"""
            key = sym.replace(" ", "")
            hit = [v for k, v in prof.items() if k == key or (not key.endswith(">") and k.startswith(key))]
            return (sum(h[0] for h in hit), sum(h[1] for h in hit)) if hit else None
        kms = {}
        if plans and world == 1:
            from rcot_amd.plan import time_symbol
            ent = st.planned.cache.get(st.planned._key(batches[0][0], cfg["paired"]))
            if ent is not None:
                # (never re-issue the optimizer / weight-pack launches: they are not idempotent)
                cand = [k for k in ([k for k, _ in top[:12]] + [k for k, v in top[12:] if v["flops"] > 0])
                        if not any(w in k for w in ("rmsprop", "adam", "pack_weight"))]
                for k in cand:
                    t_k, n_k = time_symbol(ent["plan"], k)
                    if t_k is not None and n_k:
                        kms[k] = (t_k, n_k)
        dev = {k: prof_of(k) for k in syms}
        dev = {k: v for k, v in dev.items() if v is not None}
        rank_t = lambda k: dev[k][1] if k in dev else (kms[k][0] if k in kms else 0.0)
        dom = max(syms, key=rank_t) if (dev or kms) else top[0][0]
        r = entry(dom, syms[dom])
        r["arith"] = ARITH[prec]
        r["device_profile"] = dev_info
        r["frac_in_situ_brackets"] = r["frac"]
        r["kernel_ms_per_step"] = None
        r["kernel_ms_back_to_back"] = round(kms[dom][0], 3) if dom in kms else None
        t_dom = dev[dom][1] if dom in dev else (kms[dom][0] if dom in kms else None)
        if t_dom:
            work_t = (r["algorithmic_gbytes_per_step"] * 1e9 / (HBM_PEAK_GBS * 1e9)) if r["bound"] == "hbm" else \
                     (r["algorithmic_tflop_per_step"] / r["mfma_peak_tflops"])
            r["kernel_ms_per_step"] = round(t_dom, 3)
            r["kernel_launches_timed"] = dev[dom][0] if dom in dev else kms[dom][1]
            r["frac"] = round(work_t / (t_dom * 1e-3), 4)
            r["achieved"] = round(r["frac"] * r["peak"], 2)
            r["timing"] = ("frac / achieved / kernel_ms_per_step: per-launch DEVICE time stamps of the symbol's launches inside one replayed "
                           "iteration (rcot_profile_begin / _end: the durations rocprofv3 --kernel-trace lists); kernel_ms_back_to_back: the same "
                           "launches re-issued alone; frac_in_situ_brackets / ms_per_step / per_shape: HIP events around every launch of an "
                           "eagerly launched iteration" if dom in dev else "frac / kernel_ms_per_step: the symbol's recorded launches re-issued back to back")
        def brief(k, v):
            """a runner-up symbol: device-stamp time in situ, back-to-back time, and its roofline fraction from the former"""
            t_dev = dev[k][1] if k in dev else None
            t_bb = kms[k][0] if k in kms else None
            pk = MFMA_F32_PEAK_TF if (prec != "bf16x3" or k.startswith(FP32_SYMBOLS)) else MFMA_BF16_PEAK_TF / 3.0
            t = (t_dev or t_bb or v["ms"]) * 1e-3
            hb = v["bytes"] / (HBM_PEAK_GBS * 1e9) >= v["flops"] / (pk * 1e12)
            return {"kernel": k, "kernel_ms_per_step": round(t_dev, 3) if t_dev else None, "kernel_ms_back_to_back": round(t_bb, 3) if t_bb else None,
                    "launches": v["calls"], "bound": "hbm" if hb else "mfma",
                    "frac": round((v["bytes"] / t / 1e9 / HBM_PEAK_GBS) if hb else (v["flops"] / t / 1e12 / pk), 4)}
        r["next_symbols"] = [brief(k, v) for k, v in sorted(syms.items(), key=lambda kv: -rank_t(kv[0]))[:9] if k != dom][:8]
        r["dominance"] = ("by in-situ device time of one replayed iteration (what rocprofv3 --kernel-trace sums, side stream included); a symbol of "
                          "the side stream (the 1x1 weight gradients under fp32 / bf16x6) is stretched there by the data-gradient chain it runs next "
                          "to: kernel_ms_back_to_back is what it costs alone")
        r["gemm_family"] = {"kernels": "all MFMA GEMM launches of one step (1x1 / bmm / conv / linear entry points)",
                            "launches": g_calls, "achieved_tflops": round(g_fl / (g_ms * 1e-3) / 1e12, 2),
                            "mfma_frac": round(g_fl / (g_ms * 1e-3) / 1e12 / mfma_peak, 4), "ms_per_step": round(g_ms, 3),
                            "share_of_gpu_time": round(g_ms / tot_ms, 3)}
        # HBM traffic of the dominant symbol: PMC counters cannot be read inside the timed run (separate rocprofv3 --pmc passes);
        # the committed result of those passes is reported when it belongs to this symbol
        try:
            pmf = json.load(open(os.path.join(ROOT, "profiles", "pmc_dominant.json")))
            for pm in pmf.get("entries", [pmf]):
                if pm.get("kernel") == r["kernel"]:
                    r["traffic"] = pm["traffic_bytes"]
                    r["traffic_note"] = {k: pm[k] for k in ("read_bytes", "write_bytes", "algorithmic_bytes", "launch", "source") if k in pm}
        except (OSError, KeyError, ValueError):
            pass
        ex = {"per_op_ms": {k: round(v["ms"], 2) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:8]},
              "launches_per_step": launches,
              "hbm_bound_ops_GBs": {k: round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) for k, v in summ.items()
                                    if k not in GEMM_OPS and v["ms"] > 0.5}}
        # north_star roofline unit: two-pass Restormer forward+backward at this batch
        x, _ = batches[0]
        rr = torch.randn_like(x)
        hook, Tn.grad_ready_hook = Tn.grad_ready_hook, None          # the unit is the single-GPU kernel path: no collectives
        Tn.zero_grad()
        Tn.forward(x, save=True)
        Tn.backward(rr)
        torch.cuda.synchronize()
        # timed two ways: eagerly through the Python schedule (the GPU runs behind the host) and from a recorded launch plan (the same
        # eager launches, ~6 us of host each)
        t_eager = 1e9
        for _ in range(3):
            Tn.zero_grad()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            Tn.forward(x, save=True)
            Tn.backward(rr)
            torch.cuda.synchronize()
            t_eager = min(t_eager, time.perf_counter() - t1)
        from rcot_amd.plan import LaunchPlan

        def unit():
            Tn.zero_grad()
            Tn.forward(x, save=True)
            Tn.backward(rr)
        pl = LaunchPlan(be).record(unit)
        t_plan = 1e9
        for _ in range(4):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            pl.replay()
            torch.cuda.synchronize()
            t_plan = min(t_plan, time.perf_counter() - t1)
        t1 = time.perf_counter()
        pl.replay()
        plan_host_ms = (time.perf_counter() - t1) * 1e3
        torch.cuda.synchronize()
        n_unit = pl.n_launches
        del pl
        tfb = min(t_eager, t_plan)
        Tn.grad_ready_hook = hook
        r["path"] = {"unit": f"two-pass Restormer T_net forward+backward, B={B}, {P}x{P} (north_star roofline unit)", "gemm_prec": prec,
                     "ms": round(tfb * 1e3, 2), "ms_eager": round(t_eager * 1e3, 2),
                     "ms_plan_replay": round(t_plan * 1e3, 2), "plan_host_enqueue_ms": round(plan_host_ms, 2), "launches": n_unit,
                     "algorithmic_gbytes": round(TNET_FWDBWD_BYTES_PER_PATCH * scale * B / 1e9, 1),
                     "hbm_frac": round(TNET_FWDBWD_BYTES_PER_PATCH * scale * B / tfb / (HBM_PEAK_GBS * 1e9), 4),
                     "mfma_frac": round(3 * TNET_FWD_FLOP_PER_PATCH * scale * B / tfb / (mfma_peak * 1e12), 4),
                     "target_hbm_frac": 0.40}
        ex["tnet_fwd_bwd_ms"] = r["path"]["ms"]
        ex["tnet_fwd_bwd_hbm_frac"] = r["path"]["hbm_frac"]
        return r, ex

    if not args.no_roofline:
        roof, ex = measure_roofline(args.prec)
        extra.update(ex)
        log(f"roofline pass done ({args.prec}): {roof['kernel']} {roof['share_of_gpu_time']:.3f} of GPU time, frac {roof['frac']}; "
            f"T_net unit {roof['path']['ms']} ms")
        # the other arithmetics on the same workload: the SAME number of timed steps, their dominant symbol and their T_net unit, so
        # that all numbers travel with every bench line
        extra["other_prec"] = {}
        for other in [q for q in ("bf16x6", "fp32", "bf16x3") if q != args.prec]:
            be.prec = PREC[other]
            for i in range(max(1, args.warmup)):
                step(i)
            dto = timed_run(args.steps, args.warmup)
            oroof, oex = measure_roofline(other)
            extra["other_prec"][other] = {"prec": other, "dtype": DTYPE[other], "steps": args.steps, "ms_per_step": round(dto / args.steps * 1e3, 2),
                                          "patches_per_s": round(B * world * args.steps / dto, 2), "roofline": oroof,
                                          "tnet_fwd_bwd_ms": oex["tnet_fwd_bwd_ms"], "tnet_fwd_bwd_hbm_frac": oex["tnet_fwd_bwd_hbm_frac"]}
            log(f"other arithmetic ({other}): {dto / args.steps * 1e3:.1f} ms/step, T_net unit {oex['tnet_fwd_bwd_ms']} ms")
        be.prec = PREC[args.prec]

    # ---- data-parallel runs: what the collectives cost (one eagerly launched iteration, events on the reducer's side stream)
    comm = None
    if dist.is_initialized():
        comm = {"backend": dist.get_backend(), "ranks": dist.get_world_size(),
                "rccl": ".".join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == "nccl" else None,
                "allreduce_per_half_step": st.time_collectives(lambda: step(0, eager=True)),
                "note": "SUM all-reduce of the flat gradient buffers in 32 MiB buckets on a side stream, overlapped with backward"}

    # ---- CPU baseline: the oracle's iteration on the host cores (rank 0, N=1 only; bounded sample, own process)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg, lr)

    if rank == 0:
        total = B * world * args.steps
        dtype = DTYPE[args.prec]
        line = {"metric": "128x128 patches/sec (gen+critic step)" if P == 128 else f"{P}x{P} patches/sec (gen+critic step)",
                "value": round(total / dt, 3), "unit": "patches/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
                "config": {"workload": f"BASELINE configs[{cfg['idx']}]: Restormer T_net(decoder=True)+F_net({P}), {cfg['label']}, "
                                       f"B={B}/GPU {P}x{P}", "global_batch": B * world, "patch": P,
                           "parallelism": f"dp{world}", "gemm_prec": args.prec},
                "roofline": roof, "cpu_baseline": cpu, "comm": comm,
                "losses_last_step": {k: round(v, 6) for k, v in losses.items()}, "extra": extra}
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio: flush it first so that the JSON is the LAST stdout line
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
