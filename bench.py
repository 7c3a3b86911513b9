#!/usr/bin/env python
"""Headline benchmark: 128x128 patches/sec of the full RCOT minimax iteration (critic step +
gradient-penalty step + generator step, 3 optimizer steps) on the hand-written HIP path.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" is one minimax iteration over one synthetic batch already resident in HBM.
Workload = BASELINE.json configs[1]: Restormer T_net(decoder=True) + F_net(128), denoise_50
(de_id 2, Parseval branch of the Fourier OT cost), B=8 per GPU, 128x128, RMSprop, paired L1 term on
(README recipe).  Weak scaling: every rank processes its own batch of 8, gradients are SUM
all-reduced over RCCL.  Rank 0 prints ONE JSON line.
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

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3       # dense fp32 MFMA (v_mfma_f32_32x32x2_f32)
TNET_FWDBWD_BYTES_PER_PATCH = 15.946e9   # SURVEY.md 8(d) stage model, fp32, 128x128
TNET_FWD_FLOP_PER_PATCH = 166.5e9


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


CPU_SNIPPET = r"""
import sys, time, json, os, torch
sys.path.insert(0, %(root)r)
from oracle import rcot_oracle as O
from rcot_amd import params as PP
from rcot_amd.synth import make_batch
threads, P, cb, lr = %(threads)d, %(P)d, %(cb)d, %(lr)r
torch.set_num_threads(threads)
pT = {k: torch.from_numpy(v) for k, v in PP.seeded_params(PP.tnet_param_shapes(), 31, "T").items()}
pF = {k: torch.from_numpy(v) for k, v in PP.seeded_params(PP.fnet_param_shapes(P), 32, "F").items()}
_, xd, yd = make_batch(5, cb, P, [2] * cb)
t1 = time.perf_counter()
O.minimax_iteration(pT, pF, O.RMSprop(pT, lr / 2), O.RMSprop(pF, lr), xd, yd, [2] * cb,
                    torch.full((cb, 1, 1, 1), 0.5), 1.0, 10000.0, True)
print(json.dumps({"seconds": time.perf_counter() - t1}))
"""


def cpu_baseline(P, lr, budget_s=240):
    """One full minimax iteration of the oracle (critic + GP + generator steps, RMSprop) on the host cores, in its
    own process with a wall-clock bound.  threads = min(host cores, 32) (MKL-DNN stops scaling beyond that on
    this model size); the count actually used is reported."""
    import subprocess
    cores = os.cpu_count() or 1
    threads = min(cores, 32)
    cb = 2
    code = CPU_SNIPPET % dict(root=ROOT, threads=threads, P=P, cb=cb, lr=lr)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="")
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=budget_s, env=env)
        secs = json.loads(r.stdout.strip().splitlines()[-1])["seconds"]
    except Exception as e:  # timeout or failure: report it, never block the GPU result
        log(f"cpu baseline failed: {type(e).__name__}")
        return {"value": None, "unit": "patches/s", "cores": threads, "kind": "port",
                "sample": f"oracle iteration B={cb} {P}x{P} did not finish within {budget_s}s on {threads} threads ({cores} host cores)"}
    return {"value": round(cb / secs, 4), "unit": "patches/s", "cores": threads, "kind": "port",
            "sample": f"1 minimax iteration (critic+GP+generator, RMSprop) of the oracle, B={cb}, {P}x{P}, denoise_50, "
                      f"torch {torch.__version__} CPU fp32, {threads} threads of {cores} host cores, {secs:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="patches per GPU")
    ap.add_argument("--patch", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1 or os.environ.get("RCOT_FORCE_REDUCER") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from rcot_amd.net_restormer import F_net, T_net
    from rcot_amd.profiling import GEMM_OPS, OpTimer
    from rcot_amd.synth import make_batch
    from rcot_amd.trainer import FlatOptimizer, MinimaxStep

    B, P = args.batch, args.patch
    Tn, Fn = T_net(decoder=True, seed=1234), F_net(patch_size=P, seed=1235)     # same init on every rank
    lr = 1e-4
    st = MinimaxStep(Tn, Fn, FlatOptimizer(Tn, "RMSprop", lr / 2), FlatOptimizer(Fn, "RMSprop", lr), 1.0, 10000.0)
    de = [2] * B
    st.set_de_ids(de)
    de_dev = torch.tensor(de, dtype=torch.int32, device="cuda")
    nb = 4
    batches = []
    for i in range(nb):                                 # distinct shards per rank, resident in HBM
        _, x, y = make_batch(1002 * 1000 + (i * world + rank), B, P, de)
        batches.append((x.cuda(), y.cuda()))
    gen = torch.Generator().manual_seed(77 + rank)
    alphas = [torch.rand(B, generator=gen).cuda() for _ in range(nb)]

    def step(i):
        x, y = batches[i % nb]
        st.iteration(x, y, de_dev, alphas[i % nb], True)

    log(f"rank {rank}/{world}: nets built, warmup {args.warmup}")
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    log("warmup done")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    losses = st.scalars()
    log(f"timed {args.steps} steps: {dt / args.steps * 1e3:.1f} ms/step")
    torch.cuda.synchronize()
    th = time.perf_counter()
    step(args.warmup + args.steps)
    host_ms = (time.perf_counter() - th) * 1e3          # time to ENQUEUE one step (GPU runs behind)
    torch.cuda.synchronize()
    log(f"host enqueue time of one step: {host_ms:.1f} ms")

    # ---- per-kernel timing pass (HIP events on the launch stream) -> roofline of the dominant kernel
    roof, extra = None, {}
    if not args.no_roofline:
        tm = OpTimer(Tn.be)
        step(args.warmup + args.steps)
        summ = tm.summary()
        tm.remove()
        dom_key, dom_ms, dom_fl, dom_calls = tm.replay_dominant()
        log("per-op timing pass done")
        if os.environ.get("RCOT_BENCH_SHAPES"):
            for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
                log(f"  op {k:18s} {v['ms']:8.2f} ms  x{v['calls']}")
            for row in tm.by_shape(int(os.environ.get("RCOT_BENCH_SHAPES", "40")) if os.environ.get("RCOT_BENCH_SHAPES", "").isdigit() else 40):
                log(f"  {row[2]:9.3f} ms  x{row[1]:<4d} {row[3]:>12s}  {row[0]}")
        g_ms = sum(v["ms"] for k, v in summ.items() if k in GEMM_OPS)
        g_fl = sum(v["flops"] for k, v in summ.items() if k in GEMM_OPS)
        g_calls = sum(v["calls"] for k, v in summ.items() if k in GEMM_OPS)
        tot_ms = sum(v["ms"] for v in summ.values())
        # dominant kernel = the (MFMA GEMM launch, shape) with the largest share of the step; its duration is taken from a
        # back-to-back replay between two HIP events on the launch stream (no host gaps); the whole GEMM family
        # (every 1x1 / bmm / conv / linear launch of one step, event-bracketed individually) is reported next to it.
        roof = {"bound": "mfma", "kernel": "fp32 32x32x2 MFMA GEMM: " + dom_key, "launches_per_step": dom_calls,
                "achieved": round(dom_fl / (dom_ms * 1e-3) / 1e12, 3), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                "frac": round(dom_fl / (dom_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4), "traffic": None,
                "us_per_launch": round(dom_ms * 1e3, 2), "algorithmic_gflop_per_launch": round(dom_fl / 1e9, 3),
                "family": {"kernels": "all MFMA GEMM launches of one step (gemm_xx / gemm_nt / gemm_kernel)",
                           "launches": g_calls, "achieved": round(g_fl / (g_ms * 1e-3) / 1e12, 3),
                           "frac": round(g_fl / (g_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4), "ms_per_step": round(g_ms, 3),
                           "share_of_gpu_time": round(g_ms / tot_ms, 3),
                           "note": "per-launch events include launch gaps; rocprofv3 kernel time in profiles/ is the tighter figure"}}
        # HBM traffic of the dominant launch: PMC counters cannot be read inside the timed run (separate rocprofv3 --pmc
        # passes); the committed result of those passes is reported when it belongs to this very launch.
        try:
            pm = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_dominant.json")))
            if pm["launch"] == dom_key:
                roof["traffic"] = pm["traffic_bytes"]
                roof["traffic_note"] = {"read_bytes": pm["read_bytes"], "write_bytes": pm["write_bytes"],
                                        "algorithmic_bytes": pm["algorithmic_bytes"], "source": pm["source"]}
        except (OSError, KeyError, ValueError):
            pass
        top = sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:8]
        extra["per_op_ms"] = {k: round(v["ms"], 2) for k, v in top}
        extra["hbm_bound_ops_GBs"] = {k: round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) for k, v in summ.items()
                                      if k not in GEMM_OPS and v["ms"] > 0.5}
        # north_star roofline unit: two-pass Restormer forward+backward at this batch
        x, _ = batches[0]
        r = torch.randn_like(x)
        for _ in range(2):
            Tn.zero_grad()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            Tn.forward(x, save=True)
            Tn.backward(r)
            torch.cuda.synchronize()
            tfb = time.perf_counter() - t1
        extra["tnet_fwd_bwd_ms"] = round(tfb * 1e3, 2)
        extra["tnet_fwd_bwd_hbm_frac"] = round(TNET_FWDBWD_BYTES_PER_PATCH * B / tfb / (HBM_PEAK_GBS * 1e9), 4)
        extra["tnet_fwd_bwd_mfma_frac"] = round(3 * TNET_FWD_FLOP_PER_PATCH * B / tfb / (MFMA_F32_PEAK_TF * 1e12), 4)

    # ---- CPU baseline: the oracle's iteration on the host cores (rank 0, N=1 only; bounded sample, own process)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(P, lr)

    if rank == 0:
        total = B * world * args.steps
        line = {"metric": "128x128 patches/sec (gen+critic step)", "value": round(total / dt, 3), "unit": "patches/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp32", "data": "synthetic",
                "config": {"workload": f"BASELINE configs[1]: Restormer T_net(decoder=True)+F_net({P}), denoise_50, "
                                       f"B={B}/GPU {P}x{P}, RMSprop, paired", "global_batch": B * world, "patch": P,
                           "parallelism": f"dp{world}"},
                "roofline": roof, "cpu_baseline": cpu,
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
