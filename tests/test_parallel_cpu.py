"""CPU tier, N>1 path: two gloo ranks, each with half of a global batch, must reproduce the single-process
global-batch minimax iteration (mean terms averaged, the Fourier penalty SUMMED, the RMSE over the GLOBAL
batch, alpha drawn per GLOBAL sample index — SURVEY.md section 8e), using the bucketed overlapped reducer, and
going through ``trainer.train()`` itself (which draws alpha).  The logged losses must be the global-batch values on
every rank, a seed drawn on rank 0 must reach every rank, and parameter buffers must be broadcast.
Kernel layer = torch test double."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, seeded_tensor

PS, BG, LR = 32, 4, 1e-4
DE = [2, 3, 0, 7]


def _setup(be, backbone="restormer"):
    from rcot_amd import params as P
    from rcot_amd.net_restormer import F_net, T_net
    from rcot_amd.trainer import FlatOptimizer, MinimaxStep
    D = torch.float64
    Fn = F_net(patch_size=PS, backend=be, seed=1)
    if backbone == "mprnet":                                   # the older transport map behind the same step and reducer
        from rcot_amd.mprnet_hip import MPRNetHip
        Tn = MPRNetHip(backend=be, seed=0)
        Tn.load_state_dict({k: v.to(D) for k, v in Tn.state_dict().items()})
    else:
        Tn = T_net(decoder=True, backend=be, seed=0)
        Tn.load_state_dict({k: torch.from_numpy(v).to(D) for k, v in P.seeded_params(P.tnet_param_shapes(), 31, "T").items()})
    Fn.load_state_dict({k: torch.from_numpy(v).to(D) for k, v in P.seeded_params(P.fnet_param_shapes(PS), 32, "F").items()})
    st = MinimaxStep(Tn, Fn, FlatOptimizer(Tn, "RMSprop", LR / 2), FlatOptimizer(Fn, "RMSprop", LR), 1.0, 10000.0,
                     bucket_elems=1 << 20)
    return Tn, Fn, st


def _data():
    D = torch.float64
    clean = seeded_tensor(801, (BG, 3, PS, PS), lo=0.0, hi=1.0, dtype=D)
    deg = (clean + seeded_tensor(802, (BG, 3, PS, PS), scale=50 / 255, dtype=D)).clamp(0, 1)
    return deg, clean


def _train_one(st, Tn, Fn, sl):
    """One iteration through trainer.train() on the slice ``sl`` of the global batch; returns the logged scalars."""
    from argparse import Namespace
    from rcot_amd import trainer as TR
    TR.opt = Namespace(lr=LR, step=20, pairnum=10 ** 7, batchSize=BG, seed=5, sigma=1.0, Sigma=10000.0, type="t")
    deg, clean = _data()
    batch = ([["n"] * len(DE[sl]), torch.tensor(DE[sl])], deg[sl].contiguous(), clean[sl].contiguous())
    TR.train([batch], st.To, st.Fo, Tn, Fn, 1, st)
    return st.scalars()


def _worker(rank, world, port, out_path, backbone="restormer"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from host_double import TorchDouble
    from rcot_amd import parallel as par
    assert par.broadcast_int(10 + rank) == 10                 # the job's seed is rank 0's draw (trainer.main)
    Tn, Fn, st = _setup(TorchDouble(torch.float64), backbone)
    assert st.world == world and st.redT.enabled
    probe = torch.full((8,), float(rank))
    par.broadcast_flat(probe, 0)
    assert float(probe.abs().max()) == 0.0                    # replicas start from rank 0's parameters
    per = BG // world
    logs = _train_one(st, Tn, Fn, slice(rank * per, (rank + 1) * per))
    torch.save({"T": Tn.store.flat.clone(), "F": Fn.store.flat.clone(), "nb": len(st.redT.bounds), "logs": logs},
               out_path + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("backbone", ["restormer", "mprnet"])
def test_two_ranks_equal_single_process_global_batch(tmp_path, restore_thread_count, backbone):
    from host_double import TorchDouble
    out = str(tmp_path / "rank0.pt")
    port = 29500 + (os.getpid() % 2000) + (7 if backbone == "mprnet" else 0)
    mp.spawn(_worker, args=(2, port, out, backbone), nprocs=2, join=True)
    got, got1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert got["nb"] > 3                      # several buckets were exercised
    torch.set_num_threads(4)
    Tn, Fn, st = _setup(TorchDouble(torch.float64), backbone)
    logs = _train_one(st, Tn, Fn, slice(0, BG))
    for net, key in ((Tn, "T"), (Fn, "F")):
        ref, g = net.store.flat, got[key]
        assert float((g - ref).abs().max()) <= 1e-9 * float(ref.abs().max()), key
        assert torch.equal(got[key], got1[key]), key          # replicas stay identical
    for k, v in logs.items():                                 # every rank logs the GLOBAL-batch losses
        for g in (got["logs"], got1["logs"]):
            assert abs(g[k] - v) <= 1e-9 * max(1.0, abs(v)), (k, g[k], v)


@pytest.mark.timeout(600)
def test_bench_self_launcher_starts_two_ranks():
    """`python bench.py --gpus 2` (the driver's form, no torchrun around it) must start its own ranks: the launcher, the
    rendezvous on 127.0.0.1 and one gradient-sized all-reduce, on gloo (no kernels; the GPU tier runs the real step through the
    same launcher at --gpus 1)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=500, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["ok"] and line["self_launched"]
