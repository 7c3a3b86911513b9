"""CPU tier, N>1 path: two gloo ranks, each with half of a global batch, must reproduce the single-process
global-batch minimax iteration (mean terms averaged, the Fourier penalty SUMMED, the RMSE over the GLOBAL
batch — SURVEY.md section 8e), using the bucketed overlapped reducer.  Kernel layer = torch test double."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, seeded_tensor

PS, BG, LR = 32, 4, 1e-4
DE = [2, 3, 0, 7]


def _setup(be):
    from rcot_amd import params as P
    from rcot_amd.net_restormer import F_net, T_net
    from rcot_amd.trainer import FlatOptimizer, MinimaxStep
    D = torch.float64
    Tn, Fn = T_net(decoder=True, backend=be, seed=0), F_net(patch_size=PS, backend=be, seed=1)
    Tn.load_state_dict({k: torch.from_numpy(v).to(D) for k, v in P.seeded_params(P.tnet_param_shapes(), 31, "T").items()})
    Fn.load_state_dict({k: torch.from_numpy(v).to(D) for k, v in P.seeded_params(P.fnet_param_shapes(PS), 32, "F").items()})
    st = MinimaxStep(Tn, Fn, FlatOptimizer(Tn, "RMSprop", LR / 2), FlatOptimizer(Fn, "RMSprop", LR), 1.0, 10000.0,
                     bucket_elems=1 << 20)
    return Tn, Fn, st


def _data():
    D = torch.float64
    clean = seeded_tensor(801, (BG, 3, PS, PS), lo=0.0, hi=1.0, dtype=D)
    deg = (clean + seeded_tensor(802, (BG, 3, PS, PS), scale=50 / 255, dtype=D)).clamp(0, 1)
    alpha = seeded_tensor(803, (BG,), lo=0.0, hi=1.0, dtype=D)
    return deg, clean, alpha


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from host_double import TorchDouble
    Tn, Fn, st = _setup(TorchDouble(torch.float64))
    assert st.world == world and st.redT.enabled
    deg, clean, alpha = _data()
    per = BG // world
    sl = slice(rank * per, (rank + 1) * per)
    de = DE[sl]
    st.set_de_ids(DE)
    st.iteration(deg[sl].contiguous(), clean[sl].contiguous(), torch.tensor(de, dtype=torch.int32), alpha[sl].contiguous(), True)
    if rank == 0:
        torch.save({"T": Tn.store.flat.clone(), "F": Fn.store.flat.clone(), "nb": len(st.redT.bounds)}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_equal_single_process_global_batch(tmp_path):
    from host_double import TorchDouble
    out = str(tmp_path / "rank0.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    assert got["nb"] > 3                      # several buckets were exercised
    torch.set_num_threads(4)
    Tn, Fn, st = _setup(TorchDouble(torch.float64))
    deg, clean, alpha = _data()
    st.set_de_ids(DE)
    st.iteration(deg, clean, torch.tensor(DE, dtype=torch.int32), alpha, True)
    for net, key in ((Tn, "T"), (Fn, "F")):
        ref, g = net.store.flat, got[key]
        assert float((g - ref).abs().max()) <= 1e-9 * float(ref.abs().max()), key
