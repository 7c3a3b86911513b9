"""GPU tier: host-side launch plans (rcot_amd/plan.py) against walking the schedule every iteration: same parameters after
four iterations on changing batches (the paired flag changes, i.e. a second plan), same logged losses, exactly one optimizer
step per call (the warm-up pass before the first recording must not count), recorded addresses survive allocator churn
(empty_cache + foreign allocations between replays), and the reducer's collectives stay in the list as host actions."""
import os

import pytest
import torch

from rcot_amd import params as P

pytestmark = pytest.mark.gpu


def _run(plan: bool, steps=4, ps=64, B=2, churn=False, precs=None, lr_drop_at=None, paired_of=lambda i: i % 2 == 0):
    from rcot_amd.net_restormer import F_net, T_net
    from rcot_amd.synth import make_batch
    from rcot_amd.trainer import FlatOptimizer, MinimaxStep
    from rcot_amd import lib
    os.environ["RCOT_PLAN"] = "1" if plan else "0"
    try:
        lr, de = 1e-4, [2, 3]
        Tn, Fn = T_net(decoder=True), F_net(patch_size=ps)
        Tn.load_state_dict({k: torch.from_numpy(v) for k, v in P.seeded_params(P.tnet_param_shapes(), 31, "T").items()})
        Fn.load_state_dict({k: torch.from_numpy(v) for k, v in P.seeded_params(P.fnet_param_shapes(ps), 32, "F").items()})
        st = MinimaxStep(Tn, Fn, FlatOptimizer(Tn, "RMSprop", lr / 2), FlatOptimizer(Fn, "RMSprop", lr), 1.0, 10000.0)
    finally:
        os.environ.pop("RCOT_PLAN", None)
    assert (st.planned is not None) == plan
    st.set_de_ids(de)
    de_dev = torch.tensor(de, dtype=torch.int32).cuda()
    logs, junk = [], []
    prec0 = st.be.prec
    for i in range(steps):
        if precs is not None:
            st.be.prec = {"fp32": lib.PREC_FP32, "bf16x3": lib.PREC_BF16X3, "bf16x6": lib.PREC_BF16X6}[precs[i]]
        if lr_drop_at is not None and i == lr_drop_at:                     # the schedule's decay (trainer.py:228-243)
            for o in (st.To, st.Fo):
                for g in o.param_groups:
                    g["lr"] *= 0.5
        _, x, y = make_batch(300 + i, B, ps, de)
        alpha = torch.rand(B, generator=torch.Generator().manual_seed(i))
        st.run(x.cuda(), y.cuda(), de_dev, alpha.cuda(), paired_of(i))      # default: the paired flag alternates: two plans, each replayed once
        torch.cuda.synchronize()
        logs.append(st.scalars())
        if churn:                                                          # the plan's addresses must not depend on the general pool
            junk = [torch.full((1 << 22,), float("nan"), device="cuda") for _ in range(8)]
            del junk
            torch.cuda.empty_cache()
    n = [e["plan"].n_launches for e in st.planned.cache.values()] if plan else []
    st.be.prec = prec0
    return Tn.store.flat.clone(), Fn.store.flat.clone(), logs, n


def test_plan_replay_equals_eager():
    Te, Fe, le, _ = _run(False)
    Tp, Fp, lp, n = _run(True, churn=True)
    assert len(n) == 2 and all(v > 1000 for v in n), n                      # two configurations, each a few thousand launches
    # float atomics (depthwise weight gradients) make two runs differ in the last bits, RMSprop's sign-like first steps amplify
    # that for near-zero gradients: compare the parameter UPDATE in L2 (as tests/test_graph_gpu.py does for HIP graphs)
    T0, F0, _, _ = _run(False, steps=0)
    assert float((Tp - Te).norm() / (Te - T0).norm()) < 5e-2
    assert float((Fp - Fe).norm() / (Fe - F0).norm()) < 5e-2
    assert float((Te - T0).norm()) > 0 and float((Tp - T0).norm()) > 0
    for a, b in zip(le, lp):
        for k in a:
            assert abs(a[k] - b[k]) <= max(2e-4 * max(1e-3, abs(a[k])), 5e-6), (k, a[k], b[k])      # (5e-6: the floor of a loss that is a difference of means)
    assert bool(torch.isfinite(Tp).all()) and bool(torch.isfinite(Fp).all())


def test_plan_survives_arithmetic_switch_and_takes_a_new_learning_rate_without_recording_again():
    """fp32 -> bf16x3 -> fp32 on ONE network: the cached fp32 plan's rcot_pack_weights call points at the fp32
    descriptor table, which must outlive the switch, and the packs an iteration reads must have been refreshed after the other
    arithmetic's optimizer steps; then a learning-rate decay: the rates are by-value arguments of the three optimizer
    launches and are patched into the recorded calls (PlannedMinimax._set_lr) — the fp32 plan of step 0 is REPLAYED at the new rate,
    nothing is recorded again.  Both against the eager schedule doing the same sequence (an unpatched rate would leave the last
    step twice as long: far outside the bars)."""
    precs = ["fp32", "bf16x3", "fp32", "fp32", "fp32"]     # step 2 and 3 REPLAY the fp32 plan recorded at step 0; step 4 decays the rates
    kw = dict(steps=5, precs=precs, lr_drop_at=4, paired_of=lambda i: True)
    Te, Fe, le, _ = _run(False, **kw)
    Te2, Fe2, le2, _ = _run(False, **kw)           # run-to-run spread of the eager schedule itself (float atomics x RMSprop's sign-like steps)
    Tp, Fp, lp, n = _run(True, churn=True, **kw)
    assert len(n) == 2, n                          # one plan per arithmetic; the decay at step 4 recorded nothing
    T0, F0, _, _ = _run(False, steps=0)
    rT, rF = float((Tp - Te).norm() / (Te - T0).norm()), float((Fp - Fe).norm() / (Fe - F0).norm())
    nT, nF = float((Te2 - Te).norm() / (Te - T0).norm()), float((Fe2 - Fe).norm() / (Fe - F0).norm())
    assert rT < max(5e-2, 3 * nT) and rF < max(5e-2, 3 * nF), (rT, rF, nT, nF)
    # (the critic loss is a difference of two means near zero: its bar is the eager schedule's own run-to-run spread, measured here,
    # with an absolute floor — round 5: 1.3e-6 apart on -8.2e-4 in one run of six, against a 5e-7 bar)
    for a, a2, b in zip(le, le2, lp):
        for k in a:
            assert abs(a[k] - b[k]) <= max(5e-4 * max(1e-3, abs(a[k])), 4 * abs(a[k] - a2[k]), 5e-6), (k, a[k], a2[k], b[k])
    assert bool(torch.isfinite(Tp).all()) and bool(torch.isfinite(Fp).all())


def test_recording_a_plan_is_safe_against_the_cyclic_collector():
    """An earlier MinimaxStep with its plans (each owns a torch.cuda.MemPool) is cyclic garbage after _run() returns.  A pool that
    the collector finalises INSIDE another plan's recording aborts the process (its destructor empties its cache, which asserts
    that no thread is allocating into a pool) — round 5: `pytest -q` of this file died in the middle of a recording, `-v` did not.
    LaunchPlan.record collects before it opens its pool and keeps the collector off while recording; here the collector is set
    to fire at every allocation, so a regression aborts this test instead of one run in three."""
    import gc
    _run(True, steps=1)
    old = gc.get_threshold()
    gc.set_threshold(1, 1, 1)
    try:
        Tp, Fp, _, n = _run(True, steps=2)
    finally:
        gc.set_threshold(*old)
    assert len(n) == 2 and bool(torch.isfinite(Tp).all()) and bool(torch.isfinite(Fp).all())


def test_plan_with_forced_reducer_keeps_collectives(tmp_path):
    """RCOT_FORCE_REDUCER=1 at world size 1 (RCCL): the bucketed all-reduces are host actions inside the plan"""
    import subprocess
    import sys
    from conftest import ROOT
    code = (
        "import os, torch, torch.distributed as dist\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RCOT_FORCE_REDUCER='1', RCOT_PLAN='1')\n"
        "torch.cuda.set_device(0); dist.init_process_group('nccl', rank=0, world_size=1)\n"
        "from rcot_amd.net_restormer import F_net, T_net\n"
        "from rcot_amd.synth import make_batch\n"
        "from rcot_amd.trainer import FlatOptimizer, MinimaxStep\n"
        "Tn, Fn = T_net(decoder=True, seed=1), F_net(patch_size=32, seed=2)\n"
        "st = MinimaxStep(Tn, Fn, FlatOptimizer(Tn, 'RMSprop', 5e-5), FlatOptimizer(Fn, 'RMSprop', 1e-4), 1.0, 10000.0, bucket_elems=1 << 22)\n"
        "de = [2, 3]; st.set_de_ids(de); d = torch.tensor(de, dtype=torch.int32).cuda()\n"
        "for i in range(3):\n"
        "    _, x, y = make_batch(5 + i, 2, 32, de)\n"
        "    st.run(x.cuda(), y.cuda(), d, torch.full((2,), 0.5).cuda(), True)\n"
        "torch.cuda.synchronize()\n"
        "p = list(st.planned.cache.values())[0]['plan']\n"
        "acts = sum(1 for c in p.cmds if c[1] is None)\n"
        "s = st.scalars()\n"
        "assert acts >= 6 and all(v == v for v in s.values()), (acts, s)\n"
        "print('host actions', acts, 'launches', p.n_launches)\n"
        "dist.destroy_process_group()\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
