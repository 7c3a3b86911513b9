"""RCOT minimax trainer on the MI355X kernels — host side of the reference's ``trainer.py``.

Keeps the reference's CLI flags (trainer.py:22-58), learning-rate schedule (:228-243), iteration
structure (critic step -> separate gradient-penalty step -> generator step, :262-346), print format
(:348-354) and checkpoint naming/keys (:362-371).  The compute is done by ``rcot_amd.net_restormer``
(HIP kernels); this file only sequences it.  Additions are supersets: ``--seed``, ``--synthetic``,
``--iters``, data-parallel launch through torchrun (one process per GPU, RCCL).
"""
from __future__ import annotations

import argparse
import math
import os
import time
from typing import Optional, Sequence

import torch

from . import parallel as par
from .net_restormer import F_net, T_net

# ------------------------------------------------------------------------------- CLI (reference flags)
parser = argparse.ArgumentParser(description="RCOT minimax training on MI355X (reference-compatible flags)")
parser.add_argument("--batchSize", type=int, default=4, help="training batch size (global)")
parser.add_argument("--nEpochs", type=int, default=200)
parser.add_argument("--lr", type=float, default=1e-4)
parser.add_argument("--step", type=int, default=20, help="LR decays 10x every n epochs")
parser.add_argument("--cuda", default=True)
parser.add_argument("--resume", default=None, type=str)
parser.add_argument("--start-epoch", default=1, type=int)
parser.add_argument("--threads", type=int, default=0)
parser.add_argument("--pretrained", default="", type=str)
parser.add_argument("--gpus", default="0", type=str)
parser.add_argument("--pairnum", default=0, type=int, help="num of paired samples")
parser.add_argument("--de_type", nargs="+", default=["denoise_15", "denoise_25", "denoise_50", "derain", "dehaze"])
parser.add_argument("--denoise_dir", type=str, default="data/Train/Denoise/")
parser.add_argument("--derain_dir", type=str, default="data/Train/Derain/")
parser.add_argument("--dehaze_dir", type=str, default="data/Train/Dehaze/")
parser.add_argument("--degset", default="./data/test/derain/Rain100L/input/", type=str)
parser.add_argument("--tarset", default="./data/test/derain/Rain100L/target/", type=str)
parser.add_argument("--Sigma", default=10000, type=float)
parser.add_argument("--sigma", default=1, type=float)
parser.add_argument("--optimizer", default="RMSprop", type=str)
parser.add_argument("--type", default="Deraining", type=str)
parser.add_argument("--patch_size", type=int, default=64)
parser.add_argument("--num_workers", type=int, default=4)
parser.add_argument("--data_file_dir", type=str, default="data_dir/")
# supersets
parser.add_argument("--deblur_dir", type=str, default=None, help="(the reference reads args.deblur_dir but defines no flag)")
parser.add_argument("--lowlight_dir", type=str, default=None, help="(same for lowlight)")
parser.add_argument("--single_dir", type=str, default=None, help="(same for --de_type single)")
parser.add_argument("--seed", type=int, default=None, help="seed (the reference draws an unseeded random one)")
parser.add_argument("--prec", choices=["fp32", "bf16x6", "bf16x3", "bf16x1"], default=os.environ.get("RCOT_GEMM_PREC", "fp32"),
                    help="arithmetic of the 1x1 MFMA products (include/rcot_hip.h RCOT_PREC_*; one default for HipBackend(), this CLI and "
                         "bench.py): fp32 = exact fp32 MFMA, the reference's arithmetic (default); bf16x6 = fp32-class results from the bf16 "
                         "pipe (three-term split, six products: as accurate as exact fp32, ~2 %% faster); bf16x3 = two-term split (~2^-16 per "
                         "product, ~15 %% faster; within the north_star tolerances)")
parser.add_argument("--backbone", choices=["restormer", "mprnet"], default="restormer",
                    help="restormer: Net_Restormer.T_net on the HIP kernels (the hot path).  mprnet: the reference's older Net.T_net "
                         "(BASELINE configs[0]) — on the HIP kernels when a GPU is visible (rcot_amd/mprnet_hip.py), else on stock "
                         "PyTorch ops with torch autograd on the CPU (rcot_amd/mprnet.py; RCOT_MPRNET_STOCK=1 forces that form)")
parser.add_argument("--synthetic", action="store_true", help="seeded synthetic patches (no dataset folders needed)")
parser.add_argument("--iters", type=int, default=20, help="iterations per epoch with --synthetic")

opt: Optional[argparse.Namespace] = None

DE_IDS = {"denoise_15": 0, "denoise_25": 1, "denoise_50": 2, "derain": 3, "dehaze": 4, "deblur": 5,
          "lowlight": 6, "single": 7}   # util/dataset_utils.py:40


def freeze(model):      # utils.py:23-26
    """numerically inert on the explicit schedules (no autograd graph, no BN/dropout); on the autograd front end
    (rcot_amd.autograd.TNetModule / FNetModule) it is the reference's requires_grad_(False) + eval()"""
    if isinstance(model, torch.nn.Module):
        for p in model.parameters():
            p.requires_grad_(False)
        model.eval()
    return model


def unfreeze(model):    # utils.py:28-31
    if isinstance(model, torch.nn.Module):
        for p in model.parameters():
            p.requires_grad_(True)
        model.train(True)
    return model


# ------------------------------------------------------------------------------- optimizer
class FlatOptimizer:
    """RMSprop / Adam with torch defaults over a network's flat buffers (one fused launch)."""

    def __init__(self, net, kind: str, lr: float):
        self.net, self.kind, self.lr = net, kind, lr
        st = net.store
        be = net.be
        self.param_groups = [{"lr": lr}]          # the reference pokes param_groups[...]["lr"] (:240-243)
        n = st.layout.n_total
        if kind == "RMSprop":
            self.sq = be.zeros(n)
        elif kind == "Adam":
            self.m, self.v = be.zeros(n), be.zeros(n)
            self.t_main, self.t_tail = 0, 0
        else:
            raise ValueError(kind)

    def step(self, n_live: Optional[int] = None):
        """Update params[0:n_live) (default: every live parameter).  A shorter range reproduces
        autograd's 'grad is None -> skipped' for the trailing tensors (fc2.bias in the GP step)."""
        st, be = self.net.store, self.net.be
        lr = self.param_groups[0]["lr"]
        full = st.layout.n_live
        n = full if n_live is None else n_live
        n4 = (n + 3) // 4 * 4 if n < full else full
        n4 = min(n4, st.layout.n_total)
        if self.kind == "RMSprop":
            be.rmsprop_step(st.flat, st.grad, self.sq, n4, lr)
        else:
            self.t_main += 1
            be.adam_step(st.flat, st.grad, self.m, self.v, min(n4, self._tail_start()), lr, self.t_main)
            if n >= full and self._tail_start() < full:
                self.t_tail += 1
                o = self._tail_start()
                be.adam_step(st.flat[o:], st.grad[o:], self.m[o:], self.v[o:], st.layout.n_total - o, lr, self.t_tail)
        if hasattr(self.net, "repack"):
            self.net.repack()            # private K-major weight copies follow the parameters

    def _tail_start(self):
        """Offset of tensors that skip some steps (their Adam step count differs)."""
        lay = self.net.store.layout
        return lay.offset.get("fc2.bias", lay.n_live) if isinstance(self.net, F_net) else lay.n_live

    # ---- checkpoint / replica plumbing (supersets: the reference saves no optimizer state, trainer.py:362-371)
    def _state_tensors(self):
        return {k: getattr(self, k) for k in ("sq", "m", "v") if hasattr(self, k)}

    def state_dict(self):
        """Optimizer state keyed by the reference's state_dict names (layout-independent), plus the step counters."""
        st = self.net.store
        out = {"kind": self.kind, "lr": self.param_groups[0]["lr"], "t_main": getattr(self, "t_main", 0),
               "t_tail": getattr(self, "t_tail", 0), "state": {}}
        for key, flat in self._state_tensors().items():
            d = {}
            for name, shp in st.shapes:
                o = st.layout.offset[name]
                n = 1
                for v in shp:
                    n *= v
                d[name] = flat[o:o + n].view(*shp).detach().cpu().clone()
            out["state"][key] = d
        return out

    def load_state_dict(self, sd):
        if sd.get("kind") != self.kind:
            raise ValueError(f"optimizer state is for {sd.get('kind')}, this run uses {self.kind}")
        st = self.net.store
        for key, flat in self._state_tensors().items():
            for name, shp in st.shapes:
                o = st.layout.offset[name]
                t = sd["state"][key][name]
                flat[o:o + t.numel()].copy_(t.reshape(-1).to(flat.dtype))
        if self.kind == "Adam":
            self.t_main, self.t_tail = int(sd.get("t_main", 0)), int(sd.get("t_tail", 0))

    def broadcast_state(self):
        for flat in self._state_tensors().values():
            par.broadcast_flat(flat, 0)

    def zero_state(self):
        for t in (getattr(self, "sq", None), getattr(self, "m", None), getattr(self, "v", None)):
            if t is not None:
                t.zero_()


def make_optimizers(Tnet, Fnet, name: str, lr: float):
    """trainer.py:121-126: lr/2 for the transport map, lr for the potential."""
    return FlatOptimizer(Tnet, name, lr / 2), FlatOptimizer(Fnet, name, lr)


# ------------------------------------------------------------------------------- one minimax iteration
class MinimaxStep:
    """The body of the reference's training loop (trainer.py:247-346) for one (local) batch."""

    def __init__(self, Tnet: T_net, Fnet: F_net, T_opt: FlatOptimizer, F_opt: FlatOptimizer, sigma: float,
                 Sigma: float, bucket_elems: int = 8 << 20):
        self.T, self.F, self.To, self.Fo = Tnet, Fnet, T_opt, F_opt
        self.sigma, self.Sigma = float(sigma), float(Sigma)
        self.be = Tnet.be
        self.world = par.world_size()
        self.redT = par.GradReducer(Tnet.store.grad, Tnet.store.layout.n_live, bucket_elems)
        self.redF = par.GradReducer(Fnet.store.grad, Fnet.store.layout.n_live, bucket_elems)
        # (one process: no hooks at all — a hook makes the sweep join the weight-gradient side stream before it fires)
        Tnet.grad_ready_hook = self.redT.ready if self.redT.enabled else None
        Fnet.grad_ready_hook = self.redF.ready if self.redF.enabled else None          # critic-loss backward: buckets leave while the sweep continues
        Fnet.grad_tail_hook = self.redF.ready_tail if self.redF.enabled else None      # gradient penalty: the same layout filled from its end
        self.logs = {}
        #: optional callback(tag) invoked right before each of the three optimizer steps ("F_critic", "F_gp", "T_gen"), when the
        #: gradient buffers of that half-step are final (after the reducers): gradient-level parity tests read them there
        self.grad_probe = None
        # host-side launch plans (rcot_amd/plan.py), the default: the launch sequence is recorded once per configuration and
        # re-issued from a flat command list — the same eager launches without walking the Python schedule (~50 -> ~15 ms of
        # host time per iteration).  RCOT_PLAN=0 walks the schedule every iteration.
        self.planned = None
        from .plan import PlannedMinimax, plan_default
        if plan_default() and Tnet.store.flat.is_cuda:
            self.planned = PlannedMinimax(self)

    def run(self, degraded, target, de_id, alpha, paired: bool):
        """One minimax iteration: replay of the recorded launch plan when available, else the eager launch sequence."""
        if self.planned is not None and self.grad_probe is None and self.comm_log is None:
            return self.planned.iteration(degraded, target, de_id, alpha, paired)
        return self.iteration(degraded, target, de_id, alpha, paired)

    def iteration(self, degraded, target, de_id, alpha, paired: bool):
        """degraded/target: [B,3,P,P] local shard; de_id: int32 [B] (device); alpha: [B] in [0,1)
        (the reference samples it on the CPU RNG, trainer.py:284); paired == (iteration < pairnum//batchSize)."""
        be, T, F = self.be, self.T, self.F
        B = degraded.shape[0]
        Bg = B * self.world
        # The reference evaluates T(degraded) twice per iteration (:271 for the critic, :318 for the generator) with
        # the SAME transport-map parameters (T is only stepped at :346) and the same input: the two results are
        # identical, so it is evaluated once, with the activations the generator backward needs kept resident.
        T.zero_grad()
        out = T.forward(degraded, save=True)                         # :271 == :318
        fake = out
        # ---------------- critic ("F-sub"), trainer.py:262-280
        F.zero_grad()
        both = be.empty(2 * B, *target.shape[1:])
        be.axpby(target, None, both[:B], 1.0, 0.0)
        be.axpby(fake, None, both[B:], 1.0, 0.0)
        f_out = F.forward(both, save=True)                           # F(target), F(fake) in one sweep
        dsign = be.empty(2 * B)
        be.fill(dsign[:B], -1.0 / Bg)                                # -mean F(target)  :269
        be.fill(dsign[B:], 1.0 / Bg)                                 # +mean F(fake)    :274
        self.redF.begin()
        F.backward(dsign, wgrad=True, need_dx=False)
        self.redF.finish()
        self._comm_mark("F_critic", self.redF)
        if self.grad_probe is not None:
            self.grad_probe("F_critic")
        self.Fo.step()                                               # :280
        # ---------------- gradient penalty, own optimizer step, trainer.py:283-308
        F.zero_grad()
        interp = be.empty(*target.shape)
        be.lerp(target, fake, alpha, interp)                         # :286
        gp = be.empty(1)
        # (the penalty's gradients are produced first layer to last, i.e. from the END of the flat buffer: complete buckets
        # leave last one first through grad_tail_hook while the sweep continues; finish() issues what is left, the fc weights)
        self.redF.begin()
        F.gradient_penalty_backward(interp, 1.0 / Bg, gp)
        self.redF.finish()
        self._comm_mark("F_gp", self.redF)
        if self.grad_probe is not None:
            self.grad_probe("F_gp")
        self.Fo.step(F.n_live_gp)                                    # :308 (fc2.bias has no gradient)
        # ---------------- generator ("T-sub"), trainer.py:311-346
        F.zero_grad()
        fo = F.forward(out, save=True)                               # :319 (F has taken its two steps)
        dfo = be.empty(B)
        be.fill(dfo, -1.0 / Bg)                                      # -out_disc.mean()
        dout = F.backward(dfo, wgrad=False, need_dx=True)
        sums, spec, scal = be.empty(2 * B + 2), be.empty(B), be.empty(3)
        be.ot_reduce(degraded, out, target if paired else None, sums)
        par.all_reduce_scalars(sums[2 * B:], host_action=self.redT.host_action)   # global sum res^2 for the RMSE
        gF = None
        if self._any_spectral:
            gF = be.empty(*out.shape)
            be.ot_spectrum(degraded, out, de_id, gF, spec)
        be.ot_grad(degraded, out, target if paired else None, de_id, gF, sums, spec, dout, scal, self.sigma,
                   self.Sigma, Bg)
        self.redT.begin()
        T.backward(dout)                                             # :345
        self.redT.finish()
        self._comm_mark("T_gen", self.redT)
        if self.grad_probe is not None:
            self.grad_probe("T_gen")
        self.To.step()                                               # :346
        self.logs = dict(f_out=f_out, fo=fo, scal=scal, gp=gp, B=B, Bg=Bg, paired=paired)
        return out

    #: bench.py: a dict to fill with {half-step: (events of its gradient all-reduces)}; None = no timing
    comm_log = None

    def _comm_mark(self, tag, red):
        if self.comm_log is not None and red.timing is not None:
            self.comm_log[tag] = list(red.timing)
            red.timing.clear()

    def time_collectives(self, run_one):
        """Per-half-step gradient all-reduce time of one eagerly launched iteration (``run_one()`` runs it): HIP events around
        every bucket on the reducer's side stream.  Returns {half-step: {"ms", "mbytes", "buckets"}}."""
        self.comm_log = {}
        self.redT.timing, self.redF.timing = [], []
        try:
            run_one()
            torch.cuda.synchronize()
            out = {k: {"ms": round(sum(s.elapsed_time(e) for s, e, _ in v), 3), "mbytes": round(sum(n for _, _, n in v) / 1e6, 1),
                       "buckets": len(v)} for k, v in self.comm_log.items()}
        finally:
            self.comm_log = None
            self.redT.timing = self.redF.timing = None
        return out

    _any_spectral = True

    def set_de_ids(self, de_id_host: Sequence[int]):
        # With several ranks the flag must be the same everywhere: it is part of the launch-plan cache key (plan.py), and a rank
        # that misses the cache while another hits it would issue a different number of collectives.  The spectral branch
        # handles every de_id per sample on the device, so it is simply always taken in data-parallel runs.
        self._any_spectral = self.world > 1 or any(int(d) >= 3 for d in de_id_host)

    def scalars(self):
        """Loss values of the last iteration (forces a device sync; the reference does this every 10 its)."""
        L = self.logs
        B = L["B"]
        f_out = L["f_out"].detach().cpu().double()
        scal = L["scal"].detach().cpu().double()
        w = self.world
        # every logged value is the GLOBAL-batch quantity the single-process reference would print: local sums are
        # SUM all-reduced once (4 numbers); scal[0] (rmse) and scal[2] (mean|out-target|) already use the all-reduced sums
        # and gp was computed with the global 1/B (each rank holds its share).
        loc = torch.stack([f_out[:B].sum(), f_out[B:].sum(), L["fo"].detach().cpu().double().sum(), scal[1],
                           L["gp"].detach().cpu().double().view(())])
        if w > 1:
            par.all_reduce_scalars_host(loc)
        Bg = L["Bg"]
        loss_f = float(-loc[0] / Bg + loc[1] / Bg)
        loss_t = float(-loc[2] / Bg) + self.sigma * float(scal[0] + loc[3])
        if L["paired"]:
            loss_t += self.Sigma * float(scal[2])
        return dict(Loss_F=loss_f, Loss_T=loss_t, Loss_mse=float(scal[0]), gp=float(loc[4]))


# ------------------------------------------------------------------------------- reference-shaped entry points
def adjust_learning_rate(epoch):
    return opt.lr * (0.1 ** (epoch // opt.step))            # trainer.py:228-231


def train(training_data_loader, T_optimizer, F_optimizer, Tnet, Fnet, epoch, stepper: Optional[MinimaxStep] = None):
    """Same signature / behaviour as the reference's train() (trainer.py:234-360).  Batches are
    ([names, de_id], degraded, target) with CPU or device tensors."""
    lr = adjust_learning_rate(epoch - 1)
    T_optimizer.param_groups[0]["lr"] = lr / 2
    F_optimizer.param_groups[0]["lr"] = lr
    if par.rank() == 0:
        print("Epoch={}, lr={}".format(epoch, F_optimizer.param_groups[0]["lr"]))
    st = stepper or MinimaxStep(Tnet, Fnet, T_optimizer, F_optimizer, opt.sigma, opt.Sigma)
    dev, dt = Tnet.be.device, Tnet.store.flat.dtype
    # mean critic loss of the epoch over EVERY iteration, as the reference's Dloss.append(F_train_loss.data) does (:277, :360):
    # accumulated on the device from the critic outputs of each iteration (local sums; all-reduced once at the end)
    dl_acc, n_it = torch.zeros(1, device=dev, dtype=torch.float64), 0
    if hasattr(training_data_loader, "set_epoch"):
        training_data_loader.set_epoch(epoch)                              # a resumed run continues the data stream of its epoch
    # alpha ~ U[0,1) per GLOBAL sample index (the reference draws torch.rand(B,1,1,1) on the CPU RNG, :284): every rank
    # seeds the same generator, draws the global batch's values and keeps its own slice, so the union over ranks equals
    # the single-process global-batch draw for any world size (SURVEY.md 8e trap 5).
    gen = torch.Generator()
    gen.manual_seed((opt.seed if opt.seed is not None else 0) * 1000 + epoch)
    world, rank = par.world_size(), par.rank()
    for iteration, batch in enumerate(training_data_loader):
        ([_names, de_id], degraded, target) = batch
        degraded = degraded.to(dev, dt)
        target = target.to(dev, dt)
        de_host = [int(d) for d in de_id]
        st.set_de_ids(de_host)
        de_dev = torch.tensor(de_host, dtype=torch.int32, device=dev)
        Bl = target.size(0)
        alpha = torch.rand(Bl * world, generator=gen)[rank * Bl:(rank + 1) * Bl].to(dev, dt)
        paired = iteration < opt.pairnum // opt.batchSize                  # :338 (global batch size)
        out = st.run(degraded, target, de_dev, alpha, paired)
        f_out = st.logs["f_out"]                                           # [2 B_local]: F(target), F(fake)
        # the reference averages the PER-ITERATION means F(fake).mean() - F(target).mean() (:277, :360): each iteration's sum is
        # divided by its own global sample count (the last batch of a folder may be ragged), the all-reduce sums the ranks' shares
        dl_acc += (f_out[Bl:].sum() - f_out[:Bl].sum()).double() / (Bl * world)
        n_it += 1
        if iteration % 10 == 0:
            s = st.scalars()
            if par.rank() == 0:
                print("Epoch {}({}/{}):Loss_F: {:.5}, Loss_T: {:.5}, Loss_mse: {:.5}".format(
                    epoch, iteration, len(training_data_loader), s["Loss_F"], s["Loss_T"], s["Loss_mse"]))
            sd = getattr(st, "sample_dir", None)
            if sd:                                                         # sample dumps, trainer.py:355-358
                save_image(out, sd + "output.png")
                save_image(degraded, sd + "degraded.png")
                save_image(target, sd + "target.png")
                save_image(2 * (degraded - out), sd + "res.png")
    nan = float("nan")                                                     # the reference returns NaN here too (:360)
    if n_it == 0:
        return nan, nan, nan
    par.all_reduce_scalars(dl_acc)
    return nan, nan, float(dl_acc) / n_it


from .compat import shim as _shim  # noqa: E402


def _as_picklable(net, cls):
    """The HIP-backed network as a ``Net_Restormer.<cls>`` object that pickles like the reference's module."""
    NRshim = _shim()
    C = getattr(NRshim, cls)
    if isinstance(net, C):
        return net
    kw = {"decoder": net.decoder} if cls == "T_net" else {"patch_size": net.patch_size}
    return C.from_state_dict(net.state_dict(), **kw)


def save_checkpoint(Tnet, Fnet, epoch, T_optimizer=None, F_optimizer=None):
    """trainer.py:362-371: same path pattern and dict keys, and — like the reference — whole network OBJECTS under
    "Tnet"/"Fnet" (class paths ``Net_Restormer.T_net`` / ``F_net``; see rcot_amd/compat.py), so that
    ``torch.load(p)["Tnet"]`` can be called (tester.py:54) and ``.state_dict()`` read (trainer.py:105).  Superset:
    optimizer state under "T_optimizer"/"F_optimizer" (the reference loses RMSprop's square_avg on resume)."""
    path = "checkpoint/" + "model_" + str(opt.type) + "_" + "_" + str(opt.nEpochs) + "_" + str(opt.sigma) + ".pth"
    os.makedirs("checkpoint/", exist_ok=True)
    if not hasattr(Tnet, "decoder"):
        # the MPRNet backbone: plain state_dicts under the same keys (what main_mprnet's stock-ops form writes and reads; the
        # reference's own class for it lives in Net.py, which this package does not shadow)
        state = {"epoch": epoch, "Tnet": {k: v.cpu() for k, v in Tnet.state_dict().items()},
                 "Fnet": {k: v.cpu() for k, v in Fnet.state_dict().items()}, "backbone": "mprnet"}
    else:
        state = {"epoch": epoch, "Tnet": _as_picklable(Tnet, "T_net"), "Fnet": _as_picklable(Fnet, "F_net")}
    for key, o in (("T_optimizer", T_optimizer), ("F_optimizer", F_optimizer)):
        if o is not None:
            state[key] = o.state_dict()
    torch.save(state, path)
    print("Checkpoint saved to {}".format(path))
    return path


def _state_dict_of(obj):
    return obj if isinstance(obj, dict) else obj.state_dict()


# ------------------------------------------------------------------------------- validation (trainer.py:179-227)
def psnr(pred, gt, data_range: float = 1.0) -> float:
    """skimage.metrics.peak_signal_noise_ratio as the reference calls it (:225): 10 log10(range^2 / mse), float64."""
    import numpy as np
    err = float(np.mean((np.asarray(pred, dtype=np.float64) - np.asarray(gt, dtype=np.float64)) ** 2))
    return float("inf") if err == 0.0 else 10.0 * math.log10(data_range * data_range / err)


def evaluate(Tnet, deg_list, tar_list):
    """Whole-image inference + PSNR over the validation folders — reference trainer.py:179-227.

    Same walk and the same skip rules (:195-198: H or W not a multiple of 4, or shape mismatch, are skipped but still
    counted in the divisor :226), plus two guards the reference lacks: sizes that are multiples of 4 but not of 8 are
    skipped as well (the reference's three PixelUnshuffle(2) stages raise on them), and an empty list returns NaN
    instead of dividing by zero (:226).  Every size the reference accepts is processed (images whose 1/8-resolution plane has
    an odd pixel count, e.g. 200 x 200, run that level on width-padded, masked planes: T_net._lat_pad)."""
    import numpy as np
    from PIL import Image
    pp = 0.0
    if par.rank() == 0:
        print('----------validating-----------')
    dev = Tnet.be.device
    for deg_name, tar_name in zip(deg_list, tar_list):
        deg_img = np.array(Image.open(deg_name).convert('RGB'))
        tar_img = np.array(Image.open(tar_name).convert('RGB'))
        h, w = deg_img.shape[0], deg_img.shape[1]
        if (h % 4) or (w % 4) != 0 or deg_img.shape != tar_img.shape:
            continue
        if (h % 8) or (w % 8):
            continue
        x = torch.from_numpy(np.ascontiguousarray(deg_img.transpose(2, 0, 1))).float().div(255).unsqueeze(0).to(dev)
        out = Tnet(x)                                   # T_net.__call__: inference forward, nothing saved
        im = out.squeeze(0).cpu().numpy().transpose(1, 2, 0)
        gt = (tar_img.astype(np.float32) / 255.0)
        pp += psnr(im, gt, data_range=1)
    return pp / len(deg_list) if len(deg_list) else float("nan")


def save_image(tensor, path, nrow: int = 8, padding: int = 2):
    """torchvision.utils.save_image as the reference uses it for its sample dumps (trainer.py:355-358): a grid of the
    batch (nrow 8, 2-pixel black padding), values clamped to [0,1], 8-bit PNG."""
    import numpy as np
    from PIL import Image
    t = tensor.detach().float().cpu().clamp(0, 1)
    B, C, H, W = t.shape
    if B == 1:                                  # torchvision's make_grid returns a single image as it is (no border): the testers' dumps
        arr = t[0].mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        Image.fromarray(arr).save(path)
        return
    cols = min(nrow, B)
    rows = (B + cols - 1) // cols
    grid = torch.zeros(C, rows * (H + padding) + padding, cols * (W + padding) + padding)
    for i in range(B):
        r, c = divmod(i, cols)
        grid[:, padding + r * (H + padding):padding + r * (H + padding) + H,
             padding + c * (W + padding):padding + c * (W + padding) + W] = t[i]
    arr = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    Image.fromarray(arr).save(path)


def _warn_ignored_flags():
    d = parser.parse_args([])
    for flag in ("gpus", "cuda"):
        if getattr(opt, flag) != getattr(d, flag) and par.rank() == 0:
            print(f"note: --{flag} is accepted for CLI compatibility and ignored (one process per GPU; launch with "
                  f"torchrun to use several GPUs)")


def main_mprnet():
    """BASELINE configs[0]: ``Net.T_net`` (MPRNet) + ``Net_Restormer.F_net`` trained by the same loop.  With a GPU: both networks on
    the HIP kernels (rcot_amd/mprnet_hip.py) through the same ``MinimaxStep`` / launch plans as the Restormer
    backbone.  Without one (or with RCOT_MPRNET_STOCK=1, the A/B switch): stock PyTorch ops with torch autograd — the configuration
    BASELINE.json describes (CPU plumbing run).  Synthetic patches only: the reference's ``single`` mode lacks its ``--single_dir``
    flag upstream."""
    from .mprnet import FNetTorch, MPRNetT, torch_minimax_iteration
    from .synth import SyntheticLoader
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    d = parser.parse_args([])
    unsupported = [f for f in ("denoise_dir", "derain_dir", "dehaze_dir", "deblur_dir", "lowlight_dir", "single_dir", "degset", "tarset",
                               "data_file_dir", "pretrained") if getattr(opt, f) != getattr(d, f)]
    if unsupported:
        raise SystemExit("--backbone mprnet trains on seeded synthetic patches only; these flags would be ignored: "
                         + ", ".join("--" + f for f in unsupported))
    seed = opt.seed if opt.seed is not None else int.from_bytes(os.urandom(2), "little") % 10000 + 1
    print("Random Seed: ", seed)
    # NOT a fallback of the HIP path (that one raises without its library): BASELINE configs[0] is defined as this CPU run, and
    # RCOT_MPRNET_STOCK=1 asks for it explicitly on a GPU box (the A/B partner of rcot_amd/mprnet_hip.py)
    print(f"backbone mprnet: STOCK PyTorch ops with torch autograd on {dev} (" +
          ("RCOT_MPRNET_STOCK=1" if dev == "cuda" else "no GPU visible: BASELINE configs[0] as specified; the HIP kernels need a GPU") + ")")
    torch.manual_seed(seed)
    Tn, Fn = MPRNetT(seed=seed, device=dev), FNetTorch(opt.patch_size, seed=seed + 1, device=dev)
    if opt.resume:                                                    # trainer.py:96-103, state_dict checkpoints of this backbone
        if not os.path.isfile(opt.resume):
            raise SystemExit("=> no checkpoint found at '{}'".format(opt.resume))
        ck = torch.load(opt.resume, map_location=dev, weights_only=False)
        if ck.get("backbone") != "mprnet":
            raise SystemExit(f"{opt.resume} is not an mprnet-backbone checkpoint")
        Tn.load_state_dict(ck["Tnet"])
        Fn.load_state_dict(ck["Fnet"])
        opt.start_epoch = ck["epoch"] + 1
        print("=> loaded checkpoint '{}' (epoch {})".format(opt.resume, ck["epoch"]))
    mk = torch.optim.RMSprop if opt.optimizer == "RMSprop" else torch.optim.Adam
    To, Fo = mk(Tn.parameters(), lr=opt.lr / 2), mk(Fn.parameters(), lr=opt.lr)               # trainer.py:121-126
    loader = SyntheticLoader(opt.de_type, opt.batchSize, opt.patch_size, opt.iters, seed=seed, unpaired=(opt.pairnum == 0))
    for epoch in range(opt.start_epoch, opt.nEpochs + 1):
        gen = torch.Generator().manual_seed(seed * 1000 + epoch)      # the alpha stream of train() (both forms draw the same values)
        lr = adjust_learning_rate(epoch - 1)
        for g in To.param_groups:
            g["lr"] = lr / 2
        for g in Fo.param_groups:
            g["lr"] = lr
        print("Epoch={}, lr={}".format(epoch, lr))
        t0 = time.time()
        if hasattr(loader, "set_epoch"):
            loader.set_epoch(epoch)
        for iteration, ([_n, de_id], degraded, target) in enumerate(loader):
            alpha = torch.rand(target.size(0), generator=gen)
            s = torch_minimax_iteration(Tn, Fn, To, Fo, degraded.to(dev), target.to(dev), [int(d) for d in de_id], alpha.to(dev),
                                        opt.sigma, opt.Sigma, iteration < opt.pairnum // opt.batchSize)
            if iteration % 10 == 0 or opt.iters <= 10:
                print("Epoch {}({}/{}):Loss_F: {:.5}, Loss_T: {:.5}, Loss_mse: {:.5}".format(
                    epoch, iteration, len(loader), s["Loss_F"], s["Loss_T"], s["Loss_mse"]))
        print(f"epoch {epoch}: {len(loader) * opt.batchSize / (time.time() - t0):.2f} patches/s on {dev} (stock PyTorch ops)")
        os.makedirs("checkpoint/", exist_ok=True)
        path = "checkpoint/model_" + str(opt.type) + "_" + "_" + str(opt.nEpochs) + "_" + str(opt.sigma) + ".pth"     # :363
        torch.save({"epoch": epoch, "Tnet": Tn.state_dict(), "Fnet": Fn.state_dict(), "backbone": "mprnet"}, path)
        print("Checkpoint saved to {}".format(path))
    return Tn, Fn


def main(argv=None):
    global opt
    opt = parser.parse_args(argv)
    # --backbone mprnet: with a GPU the older transport map runs on the HIP kernels (rcot_amd/mprnet_hip.py) through everything below —
    # data folders, data parallelism, validation, launch plans; without one (or with RCOT_MPRNET_STOCK=1) the stock-ops loop
    hip_mprnet = opt.backbone == "mprnet" and torch.cuda.is_available() and os.environ.get("RCOT_MPRNET_STOCK", "0") != "1"
    if opt.backbone == "mprnet" and not hip_mprnet:
        return main_mprnet()
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
        if not torch.distributed.is_initialized():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
            # RCCL over xGMI; RCOT_DIST_BACKEND=gloo exists for test boxes with ONE GPU (RCCL refuses two ranks on a device)
            torch.distributed.init_process_group(os.environ.get("RCOT_DIST_BACKEND", "nccl"))
    world, rank = par.world_size(), par.rank()
    if rank == 0:
        print(opt)
    _warn_ignored_flags()
    if opt.batchSize % world or opt.batchSize < world:
        raise SystemExit(f"--batchSize {opt.batchSize} (global) must be a positive multiple of the world size {world}")
    # one seed for the whole job: rank 0 draws it (the reference draws an unseeded random one, :79) and broadcasts it, so
    # every replica builds identical networks; the parameters are broadcast once more after resume / pretrained loading.
    seed = opt.seed if opt.seed is not None else int.from_bytes(os.urandom(2), "little") % 10000 + 1
    seed = par.broadcast_int(seed, 0)
    opt.seed = seed
    if rank == 0:
        print("Random Seed: ", seed)
    torch.manual_seed(seed)
    from . import lib as _lib
    from .ops import default_backend
    from .ops import PREC_BY_NAME
    default_backend().prec = PREC_BY_NAME[opt.prec]
    default_backend().x6_packs = default_backend().x6_packs or opt.prec == "bf16x6"
    if hip_mprnet:
        from .mprnet_hip import MPRNetHip
        Tnet = MPRNetHip(seed=seed)                                        # Net.T_net() of Net.py, same seeded initial parameters as the stock-ops form
        if rank == 0:
            print("backbone mprnet: Net.T_net on the HIP kernels")
    else:
        Tnet = _make_net("T_net", decoder=True, seed=seed)                 # trainer.py:92
    Fnet = _make_net("F_net", patch_size=opt.patch_size, seed=seed + 1)    # :93
    T_opt, F_opt = make_optimizers(Tnet, Fnet, opt.optimizer, opt.lr)
    from .compat import load_checkpoint
    if opt.resume:
        if os.path.isfile(opt.resume):
            ck = load_checkpoint(opt.resume)
            if hip_mprnet != (isinstance(ck, dict) and ck.get("backbone") == "mprnet"):
                raise SystemExit(f"{opt.resume}: checkpoint and --backbone {opt.backbone} do not match")
            opt.start_epoch = ck["epoch"] + 1
            Tnet.load_state_dict(_state_dict_of(ck["Tnet"]))
            Fnet.load_state_dict(_state_dict_of(ck["Fnet"]))
            if "T_optimizer" in ck:
                T_opt.load_state_dict(ck["T_optimizer"])
            if "F_optimizer" in ck:
                F_opt.load_state_dict(ck["F_optimizer"])
        elif rank == 0:
            print("=> no checkpoint found at '{}'".format(opt.resume))
    if opt.pretrained:
        if os.path.isfile(opt.pretrained):
            w = load_checkpoint(opt.pretrained)
            Tnet.load_state_dict(_state_dict_of(w["model"]))
            Fnet.load_state_dict(_state_dict_of(w["discr"]))
        elif rank == 0:
            print("=> no model found at '{}'".format(opt.pretrained))
    for net in (Tnet, Fnet):
        par.broadcast_flat(net.store.flat, 0)
        if world > 1 and hasattr(net, "repack"):
            net.repack()
    for o in (T_opt, F_opt):
        o.broadcast_state()
    if opt.synthetic:
        from .synth import SyntheticLoader
        loader = SyntheticLoader(opt.de_type, opt.batchSize // world, opt.patch_size, opt.iters, seed=seed, rank=rank,
                                 world=world, unpaired=(opt.pairnum == 0))
    else:
        from .data import FolderLoader
        loader = FolderLoader(opt, opt.batchSize // world, seed=seed, rank=rank, world=world, threads=opt.threads)
    import glob
    deg_list, tar_list = sorted(glob.glob(opt.degset + "*")), sorted(glob.glob(opt.tarset + "*"))   # :137-141
    stepper = MinimaxStep(Tnet, Fnet, T_opt, F_opt, opt.sigma, opt.Sigma)
    stepper.sample_dir = "./checksample/" + str(opt.type) + "/" if rank == 0 else None
    TLOSS, PLOSS = [], []
    for epoch in range(opt.start_epoch, opt.nEpochs + 1):
        t0 = time.time()
        a, b, c = train(loader, T_opt, F_opt, Tnet, Fnet, epoch, stepper)
        torch.cuda.synchronize()
        if rank == 0:
            dt = time.time() - t0
            print(f"epoch {epoch}: {len(loader) * opt.batchSize / dt:.1f} patches/s")
            p = evaluate(Tnet, deg_list, tar_list)                         # :149
            os.makedirs("./checksample/" + str(opt.type), exist_ok=True)
            with open("./checksample/" + str(opt.type) + "/validation_results.txt", "a") as f:      # :151-153
                f.write(f"Patchsize {opt.patch_size} Epoch {epoch}, psnr {p:.4f}, Batchsize {opt.batchSize}\n")
            TLOSS.append(format(b / 2))                                    # :157-162 (len(data_list) == 2)
            PLOSS.append(format(c / 2))
            try:
                import scipy.io as scio
                scio.savemat('TLOSSrain.mat', {'TLOSS': TLOSS})            # :163-164
                scio.savemat('PLOSSrain.mat', {'PLOSS': PLOSS})
            except ImportError:
                pass
            save_checkpoint(Tnet, Fnet, epoch, T_opt, F_opt)               # :165 (mprnet backbone: the state_dict form)
        if world > 1:
            torch.distributed.barrier()
    return Tnet, Fnet


def _make_net(cls, **kw):
    """The networks are built through the top-level ``Net_Restormer`` shim so that checkpoints pickle by that path."""
    return getattr(_shim(), cls)(**kw)


if __name__ == "__main__":
    main()
