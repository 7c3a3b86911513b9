"""Data-parallel layer for the RCOT minimax step: one process per GPU, RCCL over xGMI.

The reference is single-GPU; data parallelism over the batch is the only
sharding the path admits (every op is per-sample).  Gradients live in one flat fp32 buffer per
network, ordered by the time they become final in the backward sweep, so SUM all-reduces of
completed ranges ("buckets") are issued on a side HIP stream while the sweep continues, and the
fused optimizer waits on one event.  Local losses are defined with global-batch denominators, so a
plain SUM reproduces the single-process global-batch gradient (mean terms, the batch-SUMMED Fourier
penalty and the global-batch RMSE all come out right).
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist


class GradReducer:
    """Bucketed, overlapped all-reduce(SUM) of a flat gradient buffer."""

    def __init__(self, flat_grad: torch.Tensor, n_live: int, bucket_elems: int = 8 << 20, group=None):
        self.flat, self.n_live, self.group = flat_grad, n_live, group
        # RCOT_FORCE_REDUCER=1 exercises the bucketed side-stream path even at world size 1 (single-GPU test boxes)
        forced = os.environ.get("RCOT_FORCE_REDUCER") == "1"
        self.enabled = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or forced)
        self.bounds: List[int] = list(range(0, n_live, bucket_elems)) + [n_live]
        self.next_bucket = 0                     # buckets [0, next_bucket) have left from the front (ready)
        self.tail_bucket = len(self.bounds) - 1  # buckets [tail_bucket, end) have left from the back (ready_tail)
        self.cuda = flat_grad.is_cuda
        self.side = torch.cuda.Stream() if (self.enabled and self.cuda) else None
        self.handles = []
        #: set by rcot_amd.plan while the iteration is being recorded into a launch plan: collectives are host-driven, so
        #: ``host_action(fn)`` runs ``fn`` now and keeps it at the same position of every replay
        self.host_action = None
        #: bench.py sets this to a list to collect (start event, end event, bytes) of every bucket's all-reduce on the side stream
        self.timing = None

    def _do(self, fn):
        if self.host_action is not None:
            self.host_action(fn)
        else:
            fn()

    def begin(self):
        self.next_bucket = 0
        self.tail_bucket = len(self.bounds) - 1
        self.handles = []

    def ready(self, n_final: int):
        """grad[0:n_final) is final: launch every bucket that is now complete."""
        if not self.enabled:
            return
        while self.next_bucket < self.tail_bucket and self.bounds[self.next_bucket + 1] <= n_final:
            lo, hi = self.bounds[self.next_bucket], self.bounds[self.next_bucket + 1]
            self._do(lambda lo=lo, hi=hi: self._launch(lo, hi))
            self.next_bucket += 1

    def ready_tail(self, n_from: int):
        """grad[n_from:n_live) is final (a sweep that fills the buffer from its END, e.g. the gradient penalty's first-to-last
        layer pass): launch every complete bucket, last one first.  Every rank sees the same sequence of ready()/ready_tail()
        calls, so the collectives are issued in the same order everywhere."""
        if not self.enabled:
            return
        while self.tail_bucket > self.next_bucket and self.bounds[self.tail_bucket - 1] >= n_from:
            lo, hi = self.bounds[self.tail_bucket - 1], self.bounds[self.tail_bucket]
            self._do(lambda lo=lo, hi=hi: self._launch(lo, hi))
            self.tail_bucket -= 1

    def _launch(self, lo, hi):
        chunk = self.flat[lo:hi]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.side.wait_event(ev)
            with torch.cuda.stream(self.side):
                if self.timing is not None:
                    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t0.record(self.side)
                dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group)
                if self.timing is not None:
                    t1.record(self.side)
                    self.timing.append((t0, t1, chunk.numel() * 4))
        else:
            self.handles.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Flush remaining buckets and make the compute stream wait for the reductions."""
        if not self.enabled:
            return
        self.ready(self.n_live)
        if self.cuda:
            self._do(lambda: torch.cuda.current_stream().wait_stream(self.side))
        else:
            for h in self.handles:
                h.wait()
            self.handles = []


def all_reduce_scalars(t: torch.Tensor, group=None, host_action=None):
    """SUM all-reduce of a small tensor (the global sum of res^2 for the RMSE term).  ``host_action``: see GradReducer."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or os.environ.get("RCOT_FORCE_REDUCER") == "1"):
        fn = lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        if host_action is not None:
            host_action(fn)
        else:
            fn()


def all_reduce_scalars_host(t: torch.Tensor, group=None):
    """In-place SUM all-reduce of a small HOST tensor (logged scalars); staged through the device for RCCL."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return
    if dist.get_backend(group) == "nccl":
        d = t.to("cuda")
        dist.all_reduce(d, op=dist.ReduceOp.SUM, group=group)
        t.copy_(d.cpu())
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def broadcast_flat(t: torch.Tensor, src: int = 0, group=None):
    """Broadcast a flat parameter buffer from ``src`` so that every replica starts from identical weights."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(t, src=src, group=group)


def broadcast_int(v: int, src: int = 0, group=None) -> int:
    """One integer (the run's seed) from ``src`` to every rank."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return int(v)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([int(v)], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=src, group=group)
    return int(t.item())


def world_size(group=None) -> int:
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group)
    return 1


def rank(group=None) -> int:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group)
    return 0
