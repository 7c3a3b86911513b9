"""Tensor-level front end of the HIP kernels (``HipBackend``).

Every method takes torch tensors / views that live on the MI355X, checks the layout
contract of include/rcot_hip.h, and launches the matching ``rcot_*`` entry point on the
calling thread's current HIP stream.  Outputs are written in place into caller-provided
tensors.  PyTorch is used only for memory (allocation, views) and streams.

There is deliberately no alternative implementation here: constructing a HipBackend
without a GPU or without librcot_hip.so raises.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import os
import sys

import ctypes as C

import torch

from . import lib as _lib

LN = Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]]   # (mu, rs, weight, bias)


ctypes_ll = C.c_longlong


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda i: torch.cuda.current_stream(i).cuda_stream)


def _loaded_hip_runtime() -> str:
    """path of the libamdhip64 this process has already mapped (torch's), so that dlopen hands back that instance; the bare soname
    when it cannot be told (then the loader's search order decides, which is the same file on a stock ROCm image)"""
    try:
        with open("/proc/self/maps") as f:
            for ln in f:
                if "libamdhip64" in ln:
                    return ln.split(None, 5)[5].strip()
    except (OSError, IndexError):
        pass
    return "libamdhip64.so"


class _Handover:
    """Stream-to-stream ordering on ONE device with events that carry no system-scope fence (hipEventDisableTiming |
    hipEventDisableSystemFence): `dst` waits for what `src` has enqueued so far.  The schedule hands work between its two streams
    ~400 times per iteration, and the stream that RECORDS pays for every hand-over: 4.2-6.6 us with torch's events (their release is
    system-scope: the L2 write-back a host or peer reader would need) against 2.5-4.2 us with these
    (scripts/micro/event_cost.py, profiles/r06_event_cost.txt).  Both sides are kernels of this process on this device, which a
    device-scope release orders; host reads go through torch's own synchronisation.  One event per hand-over SITE (it is re-recorded
    at every replay: a wait captures the event as it stands when the wait is enqueued)."""
    _hip = None
    FLAGS = 0x2 | 0x20000000         # hipEventDisableTiming | hipEventDisableSystemFence

    def __init__(self):
        cls = _Handover
        if cls._hip is None:
            cls._hip = C.CDLL(_loaded_hip_runtime())   # the HIP runtime torch has mapped: events and streams must come from ONE runtime
            cls._hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
            cls._hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
            cls._hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
            cls._hip.hipEventDestroy.argtypes = [C.c_void_p]
        self.ev = C.c_void_p()
        rc = cls._hip.hipEventCreateWithFlags(C.byref(self.ev), cls.FLAGS)
        if rc != 0:
            raise _lib.RcotKernelError(f"hipEventCreateWithFlags: HIP error {rc}")

    def record(self, src: int):
        rc = self._hip.hipEventRecord(self.ev, src)
        if rc != 0:
            raise _lib.RcotKernelError(f"hipEventRecord: HIP error {rc}")

    def wait(self, dst: int):
        rc = self._hip.hipStreamWaitEvent(dst, self.ev, 0)
        if rc != 0:
            raise _lib.RcotKernelError(f"hipStreamWaitEvent: HIP error {rc}")

    def __call__(self, src: int, dst: int):
        self.record(src)
        self.wait(dst)

    def __del__(self):
        try:
            # (not at interpreter shutdown: the HIP runtime may already be unloading, and the process is about to release everything)
            if self.ev and not sys.is_finalizing():
                self._hip.hipEventDestroy(self.ev)
        except Exception:
            pass

_DEFAULT = {}

#: arithmetic of the big MFMA products by name (include/rcot_hip.h RCOT_PREC_*): "fp32" exact fp32 MFMA; "bf16x3" two-term split,
#: three products (~2^-16 per product); "bf16x6" three-term split, six products: fp32-class results from the bf16 pipe (weight
#: projections on the producer / consumer kernel; everything else runs the exact-fp32 kernels)
PREC_BY_NAME = {"fp32": _lib.PREC_FP32, "bf16x3": _lib.PREC_BF16X3, "bf16x6": _lib.PREC_BF16X6, "bf16x1": _lib.PREC_BF16X1}
_TWO_TERM = (_lib.PREC_BF16X3, _lib.PREC_BF16X1)      # the arithmetics that run on the two-term split kernels and packs
#: ONE default for the product (HipBackend(), trainer.py --prec, bench.py --prec; RCOT_GEMM_PREC overrides): exact fp32, the
#: reference's arithmetic.  bf16x6 gives results as close to fp64 (tests/test_x3_gpu.py::test_x6_is_as_accurate_as_the_fp32_kernel;
#: every parity test holds the fp32 bars in it) 2 % faster end to end, bf16x3 results within the north_star tolerances 15 % faster.
DEFAULT_PREC = "fp32"


def default_backend():
    """One shared HipBackend (and workspace) per device for every network of the process."""
    dev = torch.cuda.current_device() if torch.cuda.is_available() else -1
    if dev not in _DEFAULT:
        _DEFAULT[dev] = HipBackend()
    return _DEFAULT[dev]


class HipBackend:
    """Launches librcot_hip.so kernels.  One instance per device."""

    name = "hip"

    def __init__(self, device: Optional[torch.device] = None, workspace_bytes: int = 256 << 20):
        if not torch.cuda.is_available():
            raise _lib.RcotLibraryError("no HIP device visible: the RCOT hot path has no CPU fallback")
        self.L = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.ws = torch.empty(workspace_bytes // 4, dtype=torch.float32, device=self.device)
        self.ws_bytes = self.ws.numel() * 4
        # weight-gradient overlap: leaf kernels of the backward sweep run on a second HIP stream (own split-K workspace)
        # while the data-gradient chain continues; RCOT_OVERLAP=0 keeps everything on one stream
        # arithmetic of the big MFMA products (PREC_BY_NAME above).  RCOT_GEMM_PREC selects the process default (DEFAULT_PREC);
        # set ``backend.prec`` to switch at run time.
        self.prec = PREC_BY_NAME[os.environ.get("RCOT_GEMM_PREC", DEFAULT_PREC)]
        self._poison = os.environ.get("RCOT_POISON", "0") == "1"
        self.overlap = os.environ.get("RCOT_OVERLAP", "1") != "0"
        self.attn_core = os.environ.get("RCOT_ATTN_CORE", "1") != "0"      # A/B switch: rcot_attn_core_fwd vs the four separate launches
        self.attn_core_maxn = int(os.environ.get("RCOT_ATTN_CORE_MAXN", "4096"))   # largest plane rcot_attn_core_fwd is used on (it takes <= 65536; slower above 4096)
        self.multi_launch = os.environ.get("RCOT_MULTI", "1") != "0"       # A/B switch: dV, dQ, dK of a block from one launch (rcot_gemm_kmajor_multi)
        self.ln_fused = os.environ.get("RCOT_LN_FUSED", "1") != "0"        # A/B switch: LN statistics made by the projection kernel
        self.pair_launch = os.environ.get("RCOT_PAIR", "1") != "0"         # A/B switch: data + weight gradient of a 1x1 from one launch
        self.prod_stats = os.environ.get("RCOT_PROD_STATS", "1") != "0"     # A/B switch: LayerNorm statistics made by the epilogue of the product that stores the tensor (round 6)
        self._raw_events = os.environ.get("RCOT_RAW_EVENTS", "1") != "0"   # A/B switch: hand-overs between the two streams on fence-free HIP events (_Handover)
        self._ev_ring, self._ev_next = [], -1
        # networks built on this backend also keep the THREE-term weight packs of the bf16x6 arithmetic (1.5x the two-term packs,
        # refreshed with them after every optimizer step): on when that arithmetic is the process default, or asked for
        self._x6_nt = os.environ.get("RCOT_X6_NT", "1") != "0"
        self.x6_packs = self.prec == _lib.PREC_BF16X6 or os.environ.get("RCOT_X6_PACKS", "0") == "1"
        self._side = torch.cuda.Stream(device=self.device) if self.overlap else None
        self._ws_side = torch.empty_like(self.ws) if self.overlap else None
        # split-K slabs of weight gradients that wait for block_param_reduce(): their own arena (self.ws is reused by every
        # split-K launch that follows on the same stream)
        # (two generations, like the LayerNorm scratch below: the launch that closes a block reads the slabs on the SIDE stream while
        # the main stream's paired data/weight-gradient launches of the next block already write the other generation)
        self._ws_slabs_gen = [torch.empty_like(self.ws), torch.empty_like(self.ws)]
        self._held = []
        self._pcm_cache = {}                     # padded-plane geometries, operand buffers and index tables of conv_pcm_*
        # entries whose device addresses are baked into recorded launch plans (touched while ``pcm_pinning`` is set: LaunchPlan.record
        # collects them in ``_pcm_touched``): key -> number of live plans that refer to it; _pcm_trim() never evicts them
        self._pcm_pinned = {}
        self._pcm_touched = set()
        self.pcm_pinning = False
        self._plan = None                        # the LaunchPlan being recorded on this backend (rcot_amd/plan.py), else None
        self._side_pending = False
        # deferred LayerNorm parameter-gradient partials of one block (<= 1024 rows x 2*512 columns each), in TWO generations:
        # the launch that closes a block (block_param_reduce) runs on the side stream behind that block's weight-gradient
        # kernels while the main stream is already in the next block, which therefore writes the other generation
        self._ln_scratch = [[torch.empty(1024 * 1024, dtype=torch.float32, device=self.device) for _ in range(2)] for _ in range(2)]
        self._ln_rows = 0
        self._gen = 0
        self._gen_event = [None, None]          # side-stream event after the reduce that read generation g
        self._held_gen = [[], []]               # tensors that reduce (and the side kernels before it) still read
        # Side-stream policy, from full-iteration A/B runs under launch plans (round 4, B=8 128x128, ms per iteration):
        #   deferred block closes (the parameter reduce of block i under block i+1): fp32 84.3 -> 83.6 WITHOUT, bf16x3 74.5 -> 74.3: off;
        #   unpaired 1x1 weight gradients next to the data-gradient chain: fp32 83.6 with / 87.3 without (the fp32 products are
        #   MFMA-bound and leave bandwidth to a neighbour), bf16x3 74.3 with / 73.3 without (both HBM-bound: the neighbour only
        #   takes the bandwidth ln_bwd and the data gradient need): on for fp32 / bf16x6, off for bf16x3.
        #   round 5 (profiles/r05_ab_sched.txt, two runs each): deferred closes fp32 77.21 / 77.31 -> 77.49 / 77.41 (still off), bf16x6
        #   74.09 / 74.29 -> 73.85 / 73.71: on for bf16x6 (its weight gradients got shorter: the close no longer waits for the chain).
        #   round 6 (fence-free hand-overs; profiles/r06_ab_sched_nofence.txt): deferred closes fp32 77.06 -> 78.04 (off), bf16x6
        #   73.03 -> 72.48 (on), bf16x3 70.13 -> 70.71 (off); side-stream weight gradients now also pay in bf16x3 (71.9 -> 71.3).
        self._defer_close_env = os.environ.get("RCOT_DEFER_CLOSE")
        self._side_wgrad_env = os.environ.get("RCOT_SIDE_WGRAD")

    # ------------------------------------------------------------------ leaf-kernel overlap
    @property
    def defer_close(self):
        """the launch that closes a block runs on the side stream under the next block (policy above); RCOT_DEFER_CLOSE=0/1 overrides"""
        if self._defer_close_env is not None:
            return self._defer_close_env != "0"
        return self.prec == _lib.PREC_BF16X6

    @property
    def side_wgrad(self):
        """unpaired 1x1 weight gradients on the side stream (policy above); RCOT_SIDE_WGRAD=0/1 overrides"""
        if self._side_wgrad_env is not None:
            return self._side_wgrad_env != "0"
        # round 6: on in every arithmetic.  With the hand-overs on fence-free events (_Handover) the two-term arithmetics gain too:
        # bf16x3 71.9 (off) / 71.3 (on) ms per iteration in one call (profiles/r06_ab_sched_nofence.txt); rounds 4-5 had it off there
        return True

    def side_run(self, fn, *hold):
        """Run ``fn`` (kernel launches that only READ ``hold`` tensors and WRITE parameter gradients) on the side
        stream, ordered after everything enqueued so far on the current stream.  ``hold`` stays referenced until
        side_join(), so the caching allocator cannot hand those blocks out while the side kernels are pending."""
        if not self.overlap:
            fn()
            return
        self._handover_to_side()
        ws = self.ws
        self.ws = self._ws_side
        try:
            with torch.cuda.stream(self._side):
                fn()
        finally:
            self.ws = ws
        self._held.extend(hold)
        self._side_pending = True

    def side_join(self):
        """The current stream waits for the side stream; the held tensors may be released afterwards."""
        if self._side_pending:
            self._handover_from_side()
            self._side_pending = False
        self._held.clear()
        for g in (0, 1):
            self._gen_event[g] = None
            self._held_gen[g].clear()

    def _new_handover(self):
        """an event for one hand-over site: owned by the launch plan being recorded (re-recorded at every replay), else the next of a
        ring that the eagerly walked schedule re-uses (a wait keeps the state the event had when the wait was enqueued)"""
        if self._plan is not None:
            return _Handover()
        if len(self._ev_ring) < 2048:
            self._ev_ring.append(_Handover())
            return self._ev_ring[-1]
        self._ev_next = (self._ev_next + 1) % len(self._ev_ring)
        return self._ev_ring[self._ev_next]

    def _handover_to_side(self):
        """the side stream waits for what the calling stream has enqueued so far"""
        if self._raw_events:
            h, side, dev = self._new_handover(), self._side.cuda_stream, self._dev_index
            self._host(lambda: h(_raw_stream(dev), side))
        else:
            self._host(lambda: self._side.wait_stream(torch.cuda.current_stream()))

    def _handover_from_side(self):
        """the calling stream waits for what the side stream has enqueued so far"""
        if self._raw_events:
            h, side, dev = self._new_handover(), self._side.cuda_stream, self._dev_index
            self._host(lambda: h(side, _raw_stream(dev)))
        else:
            self._host(lambda: torch.cuda.current_stream().wait_stream(self._side))

    def _host(self, fn):
        """a cross-stream wait / event of the schedule: runs now; while a launch plan is recorded (rcot_amd/plan.py) it is also kept,
        at this position, for every replay"""
        if self._plan is not None:
            self._plan.host_action(fn)
        else:
            fn()

    # ------------------------------------------------------------------ plumbing
    def empty(self, *shape):
        if self._poison:                           # RCOT_POISON=1 (debugging): reads of unwritten memory surface as NaN
            return torch.full(shape, float("nan"), dtype=torch.float32, device=self.device)
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    def zeros(self, *shape):
        t = torch.empty(*shape, dtype=torch.float32, device=self.device)
        if t.numel():
            self.fill(t, 0.0)                      # (rcot_fill: no at::native launches from this package, construction included)
        return t

    def _st(self):
        # the raw hipStream_t of the calling thread's current stream on this device.  torch.cuda.current_stream().cuda_stream
        # builds a Stream object per call (device-index resolution, lazy-init checks): 3 400 calls per iteration were a quarter of
        # the host's enqueue time (scripts/host_profile.py)
        return _raw_stream(self._dev_index)

    @staticmethod
    def _chk(t: torch.Tensor, what: str):
        if t.dtype != torch.float32 or not t.is_cuda:
            raise _lib.RcotKernelError(f"{what}: expected a float32 tensor on the HIP device")

    @staticmethod
    def _bcn(t: torch.Tensor, what: str):
        """View as [B, C, N] with unit pixel stride and dense channels; returns (B, C, N, batch_stride)."""
        if t.dim() == 4:
            B, Cc, H, W = t.shape
            N = H * W
            ok = t.stride(3) == 1 and t.stride(2) == W and t.stride(1) == N
        else:
            B, Cc, N = t.shape
            ok = t.stride(2) == 1 and t.stride(1) == N
        if not ok:
            raise _lib.RcotKernelError(f"{what}: channel planes must be dense NCHW")
        return B, Cc, N, t.stride(0)

    # ------------------------------------------------------------------ K-major fast path
    @staticmethod
    def pack_shapes(Co: int, Ci: int):
        """Shapes of the (WT, WP) packs of a [Co, Ci] 1x1 weight (see rcot_pack_weight)."""
        r16, r4 = (lambda v: (v + 15) // 16 * 16), (lambda v: (v + 3) // 4 * 4)
        return (r16(Ci), r4(Co)), (r16(Co), r4(Ci))

    @staticmethod
    def fold_shapes(Co: int, Ci: int):
        """Shapes of the LN-fold pack of a [Co, Ci] 1x1 weight that follows a LayerNorm: (WTf, c12)."""
        r16, r4 = (lambda v: (v + 15) // 16 * 16), (lambda v: (v + 3) // 4 * 4)
        return (r16(Ci), r4(Co)), (2, r4(Co))

    @staticmethod
    def split_shapes(Co: int, Ci: int):
        """Sizes (in floats, 1-D) of the pre-split fragment packs of a [Co, Ci] 1x1 weight: (WTs and WTfs, WPs)."""
        c = lambda v, q: (v + q - 1) // q
        return (c(Ci, 16) * c(Co, 32) * 512,), (c(Co, 16) * c(Ci, 32) * 512,)

    @staticmethod
    def split6_shapes(Co: int, Ci: int):
        """the THREE-term form of split_shapes (bf16x6 arithmetic): 3 KiB records"""
        c = lambda v, q: (v + q - 1) // q
        return (c(Ci, 16) * c(Co, 32) * 768,), (c(Co, 16) * c(Ci, 32) * 768,)

    @property
    def prec_nt(self):
        """the arithmetic of the pixel-reduction products (weight gradients, Gram matrices): the backend's; RCOT_X6_NT=0 keeps
        them exact fp32 under bf16x6 (the round-4 first form, for A/B runs)"""
        if self.prec == _lib.PREC_BF16X6 and not self._x6_nt:
            return _lib.PREC_FP32
        return self.prec

    def pack_weight(self, W, WT, WP, fold=None, split=None, split6=None):
        """``fold`` = (ln_w, ln_b, WTf, c12): also write the LN-folded forward operand; ``split`` = (WTs, WPs, WTfs | None):
        also write the pre-split bf16 fragment packs of the bf16x3 producer / consumer kernel (rcot_pack_weight)."""
        Co, Ci = W.shape
        assert W.stride(1) == 1 and WT.is_contiguous() and WP.is_contiguous()
        assert (tuple(WT.shape), tuple(WP.shape)) == self.pack_shapes(Co, Ci)
        f = (None, None, None, None)
        if fold is not None:
            lnw, lnb, WTf, c12 = fold
            assert (tuple(WTf.shape), tuple(c12.shape)) == self.fold_shapes(Co, Ci) and WTf.is_contiguous() and c12.is_contiguous()
            f = (lnw.data_ptr(), lnb.data_ptr(), WTf.data_ptr(), c12.data_ptr())
        sp = (None, None, None)
        if split is not None:
            st_, sp_ = self.split_shapes(Co, Ci)
            assert tuple(split[0].shape) == st_ and tuple(split[1].shape) == sp_ and (split[2] is None or tuple(split[2].shape) == st_)
            sp = (split[0].data_ptr(), split[1].data_ptr(), _ptr(split[2]))
        s6 = (None, None, None)
        if split6 is not None:
            st6, sp6 = self.split6_shapes(Co, Ci)
            assert tuple(split6[0].shape) == st6 and tuple(split6[1].shape) == sp6 and (split6[2] is None or tuple(split6[2].shape) == st6)
            s6 = (split6[0].data_ptr(), split6[1].data_ptr(), _ptr(split6[2]))
        _lib.check(self.L.rcot_pack_weight(W.data_ptr(), W.stride(0), Co, Ci, WT.data_ptr(), WP.data_ptr(), *f, *sp, *s6, self._st()),
                   "rcot_pack_weight")

    def pack_table(self, items, prec=None):
        """Device descriptors for pack_weights(): items = [(W, WT, WP[, fold[, split[, split6]]]), ...] with fold = (ln_w, ln_b, WTf,
        c12) or None, split = (WTs, WPs, WTfs | None) or None, split6 likewise (pointers must stay valid).  ``prec``: leave out
        the fragment packs that arithmetic never reads (the two-term packs serve bf16x3 only, the three-term packs bf16x6 only) —
        a repack then writes a third (fp32) / half (bf16x3) of the bytes; None = every pack given."""
        rows, c2d, chunk = [], [], 0
        want3 = prec is None or prec in _TWO_TERM
        want6 = prec is None or prec == _lib.PREC_BF16X6
        # the LN-folded operand WTf and its row constants serve the split arithmetics (LayerNorm applied in the epilogue); the exact-fp32
        # kernel normalises the fragments in its loop and reads WT (round 6: a third of the repack's bytes for qkv / project_in less;
        # RCOT_F32_PC=1, the opt-in exact product on the producer / consumer path, does read them)
        want_fold = prec is None or prec != _lib.PREC_FP32 or os.environ.get("RCOT_F32_PC") == "1"
        r16, r4 = (lambda v: (v + 15) // 16 * 16), (lambda v: (v + 3) // 4 * 4)
        for d, item in enumerate(items):
            W, WT, WP = item[:3]
            fold = item[3] if len(item) > 3 and want_fold else None
            split = item[4] if len(item) > 4 and want3 else None
            split6 = item[5] if len(item) > 5 and want6 else None
            Co, Ci = W.shape
            assert W.stride(1) == 1 and (tuple(WT.shape), tuple(WP.shape)) == self.pack_shapes(Co, Ci)
            nt, np_ = r16(Ci) * r4(Co), r16(Co) * r4(Ci)
            fp = [0, 0, 0, 0]
            sp = [0, 0, 0]
            units = nt + np_
            if fold is not None:
                lnw, lnb, WTf, c12 = fold
                assert (tuple(WTf.shape), tuple(c12.shape)) == self.fold_shapes(Co, Ci)
                fp = [lnw.data_ptr(), lnb.data_ptr(), WTf.data_ptr(), c12.data_ptr()]
                units += nt
            if split is not None:
                (st_,), (sp_,) = self.split_shapes(Co, Ci)
                assert split[0].numel() == st_ and split[1].numel() == sp_ and (split[2] is None or fold is not None)
                sp = [split[0].data_ptr(), split[1].data_ptr(), 0 if split[2] is None else split[2].data_ptr()]
                units += st_ // 8 + sp_ // 8 + (st_ // 8 if split[2] is not None else 0)     # one unit per 32-byte record
            s6 = [0, 0, 0]
            if split6 is not None:
                (st6,), (sp6,) = self.split6_shapes(Co, Ci)
                assert split6[0].numel() == st6 and split6[1].numel() == sp6 and (split6[2] is None or fold is not None)
                s6 = [split6[0].data_ptr(), split6[1].data_ptr(), 0 if split6[2] is None else split6[2].data_ptr()]
                units += st6 // 12 + sp6 // 12 + (st6 // 12 if split6[2] is not None else 0)    # one unit per 48-byte record
            n = (units + 1023) // 1024 + ((r4(Co) + 63) // 64 if fold is not None else 0)
            rows.append([W.data_ptr(), W.stride(0), Co, Ci, WT.data_ptr(), WP.data_ptr(), chunk] + fp + sp + s6 + [0, 0, 0])
            c2d.extend([d] * n)
            chunk += n
        return (torch.tensor(rows, dtype=torch.int64, device=self.device),
                torch.tensor(c2d, dtype=torch.int32, device=self.device)), chunk

    def pack_weights(self, table, nchunks):
        _lib.check(self.L.rcot_pack_weights(table[0].data_ptr(), table[1].data_ptr(), nchunks, self._st()), "rcot_pack_weights")

    @staticmethod
    def kmajor_ok(N: int, K: int, a_rows: int) -> bool:
        return N % 64 == 0 and a_rows >= (K + 15) // 16 * 16

    @staticmethod
    def kmajor_worth(M: int, N: int, Z: int) -> bool:
        """The LDS-DMA kernel has no split-K: take it when its 128x128 or 64x64 tiling yields enough workgroups."""
        return N % 64 == 0 and ((M + 63) // 64) * (N // 64) * Z >= 128

    def gemm_kmajor(self, At, Bm, C, M: int, K: int, R=None, rowscale=None, ln: LN = None, beta: float = 0.0, fold=None,
                    split=None, ln_compute: bool = False):
        """C[zo,zi] (M x N) = A @ LN?(Bm) + rowscale*R + beta*C with A given transposed: At [Zo,Zi,rows>=ceil16(K),>=M]
        (rows >= K zero).  Bm: [Zo,Zi,K,N]; C/R: [Zo,Zi,M,N]; N % 128 == 0.  ``fold`` = (AtF, c12): the LN-folded
        operand (same view geometry as At) and its row constants, used by the bf16x3 kernel when ``ln`` is given;
        ``split``: the pre-split fragment pack of the operand that is multiplied (At, or AtF with ``ln``).
        ``ln_compute``: the (mu, rs) of ``ln`` are OUTPUTS made by the kernel; returns False (nothing launched) when the
        producer/consumer kernel does not take the shape — the caller then runs ln_stats and calls again without it."""
        Zo, Zi, Kb, N = Bm.shape
        assert Kb == K and C.shape[2] == M and At.stride(3) == 1 and Bm.stride(3) == 1 and C.stride(3) == 1
        r = (None, 0, 0, 0)
        if R is not None:
            assert R.stride(3) == 1 and tuple(R.shape) == tuple(C.shape)
            r = (R.data_ptr(), R.stride(2), R.stride(0), R.stride(1))
        s = (None, 0, 0)
        if rowscale is not None:
            assert rowscale.stride(2) == 1
            s = (rowscale.data_ptr(), rowscale.stride(0), rowscale.stride(1))
        mu = rs = lw = lb = None
        sLN = 0
        if ln is not None:
            mu, rs, lw, lb = ln
            sLN = mu.stride(0)
        AtF = c12 = None
        if fold is not None and ln is not None:
            AtF, c12 = fold
            assert tuple(AtF.stride()) == tuple(At.stride()) and c12.is_contiguous()
        rc = self.L.rcot_gemm_kmajor(At.data_ptr(), At.stride(2), At.stride(0), At.stride(1), At.shape[2],
                                     Bm.data_ptr(), Bm.stride(2), Bm.stride(0), Bm.stride(1),
                                     C.data_ptr(), C.stride(2), C.stride(0), C.stride(1),
                                     r[0], r[1], r[2], r[3], s[0], s[1], s[2],
                                     _ptr(mu), _ptr(rs), sLN, 1 if ln_compute else 0, _ptr(lw), _ptr(lb), _ptr(AtF), _ptr(c12),
                                     _ptr(split), Zo, Zi, M, N, K, beta, self.ws.data_ptr(), self.ws_bytes, self.prec, self._st())
        if ln_compute and rc == _lib.EUNSUPPORTED:
            return False
        _lib.check(rc, "rcot_gemm_kmajor")
        return True

    def stats_ok(self, M: int, N: int, B: int) -> bool:
        """gemm_kmajor_stats takes the shape: exact fp32, one row tile holds every channel of its pixels (rcot_gemm_kmajor_stats)"""
        return self.prod_stats and self.prec == _lib.PREC_FP32 and M <= 96 and N % 128 == 0 and self.kmajor_worth(M, N, B)

    def gemm_kmajor_stats(self, At, Bm, C, M: int, K: int, R, stats):
        """C[z] = A[z] Bm[z] + R[z] as gemm_kmajor (no LayerNorm prologue, Zi = 1) AND stats = (mu, rs) [Zo, N]: the WithBias-LayerNorm
        statistics of C over its M rows, from the product's own epilogue (rcot_gemm_kmajor_stats): the LayerNorm that follows needs
        no rcot_ln_stats launch and no statistics pass in its projection kernel."""
        Zo, Zi, Kb, N = Bm.shape
        assert Zi == 1 and Kb == K and C.shape[2] == M and At.stride(3) == 1 and Bm.stride(3) == 1 and C.stride(3) == 1
        r = (None, 0, 0, 0)
        if R is not None:
            assert R.stride(3) == 1 and tuple(R.shape) == tuple(C.shape)
            r = (R.data_ptr(), R.stride(2), R.stride(0), R.stride(1))
        mu, rs = stats
        assert tuple(mu.shape) == (Zo, N) and tuple(rs.shape) == (Zo, N) and mu.stride(1) == 1 and rs.stride(1) == 1 and mu.stride(0) == rs.stride(0)
        _lib.check(self.L.rcot_gemm_kmajor_stats(At.data_ptr(), At.stride(2), At.stride(0), At.stride(1), At.shape[2],
                                                 Bm.data_ptr(), Bm.stride(2), Bm.stride(0), Bm.stride(1),
                                                 C.data_ptr(), C.stride(2), C.stride(0), C.stride(1), r[0], r[1], r[2], r[3],
                                                 mu.data_ptr(), rs.data_ptr(), mu.stride(0), Zo, Zi, M, N, K, self._st()), "rcot_gemm_kmajor_stats")

    def gemm_kmajor_multi(self, items) -> bool:
        """Up to three independent plain products of gemm_kmajor from ONE launch (rcot_gemm_kmajor_multi): items =
        [(At, Bm, C, M, K, R | None, rowscale | None), ...] with the shapes gemm_kmajor takes and a common pixel count.  False (nothing
        launched) when the arithmetic in use runs these products on the split-bf16 kernel: the caller launches them one by one."""
        if not self.multi_launch or self.prec in _TWO_TERM:
            return False
        arr = (_lib.KmajorDesc * len(items))()
        N = items[0][1].shape[3]
        for d, (At, Bm, Cc, M, K, R, rowscale) in zip(arr, items):
            Zo, Zi, Kb, n = Bm.shape
            assert Kb == K and n == N and Cc.shape[2] == M and At.stride(3) == 1 and Bm.stride(3) == 1 and Cc.stride(3) == 1
            d.At, d.lda, d.sAo, d.sAi, d.a_rows = At.data_ptr(), At.stride(2), At.stride(0), At.stride(1), At.shape[2]
            d.Bm, d.ldb, d.sBo, d.sBi = Bm.data_ptr(), Bm.stride(2), Bm.stride(0), Bm.stride(1)
            d.C, d.ldc, d.sCo, d.sCi = Cc.data_ptr(), Cc.stride(2), Cc.stride(0), Cc.stride(1)
            if R is not None:
                assert R.stride(3) == 1 and tuple(R.shape) == tuple(Cc.shape)
                d.R, d.ldr, d.sRo, d.sRi = R.data_ptr(), R.stride(2), R.stride(0), R.stride(1)
            if rowscale is not None:
                assert rowscale.stride(2) == 1
                d.rowscale, d.sSo, d.sSi = rowscale.data_ptr(), rowscale.stride(0), rowscale.stride(1)
            d.Zo, d.Zi, d.M, d.K = Zo, Zi, M, K
        rc = self.L.rcot_gemm_kmajor_multi(arr, len(items), N, self.prec, self._st())
        if rc == _lib.EUNSUPPORTED:
            return False
        _lib.check(rc, "rcot_gemm_kmajor_multi")
        return True

    @staticmethod
    def _bcn_z(t):
        """dense-plane [B,C,H,W] / [B,C,N] view -> [B,1,C,N] (no copy)"""
        if t.dim() == 4:
            t = t.view(t.shape[0], t.shape[1], t.shape[2] * t.shape[3])
        return t.unsqueeze(1)

    @staticmethod
    def _as_z(t, B):
        """[rows, ld] pack -> [B (broadcast), 1, rows, ld]"""
        return t.view(1, 1, *t.shape).expand(B, 1, -1, -1)

    def _split_of(self, packed):
        """the pre-split pack triple (WTs, WPs, WTfs) of ``packed`` that matches the arithmetic in use: the two-term packs for
        bf16x3, the three-term ones (fifth entry of the pack tuple) for bf16x6, None for exact fp32 / absent packs"""
        if self.prec == _lib.PREC_BF16X6:
            return packed[4] if len(packed) > 4 else None
        if self.prec in _TWO_TERM:
            return packed[3] if len(packed) > 3 else None
        return None

    # ------------------------------------------------------------------ 1x1 projections
    def conv1x1_fwd(self, W, X, Y, ln: LN = None, R=None, beta: float = 0.0, packed=None, ln_compute: bool = False, stats=None):
        """Y[b] = W @ LN?(X[b]) (+R[b]) (+beta*Y[b]);  W: [Co,Ci] view with unit inner stride.
        ``packed`` = (WT, WP[, (WTf, c12) | None[, (WTs, WPs, WTfs | None)]]) from pack_weight enables the K-major LDS-DMA
        kernels when N % 64 == 0.  ``ln_compute``: the (mu, rs) tensors of ``ln`` are not yet filled — the projection kernel
        makes them when it can (bf16x3 producer/consumer kernel, unsplit), otherwise ln_stats runs first.
        ``stats`` = (mu, rs) [B, N] (only where stats_ok() holds): the LayerNorm statistics of Y over its channels, made by the
        epilogue of this product (gemm_kmajor_stats)."""
        Co, Ci = W.shape
        B, ci, N, sX = self._bcn(X, "conv1x1_fwd X")
        _, co, _, sY = self._bcn(Y, "conv1x1_fwd Y")
        assert ci == Ci and co == Co and W.stride(1) == 1
        if stats is not None:
            assert ln is None and beta == 0.0 and packed is not None and self.stats_ok(Co, N, B)
            v = self._bcn_z
            return self.gemm_kmajor_stats(self._as_z(packed[0], B), v(X), v(Y), Co, Ci, None if R is None else v(R), stats)
        kmajor = packed is not None and self.kmajor_worth(Co, N, B)
        fold = split = None
        if kmajor:
            if ln is not None and len(packed) > 2 and packed[2] is not None:
                fold = (self._as_z(packed[2][0], B), packed[2][1])
            sp_ = self._split_of(packed)
            if sp_ is not None:
                split = sp_[0] if ln is None else (sp_[2] if fold is not None else None)
        if ln_compute:
            v = self._bcn_z
            made = False
            if self.ln_fused and kmajor:
                if self.prec == _lib.PREC_FP32:
                    # exact fp32: gemm_xx_kernel makes the statistics of its own pixel columns (ln_stats_kernel's arithmetic, bit-identical)
                    made = self.gemm_kmajor(self._as_z(packed[0], B), v(X), v(Y), Co, Ci, R=None if R is None else v(R), ln=ln, beta=beta,
                                            ln_compute=True)
                elif fold is not None and split is not None and Ci % 16 == 0:
                    made = self.gemm_kmajor(self._as_z(packed[0], B), v(X), v(Y), Co, Ci, R=None if R is None else v(R), ln=ln, beta=beta,
                                            fold=fold, split=split, ln_compute=True)
            if made:
                return
            self.ln_stats(X, ln[0], ln[1])
        if kmajor:
            v = self._bcn_z
            self.gemm_kmajor(self._as_z(packed[0], B), v(X), v(Y), Co, Ci, R=None if R is None else v(R), ln=ln,
                             beta=beta, fold=fold, split=split)
            return
        sR = 0
        if R is not None:
            _, cr, _, sR = self._bcn(R, "conv1x1_fwd R")
            assert cr == Co
        mu = rs = lw = lb = None
        if ln is not None:
            mu, rs, lw, lb = ln
        _lib.check(self.L.rcot_conv1x1_fwd(W.data_ptr(), W.stride(0), X.data_ptr(), sX, Y.data_ptr(), sY, B, Ci, Co, N,
                                           _ptr(mu), _ptr(rs), _ptr(lw), _ptr(lb), _ptr(R), sR, beta, self._st()),
                   "rcot_conv1x1_fwd")

    def conv1x1_dgrad(self, W, dY, dX, beta: float = 0.0, packed=None):
        Co, Ci = W.shape
        B, co, N, sdY = self._bcn(dY, "conv1x1_dgrad dY")
        _, ci, _, sdX = self._bcn(dX, "conv1x1_dgrad dX")
        assert ci == Ci and co == Co and W.stride(1) == 1
        if packed is not None and self.kmajor_worth(Ci, N, B):
            sp_ = self._split_of(packed)
            split = sp_[1] if sp_ is not None else None
            return self.gemm_kmajor(self._as_z(packed[1], B), self._bcn_z(dY), self._bcn_z(dX), Ci, Co, beta=beta, split=split)
        _lib.check(self.L.rcot_conv1x1_dgrad(W.data_ptr(), W.stride(0), dY.data_ptr(), sdY, dX.data_ptr(), sdX, B, Ci,
                                             Co, N, beta, self._st()), "rcot_conv1x1_dgrad")

    def conv1x1_wgrad(self, dY, X, dW, ln: LN = None, beta: float = 1.0):
        Co, Ci = dW.shape
        B, co, N, sdY = self._bcn(dY, "conv1x1_wgrad dY")
        _, ci, _, sX = self._bcn(X, "conv1x1_wgrad X")
        assert ci == Ci and co == Co and dW.stride(1) == 1
        mu = rs = lw = lb = None
        if ln is not None:
            mu, rs, lw, lb = ln
        _lib.check(self.L.rcot_conv1x1_wgrad(dY.data_ptr(), sdY, X.data_ptr(), sX, dW.data_ptr(), dW.stride(0), B, Ci,
                                             Co, N, _ptr(mu), _ptr(rs), _ptr(lw), _ptr(lb), beta, self.ws.data_ptr(),
                                             self.ws_bytes, self.prec_nt, self._st()), "rcot_conv1x1_wgrad")

    def conv1x1_wgrad_slabs(self, dY, X, dW, ln: LN = None, region=(0, 1)):
        """The weight gradient of conv1x1_wgrad left as split-K slabs in part ``region`` = (index, count) of the slab arena; returns the descriptor block_param_reduce() takes (it adds the slabs to ``dW``), or None when the shape has
        no slab kernel (the caller then uses conv1x1_wgrad).  rcot_conv1x1_wgrad_slabs."""
        Co, Ci = dW.shape
        B, co, N, sdY = self._bcn(dY, "conv1x1_wgrad_slabs dY")
        _, ci, _, sX = self._bcn(X, "conv1x1_wgrad_slabs X")
        assert ci == Ci and co == Co and dW.stride(1) == 1
        mu = rs = lw = lb = None
        if ln is not None:
            mu, rs, lw, lb = ln
        idx, cnt = region
        self._gen_acquire()
        per = (self._ws_slabs.numel() // cnt) // 64 * 64
        ws = self._ws_slabs[idx * per:(idx + 1) * per]
        S, ld = C.c_int(0), C.c_int(0)
        rc = self.L.rcot_conv1x1_wgrad_slabs(dY.data_ptr(), sdY, X.data_ptr(), sX, B, Ci, Co, N, _ptr(mu), _ptr(rs), _ptr(lw),
                                             _ptr(lb), ws.data_ptr(), per * 4, self.prec_nt, C.byref(S), C.byref(ld), self._st())
        if rc == _lib.EUNSUPPORTED:
            return None
        _lib.check(rc, "rcot_conv1x1_wgrad_slabs")
        return (ws.data_ptr(), S.value, Co, Ci, ld.value, dW.data_ptr(), dW.stride(0))

    def conv1x1_dgrad_wgrad_slabs(self, W, dY, dX, X, dW, ln: LN = None, packed=None, region=(0, 1)):
        """dX = W^T dY AND the weight gradient of the same dY left as split-K slabs (the descriptor of conv1x1_wgrad_slabs), from
        ONE launch (rcot_conv1x1_dgrad_wgrad_slabs, bf16x3 only).  None (nothing launched) when the shapes / the arithmetic have
        no paired kernel: the caller runs conv1x1_dgrad and conv1x1_wgrad_slabs / conv1x1_wgrad."""
        if not self.pair_launch or self.prec != _lib.PREC_BF16X3 or packed is None or len(packed) < 4 or packed[3] is None:
            return None
        Co, Ci = W.shape
        B, co, N, sdY = self._bcn(dY, "conv1x1_dgrad_wgrad dY")
        _, ci, _, sdX = self._bcn(dX, "conv1x1_dgrad_wgrad dX")
        _, ci2, _, sX = self._bcn(X, "conv1x1_dgrad_wgrad X")
        assert ci == Ci and ci2 == Ci and co == Co and dW.stride(1) == 1 and tuple(dW.shape) == (Co, Ci)
        if not self.kmajor_worth(Ci, N, B):
            return None
        WP, WPs = packed[1], packed[3][1]
        mu = rs = lw = lb = None
        if ln is not None:
            mu, rs, lw, lb = ln
        idx, cnt = region
        self._gen_acquire()
        per = (self._ws_slabs.numel() // cnt) // 64 * 64
        ws = self._ws_slabs[idx * per:(idx + 1) * per]
        S, ld = C.c_int(0), C.c_int(0)
        rc = self.L.rcot_conv1x1_dgrad_wgrad_slabs(WP.data_ptr(), WP.stride(0), WPs.data_ptr(), dY.data_ptr(), sdY, dX.data_ptr(), sdX,
                                                   X.data_ptr(), sX, B, Ci, Co, N, _ptr(mu), _ptr(rs), _ptr(lw), _ptr(lb),
                                                   self.ws.data_ptr(), self.ws_bytes, ws.data_ptr(), per * 4, self.prec, C.byref(S),
                                                   C.byref(ld), self._st())
        if rc == _lib.EUNSUPPORTED:
            return None
        _lib.check(rc, "rcot_conv1x1_dgrad_wgrad_slabs")
        return (ws.data_ptr(), S.value, Co, Ci, ld.value, dW.data_ptr(), dW.stride(0))

    # ------------------------------------------------------------------ batched small-matrix products
    def bmm_nn(self, A, Bm, C, transA: bool = False, R=None, rowscale=None, beta: float = 0.0):
        """C[zo,zi] = op(A[zo,zi]) @ Bm[zo,zi] + rowscale[zo,zi,:,None]*R[zo,zi] + beta*C.
        A: [Zo,Zi,M,K] ([Zo,Zi,K,M] if transA); Bm: [Zo,Zi,K,N]; C/R: [Zo,Zi,M,N]; rowscale: [Zo,Zi,M]."""
        Zo, Zi, K, N = Bm.shape
        M = C.shape[2]
        assert A.stride(3) == 1 and Bm.stride(3) == 1 and C.stride(3) == 1
        assert tuple(A.shape[2:]) == ((K, M) if transA else (M, K))
        r = (None, 0, 0, 0)
        if R is not None:
            assert R.stride(3) == 1 and tuple(R.shape) == tuple(C.shape)
            r = (R.data_ptr(), R.stride(2), R.stride(0), R.stride(1))
        s = (None, 0, 0)
        if rowscale is not None:
            assert rowscale.stride(2) == 1
            s = (rowscale.data_ptr(), rowscale.stride(0), rowscale.stride(1))
        _lib.check(self.L.rcot_bmm_nn(A.data_ptr(), A.stride(2), A.stride(0), A.stride(1), int(transA),
                                      Bm.data_ptr(), Bm.stride(2), Bm.stride(0), Bm.stride(1),
                                      C.data_ptr(), C.stride(2), C.stride(0), C.stride(1),
                                      r[0], r[1], r[2], r[3], s[0], s[1], s[2],
                                      Zo, Zi, M, N, K, beta, self._st()), "rcot_bmm_nn")

    def bmm_nt(self, A, Bm, C):
        """C[zo,zi] (M x N) = A[zo,zi] (M x K) @ Bm[zo,zi]^T (N x K)."""
        Zo, Zi, M, K = A.shape
        N = Bm.shape[2]
        assert A.stride(3) == 1 and Bm.stride(3) == 1 and C.stride(3) == 1 and Bm.shape[3] == K
        _lib.check(self.L.rcot_bmm_nt(A.data_ptr(), A.stride(2), A.stride(0), A.stride(1),
                                      Bm.data_ptr(), Bm.stride(2), Bm.stride(0), Bm.stride(1),
                                      C.data_ptr(), C.stride(2), C.stride(0), C.stride(1),
                                      Zo, Zi, M, N, K, self.ws.data_ptr(), self.ws_bytes, self.prec_nt, self._st()), "rcot_bmm_nt")

    def bmm_nt_slabs(self, A, Bm):
        """bmm_nt left as split-K slabs in the workspace: (pointer, S, ld) for attn_softmax(), or None when the shape has no slab
        kernel.  The slabs live until the next split-K launch on this stream: consume them with the very next call."""
        Zo, Zi, M, K = A.shape
        N = Bm.shape[2]
        assert A.stride(3) == 1 and Bm.stride(3) == 1 and Bm.shape[3] == K
        S, ld = C.c_int(0), C.c_int(0)
        rc = self.L.rcot_bmm_nt_slabs(A.data_ptr(), A.stride(2), A.stride(0), A.stride(1), Bm.data_ptr(), Bm.stride(2),
                                      Bm.stride(0), Bm.stride(1), Zo, Zi, M, N, K, self.ws.data_ptr(), self.ws_bytes, self.prec_nt,
                                      C.byref(S), C.byref(ld), self._st())
        if rc == _lib.EUNSUPPORTED:
            return None
        _lib.check(rc, "rcot_bmm_nt_slabs")
        return (self.ws.data_ptr(), S.value, ld.value)

    # ------------------------------------------------------------------ Linear
    def linear_fwd(self, X, W, bias, Y, lrelu: float = 1.0):
        B, i = X.shape
        o = W.shape[0]
        assert X.is_contiguous() and W.is_contiguous() and Y.is_contiguous() and W.shape[1] == i
        _lib.check(self.L.rcot_linear_fwd(X.data_ptr(), W.data_ptr(), _ptr(bias), Y.data_ptr(), B, i, o, lrelu,
                                          self.ws.data_ptr(), self.ws_bytes, self._st()), "rcot_linear_fwd")

    def linear_dgrad(self, dY, W, dX):
        B, o = dY.shape
        i = W.shape[1]
        assert dY.is_contiguous() and W.is_contiguous() and dX.is_contiguous()
        _lib.check(self.L.rcot_linear_dgrad(dY.data_ptr(), W.data_ptr(), dX.data_ptr(), B, i, o, self.ws.data_ptr(),
                                            self.ws_bytes, self._st()), "rcot_linear_dgrad")

    def linear_wgrad(self, dY, X, dW, beta: float = 1.0):
        B, o = dY.shape
        i = X.shape[1]
        assert dY.is_contiguous() and X.is_contiguous() and dW.is_contiguous()
        _lib.check(self.L.rcot_linear_wgrad(dY.data_ptr(), X.data_ptr(), dW.data_ptr(), B, i, o, beta, self._st()),
                   "rcot_linear_wgrad")

    # ------------------------------------------------------------------ dense convolutions
    def conv2d_fwd(self, X, Wt, bias, Y, stride: int, pad: int, lrelu: float = 1.0, cmap: int = 0, R=None, mask=None,
                   mslope: float = 1.0):
        """``mask`` (same shape as Y): the result is stored as ``mask > 0 ? y : y * mslope`` (lrelu_bwd folded into the store)"""
        B, Ci, H, W = X.shape
        Co, _, KH, KW = Wt.shape
        assert X.is_contiguous() and Wt.is_contiguous() and Y.is_contiguous() and (R is None or R.is_contiguous())
        assert mask is None or (mask.is_contiguous() and mask.shape == Y.shape)
        _lib.check(self.L.rcot_conv2d_fwd(X.data_ptr(), Wt.data_ptr(), _ptr(bias), Y.data_ptr(), B, Ci, H, W, Co, KH,
                                          KW, stride, pad, lrelu, cmap, _ptr(R), _ptr(mask), mslope, self.ws.data_ptr(),
                                          self.ws_bytes, self._st()), "rcot_conv2d_fwd")

    def conv2d_dgrad(self, dY, Wt, dX, stride: int, pad: int, beta: float = 0.0, mask=None, mslope: float = 1.0):
        """``mask`` (same shape as dX): dX is stored as ``mask > 0 ? dx : dx * mslope``"""
        B, Ci, H, W = dX.shape
        Co, _, KH, KW = Wt.shape
        assert dY.is_contiguous() and Wt.is_contiguous() and dX.is_contiguous()
        assert mask is None or (mask.is_contiguous() and mask.shape == dX.shape)
        _lib.check(self.L.rcot_conv2d_dgrad(dY.data_ptr(), Wt.data_ptr(), dX.data_ptr(), B, Ci, H, W, Co, KH, KW,
                                            stride, pad, beta, _ptr(mask), mslope, self.ws.data_ptr(), self.ws_bytes, self._st()),
                   "rcot_conv2d_dgrad")

    def conv2d_wgrad(self, dY, X, dWt, stride: int, pad: int, beta: float = 1.0):
        B, Ci, H, W = X.shape
        Co, _, KH, KW = dWt.shape
        assert dY.is_contiguous() and X.is_contiguous() and dWt.is_contiguous()
        _lib.check(self.L.rcot_conv2d_wgrad(dY.data_ptr(), X.data_ptr(), dWt.data_ptr(), B, Ci, H, W, Co, KH, KW,
                                            stride, pad, beta, self.ws.data_ptr(), self.ws_bytes, self._st()),
                   "rcot_conv2d_wgrad")

    # ------------------------------------------------------------------ k3s1 / k4s2 convolutions as bf16x3 K-major products
    @staticmethod
    def conv_pcm_ok(Ci: int, Co: int, k: int, s: int, pad: int, H: int, W: int) -> bool:
        """shapes rcot_conv_pcm takes (the critic's nine inner convolutions): forward and data gradient"""
        if (k, s, pad) == (3, 1, 1):
            return Ci % 16 == 0 and Co % 16 == 0 and W % 4 == 0
        if (k, s, pad) == (4, 2, 1):
            return Ci % 16 == 0 and Co % 16 == 0 and W % 8 == 0 and H % 2 == 0
        return False

    def _pcm_geom(self, B: int, Ho: int, Wo: int, Cd: int, Hd: int, Wd: int, kind: str):
        """Cached per geometry: N (GEMM columns), plane pitch, guard, and the colmap for a dense [B, Cd, Hd, Wd] result.
        kind 'same': result pixel (y, x) = plane position (y + 1, x + 4) (forward, k3 data gradient); 'planes': identity map."""
        key = (B, Ho, Wo, Cd, Hd, Wd, kind)
        g = self._pcm_get(key)
        if g is None:
            import numpy as np
            Wp, Hp = Wo + 8, Ho + 2
            Ps = Hp * Wp
            N = (B * Ps + 255) // 256 * 256
            G = (Wp + 8 + 3) // 4 * 4
            if kind == "planes":
                cm = np.arange(N // 4, dtype=np.int64) * 4
            else:
                n = np.arange(N // 4, dtype=np.int64) * 4
                b, r = n // Ps, n % Ps
                yp, xp = r // Wp, r % Wp
                ok = (b < B) & (yp >= 1) & (yp <= Hd) & (xp >= 4) & (xp < 4 + Wd)
                cm = np.where(ok, b * Cd * Hd * Wd + (yp - 1) * Wd + (xp - 4), -1)
            g = dict(N=N, Wp=Wp, Hp=Hp, Ps=Ps, G=G, colmap=torch.from_numpy(cm.astype(np.int32)).to(self.device), plane=(B, Ho, Wo))
            self._pcm_put(key, g)
        return g

    def _pcm_get(self, key):
        """cache lookup that refreshes the entry's age and notes it for the launch plan being recorded"""
        v = self._pcm_cache.pop(key, None)
        if v is not None:
            self._pcm_cache[key] = v                       # dict order = least recently used first
            if self.pcm_pinning:
                self._pcm_touched.add(key)
        return v

    def _pcm_put(self, key, v):
        if self._plan is not None:
            # a lazily created operand inside a recording would live in the plan's pool and its H2D copies / fills would not be
            # part of the recording: the eager warm-up pass of the same shape and arithmetic must have created it
            raise RuntimeError(f"padded-plane cache entry {key!r} created while a launch plan is being recorded "
                               "(the warm-up pass did not cover this configuration)")
        self._pcm_cache[key] = v
        if self.pcm_pinning:
            self._pcm_touched.add(key)

    def _pcm_trim(self):
        """Geometries, index tables and zeroed operand buffers are cached per shape; a validation folder with many image sizes
        must not grow that without bound.  Called only where no padded operand is pending (start of a forward / weight gradient).
        Evicts the least recently used entries, never one a live launch plan refers to (its address is baked into the recorded
        arguments: freeing it would let the allocator hand the memory to other tensors and later replays would write into them)."""
        if len(self._pcm_cache) <= 192 or self._plan is not None:
            return                                 # (never while a launch plan is being recorded: what the recording has touched so far
                                                   #  is pinned only when it succeeds — ADVICE r5)
        for key in [k for k in self._pcm_cache if k not in self._pcm_pinned and k not in self._pcm_touched]:
            if len(self._pcm_cache) <= 128:
                break
            del self._pcm_cache[key]

    def _pcm_buffer(self, rows: int, g, tag: str = ""):
        """zero-initialised [rows][N + 2 G] operand buffer of ONE plane geometry (B, Ho, Wo): rcot_conv_pcm_prep rewrites only the
        B plane images, so the guards and the tail [B*Ps, N) stay zero only as long as every user of a buffer shares the
        geometry — two geometries may round to the same N (a training level and a tall validation plane)."""
        key = ("buf" + tag, rows) + tuple(g["plane"])
        b = self._pcm_get(key)
        if b is None:
            b = torch.zeros(rows * (g["N"] + 2 * g["G"]) + 64, dtype=torch.float32, device=self.device)
            self._pcm_put(key, b)
        return b

    def conv_pcm_tables(self, Co: int, Ci: int, k: int, kind: str):
        """(rowoff, koff, M, K): A[m][kk] = W.flat[rowoff[m] + koff[kk]] for the operand orders of csrc/conv_pcm.hip.
        kind 'fwd': M = Co, kk = (tap, ci) [k4: (a, b, ij, ci)]; 'dgrad': M = Ci [k4: (ij, ci)], kk = (tap, co) [k4: (a, b, co)]."""
        key = ("tab", Co, Ci, k, kind)
        t = self._pcm_get(key)
        if t is None:
            import numpy as np
            T = k * k
            co, ci = np.arange(Co), np.arange(Ci)
            if k == 3 and kind == "fwd":
                rowoff = co * Ci * T
                koff = (np.arange(T)[:, None] + ci[None, :] * T).reshape(-1)                       # (tap, ci) -> ci*9 + tap
            elif k == 3:
                rowoff = ci * T
                koff = ((T - 1 - np.arange(T))[:, None] + co[None, :] * Ci * T).reshape(-1)        # (tap', co): flipped tap
            elif kind == "fwd":                                                                    # k4 s2: tap (a, b), parity (i, j)
                rowoff = co * Ci * T
                ab = np.array([[a, b] for a in range(2) for b in range(2)])
                ij = np.array([[i, j] for i in range(2) for j in range(2)])
                kk = (2 * ab[:, None, 0] + ij[None, :, 0]) * 4 + (2 * ab[:, None, 1] + ij[None, :, 1])     # [ab][ij] -> ky*4 + kx
                koff = (kk[:, :, None] + ci[None, None, :] * T).reshape(-1)
            else:
                ij = np.array([[i, j] for i in range(2) for j in range(2)])
                rowoff = ((ij[:, 0] * 4 + ij[:, 1])[:, None] + ci[None, :] * T).reshape(-1)       # (ij, ci)
                ab = np.array([[a, b] for a in range(2) for b in range(2)])
                koff = ((2 * ab[:, 0] * 4 + 2 * ab[:, 1])[:, None] + co[None, :] * Ci * T).reshape(-1)     # ((a, b), co)
            dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(self.device)
            t = (dev(rowoff), dev(koff), int(rowoff.size), int(koff.size))
            self._pcm_put(key, t)
        return t

    def conv_pcm_pack(self, Wt, kind: str, out=None):
        """pre-split pack of a conv weight [Co, Ci, k, k] in the operand order ``kind`` ('fwd' | 'dgrad')"""
        Co, Ci, k, _ = Wt.shape
        rowoff, koff, M, K = self.conv_pcm_tables(Co, Ci, k, kind)
        n = (K // 16) * ((M + 31) // 32) * 512
        if out is None:
            out = torch.empty(n, dtype=torch.float32, device=self.device)
        assert Wt.is_contiguous() and out.numel() == n
        _lib.check(self.L.rcot_conv_pcm_pack(Wt.data_ptr(), rowoff.data_ptr(), koff.data_ptr(), M, K, out.data_ptr(), self._st()),
                   "rcot_conv_pcm_pack")
        return out

    def _pcm_run(self, pack, M, K, buf, g, taps, bias, lrelu, cmap_g, Y, ldy):
        ldb = g["N"] + 2 * g["G"]
        arr = (C.c_int * len(taps))(*taps)
        _lib.check(self.L.rcot_conv_pcm(pack.data_ptr(), M, K, buf.data_ptr() + 4 * g["G"], ldb, g["N"], arr, len(taps), _ptr(bias),
                                        lrelu, cmap_g["colmap"].data_ptr(), Y.data_ptr(), ldy, self.ws.data_ptr(), self.ws_bytes,
                                        self._st()), "rcot_conv_pcm")

    def conv_pcm_fwd(self, X, pack, bias4, Y, k: int, lrelu: float = 1.0):
        """Y = lrelu(conv(X) + bias): k = 3 (s1 p1) or 4 (s2 p1); ``pack`` = conv_pcm_pack(W, 'fwd'); bias4: bias padded to a
        multiple of 4 entries (or None)."""
        B, Ci, H, W = X.shape
        Co, Ho, Wo = Y.shape[1], Y.shape[2], Y.shape[3]
        assert X.is_contiguous() and Y.is_contiguous()
        self._pcm_trim()
        g = self._pcm_geom(B, Ho, Wo, Co, Ho, Wo, "same")
        rows = Ci if k == 3 else 4 * Ci
        buf = self._pcm_buffer(rows, g)
        ldb = g["N"] + 2 * g["G"]
        _lib.check(self.L.rcot_conv_pcm_prep(X.data_ptr(), buf.data_ptr() + 4 * g["G"], ldb, B, Ci, H, W, 0 if k == 3 else 1, self._st()),
                   "rcot_conv_pcm_prep")
        Wp = g["Wp"]
        taps = [(ky - 1) * Wp + (kx - 1) for ky in range(3) for kx in range(3)] if k == 3 else [a * Wp + b for a in range(2) for b in range(2)]
        self._pcm_run(pack, Co, rows * len(taps), buf, g, taps, bias4, lrelu, g, Y, Ho * Wo)

    def conv_pcm_wgrad(self, dZ, X, dW, beta: float = 1.0):
        """dW = beta dW + 3x3 (s1 p1) weight gradient of dZ [B, Co, H, W] and X [B, Ci, H, W] as ONE pixel-reduction product over
        padded planes (rcot_conv_pcm_wgrad).  Leaves the padded copy of dZ in the buffer conv_pcm_dgrad(..., prepped=True)
        reads.  False (nothing launched) when the kernel does not take the shape."""
        B, Co, H, W = dZ.shape
        Ci = X.shape[1]
        assert dZ.is_contiguous() and X.is_contiguous() and dW.is_contiguous() and tuple(dW.shape) == (Co, Ci, 3, 3)
        self._pcm_trim()
        g = self._pcm_geom(B, H, W, Ci, H, W, "same")
        bz, bx = self._pcm_buffer(Co, g), self._pcm_buffer(Ci, g, "x")
        ldb = g["N"] + 2 * g["G"]
        _lib.check(self.L.rcot_conv_pcm_prep(dZ.data_ptr(), bz.data_ptr() + 4 * g["G"], ldb, B, Co, H, W, 0, self._st()), "rcot_conv_pcm_prep")
        _lib.check(self.L.rcot_conv_pcm_prep(X.data_ptr(), bx.data_ptr() + 4 * g["G"], ldb, B, Ci, H, W, 0, self._st()), "rcot_conv_pcm_prep")
        rc = self.L.rcot_conv_pcm_wgrad(bz.data_ptr() + 4 * g["G"], bx.data_ptr() + 4 * g["G"], ldb, g["N"], g["Wp"], Co, Ci, dW.data_ptr(),
                                        beta, self.ws.data_ptr(), self.ws_bytes, self.prec_nt, self._st())
        if rc == _lib.EUNSUPPORTED:
            return False
        _lib.check(rc, "rcot_conv_pcm_wgrad")
        return True

    def conv_pcm_dgrad(self, dZ, pack, dX, k: int, prepped: bool = False):
        """dX = conv data gradient of dZ [B, Co, Ho, Wo]; ``pack`` = conv_pcm_pack(W, 'dgrad').  ``prepped`` (k = 3): the padded
        copy of dZ is already in its buffer (conv_pcm_wgrad of the same dZ ran just before)."""
        B, Co, Ho, Wo = dZ.shape
        Ci, H, W = dX.shape[1], dX.shape[2], dX.shape[3]
        assert dZ.is_contiguous() and dX.is_contiguous()
        if k == 3:
            g = self._pcm_geom(B, Ho, Wo, Ci, H, W, "same")
            buf = self._pcm_buffer(Co, g)
            ldb = g["N"] + 2 * g["G"]
            if not prepped:
                _lib.check(self.L.rcot_conv_pcm_prep(dZ.data_ptr(), buf.data_ptr() + 4 * g["G"], ldb, B, Co, Ho, Wo, 0, self._st()), "rcot_conv_pcm_prep")
            Wp = g["Wp"]
            taps = [-(ky - 1) * Wp - (kx - 1) for ky in range(3) for kx in range(3)]
            # the pack's tap order is flipped (t' = 8 - t reads W[..][8 - t']): tap slot t' carries the offset of kernel tap 8 - t'
            taps = taps[::-1]
            self._pcm_run(pack, Ci, Co * 9, buf, g, taps, None, 1.0, g, dX, H * W)
            return
        g = self._pcm_geom(B, Ho, Wo, Co, Ho, Wo, "same")              # geometry of dZ; results as parity planes in the same index space
        gp = self._pcm_geom(B, Ho, Wo, 0, 0, 0, "planes")
        buf = self._pcm_buffer(Co, g)
        ldb = g["N"] + 2 * g["G"]
        _lib.check(self.L.rcot_conv_pcm_prep(dZ.data_ptr(), buf.data_ptr() + 4 * g["G"], ldb, B, Co, Ho, Wo, 0, self._st()), "rcot_conv_pcm_prep")
        Wp = g["Wp"]
        taps = [-a * Wp - b for a in range(2) for b in range(2)]
        planes = self.empty(4 * Ci, g["N"])
        self._pcm_run(pack, 4 * Ci, 4 * Co, buf, g, taps, None, 1.0, gp, planes, g["N"])
        _lib.check(self.L.rcot_conv_pcm_merge(planes.data_ptr(), g["N"], dX.data_ptr(), B, Ci, H, W, self._st()), "rcot_conv_pcm_merge")

    def pixel_shuffle(self, inp, out, mode: int):
        """mode 1: PixelUnshuffle(2), mode 2: PixelShuffle(2); inp [B,C,H,W] contiguous."""
        B, Cc, H, W = inp.shape
        assert inp.is_contiguous() and out.is_contiguous()
        _lib.check(self.L.rcot_pixel_shuffle(inp.data_ptr(), out.data_ptr(), B * Cc, H, W, mode, self._st()),
                   "rcot_pixel_shuffle")

    # ------------------------------------------------------------------ LayerNorm
    def ln_stats(self, x, mu, rs):
        B, Cc, N, sx = self._bcn(x, "ln_stats x")
        assert sx == Cc * N
        _lib.check(self.L.rcot_ln_stats(x.data_ptr(), mu.data_ptr(), rs.data_ptr(), B, Cc, N, self._st()), "rcot_ln_stats")

    def ln_bwd(self, g, x, mu, rs, w, dres, dx, dw, db, slot=None):
        """``slot`` (0/1): leave the dw/db partial rows in the backend's LN scratch ``slot`` for block_param_reduce()."""
        B, Cc, N, sx = self._bcn(x, "ln_bwd x")
        assert sx == Cc * N and g.is_contiguous() and dx.is_contiguous() and (dres is None or dres.is_contiguous())
        if slot is None:
            ws, nb, dwp, dbp = self.ws, self.ws_bytes, dw.data_ptr(), db.data_ptr()
        else:
            gen = self._gen
            self._gen_acquire()
            sc = self._ln_scratch[gen][slot]
            ws, nb, dwp, dbp = sc, sc.numel() * 4, None, None
            self._ln_rows = int(self.L.rcot_ln_bwd_rows(B, Cc, N))
        _lib.check(self.L.rcot_ln_bwd(g.data_ptr(), x.data_ptr(), mu.data_ptr(), rs.data_ptr(), w.data_ptr(), _ptr(dres),
                                      dx.data_ptr(), dwp, dbp, B, Cc, N, ws.data_ptr(), nb, self._st()), "rcot_ln_bwd")

    @property
    def _ws_slabs(self):
        """the slab arena of the generation the block being swept writes"""
        return self._ws_slabs_gen[self._gen]

    def _gen_acquire(self):
        """before the calling stream writes the current generation of per-block scratch (LayerNorm partial rows, weight-gradient
        slabs): the deferred reduce that read this generation (two blocks ago, side stream) must have finished; what it held may be
        released — later allocations on this stream are ordered behind the wait"""
        gen = self._gen
        if self._gen_event[gen] is not None:
            if self._side is not None and torch.cuda.current_stream() == self._side:
                return                             # a writer ON the side stream is already ordered behind that reduce
            ev = self._gen_event[gen]
            if isinstance(ev, _Handover):
                dev = self._dev_index
                self._host(lambda: ev.wait(_raw_stream(dev)))
            else:
                self._host(lambda: torch.cuda.current_stream().wait_event(ev))
            self._gen_event[gen] = None
            self._held_gen[gen].clear()

    def block_param_reduce(self, C, gw1, gb1, gw2, gb2, dWo_part, gWo, dtemp_part, gtemp, slabs=(), close_block=False):
        """LN partials of scratch slot 0 -> (gw1, gb1), slot 1 -> (gw2, gb2); gWo += dWo_part.sum(0); gtemp += dtemp_part.sum(0);
        every descriptor of ``slabs`` (conv1x1_wgrad_slabs, at most 4): its weight gradient += the sum of its slabs.
        ``close_block``: this is the last launch of a transformer block's backward.  With the side stream on it is then enqueued
        THERE, behind the block's weight-gradient kernels (it only writes parameter gradients), and the calling stream goes on
        with the next block without waiting (side_join() before anything reads those gradients); otherwise the calling stream
        first waits for the side stream."""
        B, heads = dtemp_part.shape
        slabs = [d for d in slabs if d is not None]
        rows = (ctypes_ll * (7 * len(slabs)))(*[v for d in slabs for v in d]) if slabs else None
        g = self._gen
        sc0, sc1, ln_rows = self._ln_scratch[g][0], self._ln_scratch[g][1], self._ln_rows

        def launch():
            _lib.check(self.L.rcot_block_param_reduce(sc0.data_ptr(), sc1.data_ptr(), ln_rows, C,
                                                      gw1.data_ptr(), gb1.data_ptr(), gw2.data_ptr(), gb2.data_ptr(),
                                                      dWo_part.data_ptr(), gWo.data_ptr(), dtemp_part.data_ptr(), gtemp.data_ptr(), B,
                                                      heads, rows, len(slabs), self._st()), "rcot_block_param_reduce")
        if close_block and self.overlap and self.defer_close:
            self._handover_to_side()
            with torch.cuda.stream(self._side):
                launch()
            if self._raw_events:
                ev, side = self._new_handover(), self._side.cuda_stream
                self._host(lambda: ev.record(side))
            else:
                ev = torch.cuda.Event()
                self._host(lambda: ev.record(self._side))
            self._gen_event[g] = ev
            self._held_gen[g] = self._held + [dWo_part, dtemp_part]
            self._held = []
            self._side_pending = True
            self._gen ^= 1
            return
        if close_block:
            self.side_join()
        launch()

    # ------------------------------------------------------------------ depthwise stencils
    def dwconv3x3(self, x, w, y, flip: bool = False):
        B, Cc, H, W = x.shape
        assert x.is_contiguous() and y.is_contiguous() and w.is_contiguous()
        _lib.check(self.L.rcot_dwconv3x3(x.data_ptr(), w.data_ptr(), y.data_ptr(), B, Cc, H, W, int(flip), self._st()),
                   "rcot_dwconv3x3")

    def gdfn_gate_fwd(self, p, w, g):
        B, c2, H, W = p.shape
        assert p.is_contiguous() and g.is_contiguous() and w.is_contiguous()
        _lib.check(self.L.rcot_gdfn_gate_fwd(p.data_ptr(), w.data_ptr(), g.data_ptr(), B, c2 // 2, H, W, self._st()),
                   "rcot_gdfn_gate_fwd")

    def gdfn_gate_bwd(self, p, w, dg, dd, dw=None):
        """dd from dg; with ``dw`` the depthwise weight gradient is accumulated in the same pass."""
        B, c2, H, W = p.shape
        assert p.is_contiguous() and dg.is_contiguous() and dd.is_contiguous() and (dw is None or dw.is_contiguous())
        _lib.check(self.L.rcot_gdfn_gate_bwd(p.data_ptr(), w.data_ptr(), dg.data_ptr(), dd.data_ptr(), _ptr(dw), B, c2 // 2,
                                             H, W, self._st()), "rcot_gdfn_gate_bwd")

    @staticmethod
    def gdfn_bwd_needs_scratch(H: int, W: int) -> bool:
        """True when rcot_gdfn_bwd cannot take its fused route for this plane size (mirrors the C dispatcher)."""
        wq = W // 4
        if wq > 64 or 64 % wq:
            return True
        for rs in (16, 8, 4):
            tpp = -(-H // rs) * wq
            if not (tpp % 256 == 0 or tpp % 64 == 0 or (tpp < 64 and tpp & (tpp - 1) == 0)):
                return True
        return False

    def gdfn_bwd(self, p, w, dg, dp, dw):
        """dp = dwconv3x3(gate backward of dg, w, flip) and dw += depthwise weight gradient, one pass (dd stays on chip)."""
        B, c2, H, W = p.shape
        assert p.is_contiguous() and dg.is_contiguous() and dp.is_contiguous() and dw.is_contiguous() and w.is_contiguous()
        scratch = self.empty(B, c2, H, W) if self.gdfn_bwd_needs_scratch(H, W) else None
        _lib.check(self.L.rcot_gdfn_bwd(p.data_ptr(), w.data_ptr(), dg.data_ptr(), dp.data_ptr(), dw.data_ptr(), _ptr(scratch), B,
                                        c2 // 2, H, W, self._st()), "rcot_gdfn_bwd")

    def dwconv3x3_bwd(self, dy, x, w, dx, dw):
        """dx = dwconv3x3(dy, w, flip) and dw += wgrad(dy, x) in one pass over dy."""
        B, Cc, H, W = x.shape
        assert dy.is_contiguous() and x.is_contiguous() and dx.is_contiguous() and dw.is_contiguous() and w.is_contiguous()
        _lib.check(self.L.rcot_dwconv3x3_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), dx.data_ptr(), dw.data_ptr(), B, Cc, H, W,
                                             self._st()), "rcot_dwconv3x3_bwd")

    def dwconv3x3_wgrad(self, dy, x, dw):
        B, Cc, H, W = x.shape
        assert dy.is_contiguous() and x.is_contiguous() and dw.is_contiguous()
        _lib.check(self.L.rcot_dwconv3x3_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), B, Cc, H, W, self._st()),
                   "rcot_dwconv3x3_wgrad")

    # ------------------------------------------------------------------ MDTA small-matrix core
    def row_sumsq(self, x, out):
        """x: [B,R,N] view (dense rows); out: [B,R] contiguous."""
        B, R, N, sx = self._bcn(x, "row_sumsq x")
        assert out.is_contiguous()
        _lib.check(self.L.rcot_row_sumsq(x.data_ptr(), out.data_ptr(), B, R, N, sx, self._st()), "rcot_row_sumsq")

    def attn_softmax(self, Graw, sq, temp, Gn, A):
        """``Graw``: the Gram tensor [B, heads, c, c], or the slab descriptor of bmm_nt_slabs()."""
        B, heads, c, _ = Gn.shape
        for t in (sq, temp, Gn, A):
            assert t.is_contiguous()
        if isinstance(Graw, tuple):
            gp, S, ld = Graw
        else:
            assert Graw.is_contiguous() and tuple(Graw.shape) == tuple(Gn.shape)
            gp, S, ld = Graw.data_ptr(), 0, c
        _lib.check(self.L.rcot_attn_softmax(gp, S, ld, sq.data_ptr(), temp.data_ptr(), Gn.data_ptr(), A.data_ptr(),
                                            B, heads, c, self._st()), "rcot_attn_softmax")

    def attn_core_fwd(self, u, temp, WoT, sq, Gn, A, MfT) -> bool:
        """sq, Gn, A and the folded operand MfT = (W_o blockdiag(A))^T straight from u = [q | k | v] (rcot_attn_core_fwd: the
        64x64 / 32x32 / 16x16 levels).  False when the shape has no such kernel: the caller runs the four separate launches."""
        B, heads, c, _ = Gn.shape
        N = u.shape[2] * u.shape[3]
        if not self.attn_core or N > self.attn_core_maxn:
            return False
        for t in (u, temp, sq, Gn, A, MfT):
            assert t.is_contiguous()
        assert WoT.stride(1) == 1 and tuple(MfT.shape) == (B, heads * c, heads * c) and u.shape[1] == 3 * heads * c
        rc = self.L.rcot_attn_core_fwd(u.data_ptr(), u.stride(0), temp.data_ptr(), WoT.data_ptr(), WoT.stride(0), sq.data_ptr(),
                                       Gn.data_ptr(), A.data_ptr(), MfT.data_ptr(), MfT.stride(1), MfT.stride(0), B, heads, c, N,
                                       self.ws.data_ptr(), self.ws_bytes, self._st())
        if rc == _lib.EUNSUPPORTED:
            return False
        _lib.check(rc, "rcot_attn_core_fwd")
        return True

    def attn_bwd_small(self, dA, A, Gn, sq, temp, dtemp_part, Eq, EqT, Dq, Dk):
        B, heads, c, _ = A.shape
        for t in (dA, A, Gn, sq, temp, dtemp_part, Eq, EqT, Dq, Dk):
            assert t.is_contiguous()
        _lib.check(self.L.rcot_attn_bwd_small(dA.data_ptr(), A.data_ptr(), Gn.data_ptr(), sq.data_ptr(), temp.data_ptr(),
                                              dtemp_part.data_ptr(), Eq.data_ptr(), EqT.data_ptr(), Dq.data_ptr(), Dk.data_ptr(), B, heads,
                                              c, self._st()), "rcot_attn_bwd_small")

    def attn_bwd_fused(self, dM, Wo, A, Gn, sq, temp, Mf, dWo_part, dtemp_part, Eq, EqT, Dq, Dk):
        """Mf = W_o blockdiag(A), per-image dW_o, and the outputs of attn_bwd_small: two launches instead of four."""
        B, heads, c, _ = A.shape
        for t in (dM, Wo, A, Gn, sq, temp, Mf, dWo_part, dtemp_part, Eq, EqT, Dq, Dk):
            assert t.is_contiguous()
        _lib.check(self.L.rcot_attn_bwd_fused(dM.data_ptr(), Wo.data_ptr(), A.data_ptr(), Gn.data_ptr(), sq.data_ptr(),
                                              temp.data_ptr(), Mf.data_ptr(), dWo_part.data_ptr(), dtemp_part.data_ptr(),
                                              Eq.data_ptr(), EqT.data_ptr(), Dq.data_ptr(), Dk.data_ptr(), B, heads, c,
                                              self.ws.data_ptr(), self.ws_bytes, self._st()), "rcot_attn_bwd_fused")

    def attn_core_bwd(self, dM, Wo, A, Gn, sq, temp, Mf, dWo_part, dtemp_part, Eq, EqT, Dq, Dk) -> bool:
        """Everything attn_bwd_fused returns, in ONE launch (rcot_attn_core_bwd).  ``dM``: the dense [B, C, C] tensor or the slab
        descriptor (pointer, S <= 8, ld) of bmm_nt_slabs.  False when there is no such kernel for the head width."""
        B, heads, c, _ = A.shape
        if not self.attn_core or c > 48:
            # c = 96 (decoder_level1 / refinement / noise_level3): measured equal or slower than the two-launch form (57.8 vs 57 us
            # at the 128x128 level, 93 vs 84 us at 16x16): one workgroup per (head, image) walks 4x the MFMA work alone
            return False
        for t in (Wo, A, Gn, sq, temp, Mf, dWo_part, dtemp_part, Eq, EqT, Dq, Dk):
            assert t.is_contiguous()
        if isinstance(dM, tuple):
            dp, S, ld = dM
            if S > 8:
                return False
        else:
            assert dM.is_contiguous()
            dp, S, ld = dM.data_ptr(), 0, heads * c
        rc = self.L.rcot_attn_core_bwd(dp, S, ld, Wo.data_ptr(), A.data_ptr(), Gn.data_ptr(), sq.data_ptr(), temp.data_ptr(),
                                       Mf.data_ptr(), dWo_part.data_ptr(), dtemp_part.data_ptr(), Eq.data_ptr(), EqT.data_ptr(),
                                       Dq.data_ptr(), Dk.data_ptr(), B, heads, c, self._st())
        if rc == _lib.EUNSUPPORTED:
            return False
        _lib.check(rc, "rcot_attn_core_bwd")
        return True

    @staticmethod
    def attn_fused_ok(c: int) -> bool:
        return c in (48, 96)

    def batch_reduce(self, src, dst, beta: float = 1.0):
        """dst = beta*dst + src.sum(0); src: [B, ...] contiguous."""
        B = src.shape[0]
        assert src.is_contiguous() and dst.is_contiguous() and src.numel() == B * dst.numel()
        _lib.check(self.L.rcot_batch_reduce(src.data_ptr(), dst.data_ptr(), B, dst.numel(), beta, self._st()),
                   "rcot_batch_reduce")

    # ------------------------------------------------------------------ elementwise / critic pieces
    def lrelu_bwd(self, dy, a, dz, slope: float = 0.2):
        assert dy.is_contiguous() and a.is_contiguous() and dz.is_contiguous()
        _lib.check(self.L.rcot_lrelu_bwd(dy.data_ptr(), a.data_ptr(), dz.data_ptr(), dy.numel(), slope, self._st()),
                   "rcot_lrelu_bwd")

    def bias_grad(self, dz, db):
        """db[c] += sum over batch and pixels of dz [B,C,...]."""
        B, Cc = dz.shape[0], dz.shape[1]
        assert dz.is_contiguous()
        _lib.check(self.L.rcot_bias_grad(dz.data_ptr(), db.data_ptr(), B, Cc, dz.numel() // (B * Cc), self._st()),
                   "rcot_bias_grad")

    def axpby(self, x, y, out, a: float = 1.0, b: float = 1.0):
        """out = a*x + b*y on [rows, cols]-like tensors: dim 0 may be strided, the rest must be dense."""
        def rc(t):
            rows = t.shape[0]
            cols = t.numel() // rows
            assert t[0].is_contiguous()
            return rows, cols, (t.stride(0) if rows > 1 else cols)
        rows, cols, sx = rc(x)
        _, _, so = rc(out)
        sy = 0
        if y is not None:
            _, _, sy = rc(y)
            assert y.shape == x.shape
        assert out.shape == x.shape
        _lib.check(self.L.rcot_axpby2d(x.data_ptr(), sx, _ptr(y), sy, out.data_ptr(), so, rows, cols, a, b, self._st()),
                   "rcot_axpby2d")

    def fill(self, t, v: float = 0.0):
        """t[...] = v for a dense fp32 tensor / view (rcot_fill): loss seeds and gradient-buffer zeroing without a PyTorch launch"""
        assert t.is_contiguous() and t.dtype == torch.float32
        if t.numel():
            _lib.check(self.L.rcot_fill(t.data_ptr(), t.numel(), float(v), self._st()), "rcot_fill")

    # ------------------------------------------------------------------ MPRNet backbone pieces (csrc/mprnet_ops.hip; Net.py:19-176)
    def prelu_fwd(self, x, slope, y):
        """y = x > 0 ? x : slope x with the ONE device-resident slope of nn.PReLU() (Net.py:185)"""
        assert x.is_contiguous() and y.is_contiguous() and slope.numel() == 1
        _lib.check(self.L.rcot_prelu_fwd(x.data_ptr(), slope.data_ptr(), y.data_ptr(), x.numel(), self._st()), "rcot_prelu_fwd")

    def prelu_bwd(self, dy, x, slope, dx, dslope):
        """dx = x > 0 ? dy : slope dy (dx may alias dy); dslope[0] += sum_{x <= 0} x dy"""
        assert dy.is_contiguous() and x.is_contiguous() and dx.is_contiguous() and slope.numel() == 1 and dslope.numel() == 1
        _lib.check(self.L.rcot_prelu_bwd(dy.data_ptr(), x.data_ptr(), slope.data_ptr(), dx.data_ptr(), dslope.data_ptr(), x.numel(),
                                         self.ws.data_ptr(), self.ws_bytes, self._st()), "rcot_prelu_bwd")

    def row_dot(self, a, b, out, scale: float = 1.0):
        """out[b, c] = scale * sum over the plane of a * (b or 1); a, b: [B, C, H, W] dense"""
        rows = a.shape[0] * a.shape[1]
        assert a.is_contiguous() and (b is None or (b.is_contiguous() and b.shape == a.shape)) and out.is_contiguous() and out.numel() == rows
        _lib.check(self.L.rcot_row_dot(a.data_ptr(), _ptr(b), out.data_ptr(), rows, a.numel() // rows, float(scale), self._st()),
                   "rcot_row_dot")

    def row_scale_add(self, a, s, x, t, tscale: float, out):
        """out = a * s[b, c] + (x or 0) + (t[b, c] * tscale or 0); out may alias a or x"""
        rows = a.shape[0] * a.shape[1]
        assert a.is_contiguous() and out.is_contiguous() and out.shape == a.shape and s.is_contiguous() and s.numel() == rows
        assert (x is None or (x.is_contiguous() and x.shape == a.shape)) and (t is None or (t.is_contiguous() and t.numel() == rows))
        _lib.check(self.L.rcot_row_scale_add(a.data_ptr(), s.data_ptr(), _ptr(x), _ptr(t), float(tscale), out.data_ptr(), rows,
                                             a.numel() // rows, self._st()), "rcot_row_scale_add")

    def ca_gate_fwd(self, mean, W1, W2, hid, gate):
        """CALayer.conv_du on pooled means [B, C]: hid = relu(W1 mean), gate = sigmoid(W2 hid) (Net.py:42-47)"""
        B, Cc = mean.shape
        Cr = W1.shape[0]
        assert all(t.is_contiguous() for t in (mean, W1, W2, hid, gate)) and W1.numel() == Cr * Cc == W2.numel()
        _lib.check(self.L.rcot_ca_gate_fwd(mean.data_ptr(), W1.data_ptr(), W2.data_ptr(), hid.data_ptr(), gate.data_ptr(), B, Cc, Cr,
                                           self._st()), "rcot_ca_gate_fwd")

    def ca_gate_bwd(self, dgate, gate, hid, mean, W1, W2, dW1, dW2, dmean):
        B, Cc = mean.shape
        Cr = W1.shape[0]
        assert all(t.is_contiguous() for t in (dgate, gate, hid, mean, W1, W2, dW1, dW2, dmean))
        _lib.check(self.L.rcot_ca_gate_bwd(dgate.data_ptr(), gate.data_ptr(), hid.data_ptr(), mean.data_ptr(), W1.data_ptr(),
                                           W2.data_ptr(), dW1.data_ptr(), dW2.data_ptr(), dmean.data_ptr(), B, Cc, Cr, self.ws.data_ptr(),
                                           self.ws_bytes, self._st()),
                   "rcot_ca_gate_bwd")

    def bilinear_down2(self, x, y):
        """nn.Upsample(scale_factor=0.5, bilinear, align_corners=False): [B, C, H, W] -> [B, C, H/2, W/2] (Net.py:149)"""
        B, Cc, H, W = x.shape
        assert x.is_contiguous() and y.is_contiguous() and tuple(y.shape) == (B, Cc, H // 2, W // 2)
        _lib.check(self.L.rcot_bilinear_down2(x.data_ptr(), y.data_ptr(), B * Cc, H, W, self._st()), "rcot_bilinear_down2")

    def bilinear_down2_bwd(self, dy, dx, beta: float = 0.0):
        B, Cc, H, W = dx.shape
        assert dy.is_contiguous() and dx.is_contiguous() and tuple(dy.shape) == (B, Cc, H // 2, W // 2)
        _lib.check(self.L.rcot_bilinear_down2_bwd(dy.data_ptr(), dx.data_ptr(), B * Cc, H, W, float(beta), self._st()),
                   "rcot_bilinear_down2_bwd")

    def bilinear_up2(self, x, skip, y):
        """y [B, C, 2H, 2W] = nn.Upsample(scale_factor=2, bilinear, align_corners=False)(x) + (skip or 0) (Net.py:167-176)"""
        B, Cc, H, W = x.shape
        assert x.is_contiguous() and y.is_contiguous() and tuple(y.shape) == (B, Cc, 2 * H, 2 * W)
        assert skip is None or (skip.is_contiguous() and skip.shape == y.shape)
        _lib.check(self.L.rcot_bilinear_up2(x.data_ptr(), _ptr(skip), y.data_ptr(), B * Cc, H, W, self._st()), "rcot_bilinear_up2")

    def bilinear_up2_bwd(self, dy, dx):
        B, Cc, H, W = dx.shape
        assert dy.is_contiguous() and dx.is_contiguous() and tuple(dy.shape) == (B, Cc, 2 * H, 2 * W)
        _lib.check(self.L.rcot_bilinear_up2_bwd(dy.data_ptr(), dx.data_ptr(), B * Cc, H, W, self._st()), "rcot_bilinear_up2_bwd")

    def conv_weight_flip(self, src, dst, table, n: int, Co: int, Ci: int, K: int):
        """dst[.. ci, co, K-1-ky, K-1-kx] = src[.. co, ci, ky, kx] for the n weights whose element offsets the DEVICE int64 table
        lists (pairs: in src, in dst): the operand with which rcot_conv2d_fwd computes a stride-1 data gradient"""
        assert src.is_contiguous() and dst.is_contiguous() and table.dtype == torch.int64 and table.numel() == 2 * n
        _lib.check(self.L.rcot_conv_weight_flip(src.data_ptr(), dst.data_ptr(), table.data_ptr(), n, Co, Ci, K, K, self._st()),
                   "rcot_conv_weight_flip")

    def lerp(self, t, f, alpha, out):
        B = t.shape[0]
        assert t.is_contiguous() and f.is_contiguous() and out.is_contiguous() and alpha.is_contiguous()
        _lib.check(self.L.rcot_lerp(t.data_ptr(), f.data_ptr(), alpha.data_ptr(), out.data_ptr(), B, t.numel() // B,
                                    self._st()), "rcot_lerp")

    def gp_penalty(self, g, norms, u0, gp_out, inv_global_batch: float):
        B = g.shape[0]
        assert g.is_contiguous() and u0.is_contiguous()
        _lib.check(self.L.rcot_gp_penalty(g.data_ptr(), norms.data_ptr(), u0.data_ptr(), gp_out.data_ptr(), B,
                                          g.numel() // B, inv_global_batch, self._st()), "rcot_gp_penalty")

    # ------------------------------------------------------------------ OT cost
    def ot_reduce(self, degraded, restored, target, sums):
        B = degraded.shape[0]
        assert degraded.is_contiguous() and restored.is_contiguous() and (target is None or target.is_contiguous())
        _lib.check(self.L.rcot_ot_reduce(degraded.data_ptr(), restored.data_ptr(), _ptr(target), sums.data_ptr(), B,
                                         degraded.numel() // B, self._st()), "rcot_ot_reduce")

    def ot_spectrum(self, degraded, restored, de_id, gF, spec):
        B, _, H, W = degraded.shape
        assert de_id.dtype == torch.int32 and de_id.is_cuda
        _lib.check(self.L.rcot_ot_spectrum(degraded.data_ptr(), restored.data_ptr(), de_id.data_ptr(), gF.data_ptr(),
                                           spec.data_ptr(), self.ws.data_ptr(), self.ws_bytes, B, H, W, self._st()),
                   "rcot_ot_spectrum")

    def ot_grad(self, degraded, restored, target, de_id, gF, sums, spec, dout, scal, sigma, Sigma, global_batch):
        B = degraded.shape[0]
        assert de_id.dtype == torch.int32 and dout.is_contiguous()
        _lib.check(self.L.rcot_ot_grad(degraded.data_ptr(), restored.data_ptr(), _ptr(target), de_id.data_ptr(),
                                       _ptr(gF), sums.data_ptr(), spec.data_ptr(), dout.data_ptr(), scal.data_ptr(), B,
                                       degraded.numel() // B, sigma, Sigma, global_batch, self._st()), "rcot_ot_grad")

    # ------------------------------------------------------------------ data contract
    def patch_prep(self, clean_img, deg_img, y0: int, x0: int, P: int, mode: int, sigma: float, seed: int, deg_out, clean_out):
        """clean_img / deg_img: uint8 [H, W, 3] on the device (deg_img None -> synthetic noise of ``sigma``); writes the
        [3, P, P] float slots of the batch tensors (rcot_patch_prep)."""
        H, W, _ = clean_img.shape
        assert clean_img.dtype == torch.uint8 and clean_img.is_contiguous() and clean_img.is_cuda
        assert deg_img is None or (deg_img.dtype == torch.uint8 and deg_img.is_contiguous() and tuple(deg_img.shape) == (H, W, 3))
        assert deg_out.is_contiguous() and clean_out.is_contiguous() and tuple(deg_out.shape) == (3, P, P)
        _lib.check(self.L.rcot_patch_prep(_ptr(deg_img), clean_img.data_ptr(), H, W, y0, x0, P, mode, float(sigma),
                                          int(seed) & 0xFFFFFFFFFFFFFFFF, deg_out.data_ptr(), clean_out.data_ptr(), self._st()),
                   "rcot_patch_prep")

    # ------------------------------------------------------------------ optimizers
    def rmsprop_step(self, p, g, sq, n, lr, alpha=0.99, eps=1e-8, grad_scale=1.0):
        _lib.check(self.L.rcot_rmsprop_step(p.data_ptr(), g.data_ptr(), sq.data_ptr(), n, lr, alpha, eps, grad_scale,
                                            self._st()), "rcot_rmsprop_step")

    def adam_step(self, p, g, m, v, n, lr, step, b1=0.9, b2=0.999, eps=1e-8, grad_scale=1.0):
        _lib.check(self.L.rcot_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr, b1, b2, eps,
                                         step, grad_scale, self._st()), "rcot_adam_step")
