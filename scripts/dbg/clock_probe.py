"""Clock and power of the part while ONE kernel runs back to back for ~4 s: the exact-fp32 forward projection 510 <- 96 at 8 x 128 x 128 on
random operands, then on zero operands; rocm-smi sampled from a second thread every 0.25 s.  (scripts/dbg/data_power_probe.py: the same kernels
take 20-30 % longer on random operands.)"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rcot_amd import lib
from rcot_amd.ops import HipBackend
be = HipBackend()
B, N, Co, Ci = 8, 16384, 510, 96
W = torch.randn(Co, Ci, device="cuda") * 0.1
st, sp = be.pack_shapes(Co, Ci)
WT, WP = torch.zeros(*st, device="cuda"), torch.zeros(*sp, device="cuda")
be.pack_weight(W, WT, WP, None)
def sample(stop, out):
    while not stop.is_set():
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        s = [l.split(":")[-1].strip() for l in r.splitlines() if ("sclk" in l or "mclk" in l or "fclk" in l or "Power (W)" in l)]
        out.append(" | ".join(s))
        time.sleep(0.25)
for kind in ("random", "zeros", "random"):
    mk = torch.randn if kind == "random" else torch.zeros
    sets = [(mk(B, Ci, N, device="cuda"), torch.empty(B, Co, N, device="cuda")) for _ in range(3)]
    g, strm = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    for X, Y in sets: be.conv1x1_fwd(W, X, Y, packed=(WT, WP, None))
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=strm):
        for i in range(300): be.conv1x1_fwd(W, sets[i % 3][0], sets[i % 3][1], packed=(WT, WP, None))
    g.replay(); torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(80): g.replay()
    e.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    print(f"== {kind} operands: {s.elapsed_time(e) / (80 * 300) * 1e3:.1f} us per launch over {s.elapsed_time(e) / 1e3:.1f} s")
    for o in out[2:10]: print("   ", o)
