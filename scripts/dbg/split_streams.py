"""Premise test: does the transport map's forward+backward for B=8 finish sooner as TWO independent half-batch instances on two
streams?  Each instance is captured as a HIP graph (host enqueue out of the picture) and the graphs are replayed alone, back to
back, and concurrently."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rcot_amd import lib
from rcot_amd.net_restormer import T_net
from rcot_amd.ops import HipBackend

def make(B, seed):
    be = HipBackend()
    be.overlap = False
    be.prec = lib.PREC_BF16X3
    net = T_net(decoder=True, backend=be, seed=seed)
    x = torch.rand(B, 3, 128, 128, device="cuda")
    d = torch.randn(B, 3, 128, 128, device="cuda")
    def run():
        net.zero_grad()
        net.forward(x, save=True)
        net.backward(d)
    for _ in range(2): run()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        run()
    return g, st

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

g8, s8 = make(8, 1)
def r8():
    with torch.cuda.stream(s8): g8.replay()
print(f"one B=8 graph: {timed(r8):.1f} ms", flush=True)
ga, sa = make(4, 1)
gb, sb = make(4, 2)
def seq():
    with torch.cuda.stream(sa): ga.replay(); gb.replay()
def par():
    with torch.cuda.stream(sa): ga.replay()
    with torch.cuda.stream(sb): gb.replay()
print(f"two B=4 graphs back to back: {timed(seq):.1f} ms", flush=True)
print(f"two B=4 graphs on two streams: {timed(par):.1f} ms", flush=True)
