"""Achievable HBM rates of plain streaming kernels on this GPU (torch fill / copy / read-reduce), to put kernel store and
load phases into proportion."""
import torch
def tm(f, reps=10):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for mb in (64, 256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
    t_fill = tm(lambda: x.fill_(1.0))
    t_copy = tm(lambda: y.copy_(x))
    t_sum = tm(lambda: x.sum())
    print(f"{mb:5d} MB: fill {mb/1024/t_fill*1e3/1024*1024:7.0f} GB/s   copy (r+w) {2*mb/1024/t_copy*1e3:7.0f} GB/s   sum (read) {mb/1024/t_sum*1e3:7.0f} GB/s")
