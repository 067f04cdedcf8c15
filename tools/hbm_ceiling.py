"""Practical HBM rates of this board for buffers well past the 256 MB MALL: torch copy / fill / sum of N-MB fp32 buffers (GB/s of
bytes moved).  python tools/hbm_ceiling.py"""
import torch
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for mb in (64, 128, 246, 512, 1024, 2048):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda"); a.normal_()
    tc = t(lambda: b.copy_(a)); tf = t(lambda: b.fill_(1.0)); ts = t(lambda: a.sum())
    print(f"{mb:5d} MB buffers: copy {2*mb/1024/tc/1000*1.048576:6.2f} TB/s ({tc*1e6:7.1f} us)   fill {mb/1024/tf/1000*1.048576:6.2f} TB/s   sum (read) {mb/1024/ts/1000*1.048576:6.2f} TB/s", flush=True)
