"""Main-loop rate of the NT GEMM: time vs K at tile-exact sizes (one or two full rounds of 256 workgroups), beside
torch.matmul (hipBLASLt).  The slope is the k-loop cost, the intercept prologue + epilogue."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg; amd_pkg.load()
from ts_asr_whisper_amd import ops
bf = torch.bfloat16
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (M, N) in ((16384, 1024), (16384, 2048), (16384, 4096)):
    rows = []
    for K in (640, 1280, 2560, 5120, 10240):
        A = (torch.randn(M, K, device="cuda") * 0.5).to(bf); W = (torch.randn(N, K, device="cuda") * 0.5).to(bf)
        C = torch.empty(M, N, dtype=bf, device="cuda")
        t1 = timeit(lambda: ops.gemm_nt(A, W, C, M, N, K))
        t2 = timeit(lambda: torch.matmul(A, W.t(), out=C))
        rows.append((K, t1, t2))
        print(f"M{M} N{N} K{K}: ours {t1:7.1f} us {2*M*N*K/t1/1e6:6.0f} TF | torch {t2:7.1f} us {2*M*N*K/t2/1e6:6.0f} TF", flush=True)
    (k0, a0, b0), (k1, a1, b1) = rows[-2], rows[-1]
    rounds = M * N / 65536 / 256
    print(f"   slope per k-step per round: ours {(a1-a0)/((k1-k0)/64)/rounds*1e3:.0f} ns, torch {(b1-b0)/((k1-k0)/64)/rounds*1e3:.0f} ns (2048 MFMA cycles = {2048/2.0:.0f} ns @2.0 GHz)")
