"""A/B builds of the library on the fc1 + GELU (+ saved derivative) GEMM of the encoder layer.  python tools/ab_fc1.py libA.so libB.so"""
import os, subprocess, sys
CHILD = r'''
import sys, torch
sys.path.insert(0, ".")
import amd_pkg; amd_pkg.load()
from ts_asr_whisper_amd import ops, _lib as L
bf = torch.bfloat16
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
M, N, K = 24000, 5120, 1280
A = (torch.randn(M, K, device="cuda") * 0.5).to(bf); W = (torch.randn(N, K, device="cuda") * 0.03).to(bf)
bias = torch.randn(N, device="cuda") * 0.1
C = torch.empty(M, N, dtype=bf, device="cuda"); U = torch.empty(M, N, dtype=bf, device="cuda")
t0 = timeit(lambda: ops.gemm_nt(A, W, C, M, N, K, bias=bias))
t1 = timeit(lambda: ops.gemm_nt(A, W, C, M, N, K, bias=bias, flags=L.EPI_GELU))
t2 = timeit(lambda: ops.gemm_nt(A, W, C, M, N, K, bias=bias, aux=U, flags=L.EPI_GELU | L.EPI_GELU_DAUX))
print(f"bias {t0:.0f} us  bias+gelu {t1:.0f} us  bias+gelu+daux {t2:.0f} us   checksum {float(C.float().sum()):.3f} {float(U.float().sum()):.3f}")
'''
for rep in range(3):
    for l in sys.argv[1:]:
        env = dict(os.environ, DICOW_HIP_LIB=os.path.abspath(l))
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(f"{os.path.basename(l):22s}", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
