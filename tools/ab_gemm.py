"""A/B two builds of libdicow_hip.so on the NT GEMM at the step's shapes (alternating subprocess runs).
   python tools/ab_gemm.py libA.so libB.so"""
import os, subprocess, sys
CHILD = r'''
import sys, os, torch
sys.path.insert(0, ".")
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
out = []
for (M, N, K) in ((24000, 3840, 1280), (24000, 1280, 1280), (24000, 5120, 1280), (24000, 1280, 5120), (16384, 2048, 5120)):
    A = (torch.randn(M, K, device="cuda") * 0.5).to(bf); W = (torch.randn(N, K, device="cuda") * 0.5).to(bf)
    C = torch.empty(M, N, dtype=bf, device="cuda")
    t = timeit(lambda: ops.gemm_nt(A, W, C, M, N, K))
    out.append(f"{2*M*N*K/t/1e6:5.0f}")
print(" ".join(out))
'''
libs = sys.argv[1:]
print("TF at (24000,3840,1280) (24000,1280,1280) (24000,5120,1280) (24000,1280,5120) (16384,2048,5120)")
for rep in range(3):
    for l in libs:
        env = dict(os.environ, DICOW_HIP_LIB=os.path.abspath(l))
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(f"{os.path.basename(l):24s}", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
