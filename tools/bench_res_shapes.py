"""The layer's NT GEMM shapes with their epilogues (large-v3-turbo, B=16), a 256 MB copy between launches:
   python tools/bench_res_shapes.py   (DICOW_HIP_LIB=tools/libv_<x>.so for a variant build)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg
amd_pkg.load()
from ts_asr_whisper_amd import ops, _lib as L
bf = torch.bfloat16
M, D, F = 24000, 1280, 5120
def rnd(*s): return (torch.randn(*s, device="cuda") * 0.5).to(bf)
junk = torch.empty(64 << 20, device="cuda"); junk2 = torch.empty_like(junk)
def timeit(fn, iters=12):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for _ in range(2): fn()
    for s, e in ev:
        junk2.copy_(junk)
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in ev)
    return t[len(t) // 2] * 1e3
out = []
for name, n, k, epi in [("out-proj res", D, D, "res"), ("fc2 res", D, F, "res"), ("qkv", 3 * D, D, "bias"), ("fc1 gelu", F, D, "gelu"), ("dgrad plain", D, D, "")]:
    A, W = rnd(M, k), rnd(n, k)
    bias = torch.randn(n, device="cuda")
    if epi == "res":
        C = torch.empty(M, n, device="cuda"); R = torch.randn(M, n, device="cuda")
        fn = lambda: ops.gemm_nt(A, W, C, M, n, k, bias=bias, residual=R)
    elif epi == "gelu":
        C = torch.empty(M, n, dtype=bf, device="cuda")
        fn = lambda: ops.gemm_nt(A, W, C, M, n, k, bias=bias, flags=L.EPI_GELU)
    elif epi == "bias":
        C = torch.empty(M, n, dtype=bf, device="cuda")
        fn = lambda: ops.gemm_nt(A, W, C, M, n, k, bias=bias)
    else:
        C = torch.empty(M, n, dtype=bf, device="cuda")
        fn = lambda: ops.gemm_nt(A, W, C, M, n, k)
    t = timeit(fn)
    out.append(f"{name} {t:6.1f}")
print(" | ".join(out), flush=True)
