"""GEMM micro-benchmark: python tools/bench_gemm.py nt|tn M N K [iters] [epi]
   epi (nt only): plain | bias | qkv (bias+scale) | gelu (bias+gelu+aux) | res (bias+residual, fp32 out) | gbwd (gelu-bwd)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg; amd_pkg.load()
from ts_asr_whisper_amd import ops, _lib as L
kind, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
epi = sys.argv[6] if len(sys.argv) > 6 else "plain"
bf = torch.bfloat16
sa = float(os.environ.get("SCALE_A", "0.5")); sw = float(os.environ.get("SCALE_W", "0.5"))
def rnd(*s, sc=0.5): return (torch.randn(*s, device="cuda") * sc).to(bf)
if kind == "nt":
    A, W = rnd(M, K, sc=sa), rnd(N, K, sc=sw)
    C = torch.empty(M, N, dtype=torch.float32 if epi == "res" else bf, device="cuda")
    kw = {}
    if epi in ("bias", "qkv", "gelu", "res"): kw["bias"] = torch.randn(N, device="cuda")
    if epi == "qkv": kw.update(flags=L.EPI_SCALE_N, scale=0.125, scale_ncols=N // 3)
    if epi == "gelu": kw.update(flags=L.EPI_GELU, aux=torch.empty(M, N, dtype=bf, device="cuda"))
    if epi == "res": kw["residual"] = torch.randn(M, N, device="cuda")
    if epi == "gbwd": kw.update(flags=L.EPI_GELU_BWD, aux=rnd(M, N))
    fn = lambda: ops.gemm_nt(A, W, C, M, N, K, **kw)
else:   # tn: C[N,K] += A[M,N]^T B[M,K]
    A, Bm, C = rnd(M, N), rnd(M, K), torch.zeros(N, K, device="cuda")
    fn = lambda: ops.gemm_tn(A, Bm, C, M, N, K)
for _ in range(3): fn()
torch.cuda.synchronize()
if os.environ.get("INTERLEAVE"):
    # in-situ-like: a memory-bound kernel between GEMMs (cools the chip, evicts the caches); only the GEMM is timed
    x1, x2 = torch.empty(1 << 28, dtype=torch.uint8, device="cuda"), torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
    evs = []
    for _ in range(iters):
        x1.copy_(x2)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); evs.append((s, e))
    torch.cuda.synchronize()
    ms = sum(s.elapsed_time(e) for s, e in evs) / iters
else:
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
print(f"{kind} M{M} N{N} K{K} {epi} variant={os.environ.get('DICOW_NT_VARIANT','0')} splits={os.environ.get('DICOW_TN_SPLITS','auto')}"
      f"{' interleaved' if os.environ.get('INTERLEAVE') else ''}: {ms:.4f} ms  {2*M*N*K/ms/1e9:.1f} TF", flush=True)
