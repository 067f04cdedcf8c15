"""GEMM micro-benchmark: python tools/bench_gemm.py nt|tn M N K [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg; amd_pkg.load()
from ts_asr_whisper_amd import ops
kind, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
bf = torch.bfloat16
def rnd(*s): return (torch.randn(*s, device="cuda") * 0.5).to(bf)
if kind == "nt":
    A, W, C = rnd(M, K), rnd(N, K), torch.empty(M, N, dtype=bf, device="cuda")
    fn = lambda: ops.gemm_nt(A, W, C, M, N, K)
else:   # tn: C[N,K] += A[M,N]^T B[M,K]
    A, Bm, C = rnd(M, N), rnd(M, K), torch.zeros(N, K, device="cuda")
    fn = lambda: ops.gemm_tn(A, Bm, C, M, N, K)
for _ in range(3): fn()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters): fn()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / iters
print(f"{kind} M{M} N{N} K{K} splits={os.environ.get('DICOW_TN_SPLITS','auto')}: {ms:.4f} ms  {2*M*N*K/ms/1e9:.1f} TF", flush=True)
