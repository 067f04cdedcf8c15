"""whisper-base B=8 GEMM shapes (rows 12000 encoder / 1024 decoder), shipped dispatch: python tools/bench_base_shapes.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg
amd_pkg.load()
from ts_asr_whisper_amd import ops, _lib as L

def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us

bf = torch.bfloat16
def rnd(*s): return (torch.randn(*s, device="cuda") * 0.5).to(bf)
shapes = [("enc out-proj", 12000, 512, 512, "res"), ("enc fc2", 12000, 512, 2048, "res"), ("enc qkv dgrad", 12000, 512, 1536, ""),
          ("enc out dgrad", 12000, 512, 512, ""), ("enc fc1 dgrad", 12000, 512, 2048, ""), ("dec cross kv", 12000, 1024, 512, "bias"),
          ("dec cross kv dgrad", 12000, 512, 1024, ""), ("enc qkv", 12000, 1536, 512, "bias"), ("enc fc1", 12000, 2048, 512, "bias"),
          ("dec qkv", 1024, 1536, 512, "bias"), ("dec out", 1024, 512, 512, "res"), ("dec fc1", 1024, 2048, 512, "bias"),
          ("dec fc2", 1024, 512, 2048, "res"), ("lm head", 1024, 51904, 512, "")]
for name, m, n, k, epi in shapes:
    A, W = rnd(m, k), rnd(n, k)
    bias = torch.randn(n, device="cuda")
    if epi == "res":
        C = torch.empty(m, n, device="cuda"); R = torch.randn(m, n, device="cuda")
        fn = lambda: ops.gemm_nt(A, W, C, m, n, k, bias=bias, residual=R)
    elif epi == "bias":
        C = torch.empty(m, n, dtype=bf, device="cuda")
        fn = lambda: ops.gemm_nt(A, W, C, m, n, k, bias=bias)
    else:
        C = torch.empty(m, n, dtype=bf, device="cuda")
        fn = lambda: ops.gemm_nt(A, W, C, m, n, k)
    before = ops.gemm_dispatch_log()
    fn()
    after = ops.gemm_dispatch_log()
    kern = ",".join(k for k in after if after[k] != before.get(k, 0))
    t = timeit(fn)
    Cb = torch.empty(m, n, dtype=bf, device="cuda")
    t2 = timeit(lambda: torch.matmul(A, W.t(), out=Cb))
    print(f"{name:20s} M={m:6d} N={n:6d} K={k:5d} {epi:5s} {t:8.1f} us {2*m*n*k/t/1e6:8.1f} TF   torch(hipBLASLt, no epilogue) {t2:8.1f} us   [{kern}]", flush=True)
