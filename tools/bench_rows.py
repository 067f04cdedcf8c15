"""Row-kernel (FDDT+LayerNorm) variants micro-benchmark."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg; amd_pkg.load()
from ts_asr_whisper_amd import ops
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
B, T, D = 16, 1500, 1280
M = B * T
bf = torch.bfloat16
h = torch.randn(M, D, device="cuda"); st = torch.softmax(torch.randn(B, 4, T, device="cuda"), 1)
w = [torch.randn(D, device="cuda") for _ in range(4)]; b = [torch.randn(D, device="cuda") for _ in range(4)]
ho = torch.empty_like(h); y = torch.empty(M, D, dtype=bf, device="cuda"); mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
lw, lb = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
dy = (torch.randn(M, D, device="cuda")).to(bf); gres = torch.randn(M, D, device="cuda"); g0 = torch.empty_like(h); g0b = torch.empty(M, D, dtype=bf, device="cuda")
z = lambda: torch.zeros(D, device="cuda")
dlw, dlb, cs = z(), z(), z(); dw = [z() for _ in range(4)]; db = [z() for _ in range(4)]
ops.fddt_ln_fwd(h, M, D, mode=1, stno=st, T=T, w=w, b=b, h_out=ho, ln_w=lw, ln_b=lb, y_bf16=y, mean=mean, rstd=rstd)
def rep(name, ms, byt): print(f"{name:44s} {ms*1e3:8.1f} us  {byt/ms/1e6:7.0f} GB/s", flush=True)
f32, b16 = M * D * 4, M * D * 2
rep("fwd copy (mode0, no LN)", timeit(lambda: ops.fddt_ln_fwd(h, M, D, mode=0, pos=None, h_out=ho)), 2 * f32)
rep("fwd FDDT only", timeit(lambda: ops.fddt_ln_fwd(h, M, D, mode=1, stno=st, T=T, w=w, b=b, h_out=ho)), 2 * f32)
rep("fwd LN only -> bf16", timeit(lambda: ops.fddt_ln_fwd(h, M, D, mode=0, ln_w=lw, ln_b=lb, y_bf16=y, mean=mean, rstd=rstd)), f32 + b16)
rep("fwd FDDT+LN", timeit(lambda: ops.fddt_ln_fwd(h, M, D, mode=1, stno=st, T=T, w=w, b=b, h_out=ho, ln_w=lw, ln_b=lb, y_bf16=y, mean=mean, rstd=rstd)), 2 * f32 + b16)
rep("bwd copy (mode0, no LN, g_res->g_out)", timeit(lambda: ops.fddt_ln_bwd(h, M, D, mode=0, g_res=gres, g_out=g0)), 3 * f32)
rep("bwd LN only", timeit(lambda: ops.fddt_ln_bwd(h, M, D, mode=0, ln_w=lw, mean=mean, rstd=rstd, d_y=dy, g_res=gres, g_out=g0, g_out_bf16=g0b, dln_w=dlw, dln_b=dlb, colsum_out=cs)), 3 * f32 + 2 * b16)
rep("bwd FDDT only", timeit(lambda: ops.fddt_ln_bwd(h, M, D, mode=1, stno=st, T=T, w=w, b=b, g_res=gres, g_out=g0, g_out_bf16=g0b, dw=dw, db=db, colsum_out=cs)), 3 * f32 + b16)
rep("bwd FDDT+LN (full)", timeit(lambda: ops.fddt_ln_bwd(h, M, D, mode=1, stno=st, T=T, w=w, b=b, ln_w=lw, mean=mean, rstd=rstd, d_y=dy, g_res=gres, g_out=g0, g_out_bf16=g0b, dln_w=dlw, dln_b=dlb, dw=dw, db=db, colsum_out=cs)), 3 * f32 + 2 * b16)
rep("bwd FDDT+LN no param grads", timeit(lambda: ops.fddt_ln_bwd(h, M, D, mode=1, stno=st, T=T, w=w, b=b, ln_w=lw, mean=mean, rstd=rstd, d_y=dy, g_res=gres, g_out=g0, g_out_bf16=g0b)), 3 * f32 + 2 * b16)
x = torch.empty(M, D, device="cuda")
rep("torch copy f32 (reference stream rate)", timeit(lambda: x.copy_(h)), 2 * f32)
