"""FDDT+LN forward / backward row kernels at the bench shape, a few launches (for rocprofv3 counter passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg; amd_pkg.load()
from ts_asr_whisper_amd import ops
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
for _ in range(4):
    ops.fddt_ln_fwd(h, M, D, mode=1, stno=st, T=T, w=w, b=b, h_out=ho, ln_w=lw, ln_b=lb, y_bf16=y, mean=mean, rstd=rstd)
    ops.fddt_ln_bwd(h, M, D, mode=1, stno=st, T=T, w=w, b=b, ln_w=lw, mean=mean, rstd=rstd, d_y=dy, g_res=gres, g_out=g0, g_out_bf16=g0b, dln_w=dlw, dln_b=dlb, dw=dw, db=db, colsum_out=cs)
    ops.fddt_ln_fwd(h, M, D, mode=0, ln_w=lw, ln_b=lb, y_bf16=y, mean=mean, rstd=rstd)
    ops.fddt_ln_fwd(h, M, D, mode=1, stno=st, T=T, w=w, b=b, h_out=ho)
torch.cuda.synchronize()
