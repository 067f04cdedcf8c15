"""Where does gemm_ntq_kernel's result differ?  (DICOW_HIP_LIB=<ntq all-shapes build>)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg; amd_pkg.load()
from ts_asr_whisper_amd import ops
bf = torch.bfloat16
M, D = 24000, 1280
g = torch.Generator(device="cuda").manual_seed(2)
A = torch.randn(M, D, device="cuda", generator=g).to(bf)
eye = torch.eye(D, device="cuda").to(bf)
C = torch.empty(M, D, dtype=bf, device="cuda")
ops.gemm_nt(A, eye, C, M, D, D)
torch.cuda.synchronize()
print(ops.gemm_dispatch_log() if hasattr(ops, "gemm_dispatch_log") else "")
bad = (C != A)
print("bad elements", int(bad.sum()), "of", bad.numel())
if bad.any():
    r, c = bad.nonzero(as_tuple=True)
    print("rows mod 320 histogram (32-row blocks):", torch.bincount((r % 320) // 32, minlength=10).tolist())
    print("cols mod 256 histogram (32-col blocks):", torch.bincount((c % 256) // 32, minlength=8).tolist())
    print("first bad:", [(int(a), int(b), float(C[a, b]), float(A[a, b])) for a, b in list(zip(r[:8].tolist(), c[:8].tolist()))])
    print("tile rows (m // 320) with errors:", torch.unique(r // 320)[:20].tolist(), "tile cols:", torch.unique(c // 256).tolist())
