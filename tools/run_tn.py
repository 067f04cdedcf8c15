"""The layer's four weight-gradient GEMMs, a few launches each (for rocprofv3 counter passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg; amd_pkg.load()
from ts_asr_whisper_amd import ops
M = 24000
for (N1, N2) in ((5120, 1280), (1280, 5120), (3840, 1280), (1280, 1280)):
    A = (torch.randn(M, N1, device="cuda") * 0.5).to(torch.bfloat16); B = (torch.randn(M, N2, device="cuda") * 0.5).to(torch.bfloat16)
    C = torch.zeros(N1, N2, device="cuda")
    for _ in range(4): ops.gemm_tn(A, B, C, M, N1, N2)
torch.cuda.synchronize()
