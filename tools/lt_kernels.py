"""Which hipBLASLt kernels torch.matmul picks for the encoder layer's GEMM shapes (run under rocprofv3 --kernel-trace)."""
import torch
bf = torch.bfloat16
M = 24000
for N, K in ((3840, 1280), (1280, 1280), (5120, 1280), (1280, 5120), (1280, 3840)):
    A = (torch.randn(M, K, device="cuda") * 0.5).to(bf); W = (torch.randn(N, K, device="cuda") * 0.03).to(bf)
    for _ in range(3):
        C = torch.matmul(A, W.t())
    torch.cuda.synchronize()
