"""The encoder layer's NT GEMMs (large-v3-turbo, B = 16: M = 24000) per EPILOGUE KIND, each on the ring kernel's two tile shapes, and the
two ways to get the next layer's FDDT + LayerNorm behind fc2 (tables b and c).
   DICOW_HIP_LIB=tools/libv_ntabl.so python tools/ab_epilogues.py
(libv_ntabl.so = `tools/build_var.sh ntabl "-DDICOW_ABLATIONS" "" "" ""`; variant 0 = the shipped choice, 21 = 256 x 256 forced, 22 = 192 x 320
forced.)  A 256 MB copy runs between launches (operands arrive from HBM / MALL, not from a warm L2); median of 12."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg

amd_pkg.load()
from ts_asr_whisper_amd import ops, _lib as L

bf = torch.bfloat16
M, D, F = 24000, 1280, 5120


def rnd(*s):
    return (torch.randn(*s, device="cuda") * 0.5).to(bf)


junk = torch.empty(64 << 20, device="cuda")
junk2 = torch.empty_like(junk)


def timeit(fn, iters=12):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for _ in range(2):
        fn()
    for s, e in ev:
        junk2.copy_(junk)
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in ev)
    return t[len(t) // 2] * 1e3


cases = []
for name, n, k, epi in [("out-proj + bias + fp32 residual", D, D, "res"), ("fc2 + bias + fp32 residual", D, F, "res"),
                        ("qkv + bias + q scale", 3 * D, D, "qkv"), ("fc1 + GELU (inference)", F, D, "gelu"),
                        ("fc1 + GELU + saved gelu' (training)", F, D, "gelu_t"), ("fc2 dgrad x gelu' + column sums", F, D, "mulaux"),
                        ("plain dgrad (N = 1280)", D, D, ""), ("plain dgrad (N = 1280, K = 3840)", D, 3 * D, "")]:
    A, W = rnd(M, k), rnd(n, k)
    bias = torch.randn(n, device="cuda")
    if epi == "res":
        C = torch.empty(M, n, device="cuda"); R = torch.randn(M, n, device="cuda")
        fn = lambda A=A, W=W, C=C, R=R, n=n, k=k, bias=bias: ops.gemm_nt(A, W, C, M, n, k, bias=bias, residual=R)
    elif epi == "gelu":
        C = torch.empty(M, n, dtype=bf, device="cuda")
        fn = lambda A=A, W=W, C=C, n=n, k=k, bias=bias: ops.gemm_nt(A, W, C, M, n, k, bias=bias, flags=L.EPI_GELU)
    elif epi == "gelu_t":
        C = torch.empty(M, n, dtype=bf, device="cuda"); X = torch.empty(M, n, dtype=bf, device="cuda")
        fn = lambda A=A, W=W, C=C, X=X, n=n, k=k, bias=bias: ops.gemm_nt(A, W, C, M, n, k, bias=bias, aux=X, flags=L.EPI_GELU | L.EPI_GELU_DAUX)
    elif epi == "mulaux":
        C = torch.empty(M, n, dtype=bf, device="cuda"); X = rnd(M, n); cs = torch.zeros(n, device="cuda")
        fn = lambda A=A, W=W, C=C, X=X, cs=cs, n=n, k=k: ops.gemm_nt(A, W, C, M, n, k, aux=X, flags=L.EPI_MUL_AUX, colsum_out=cs)
    elif epi == "qkv":
        C = torch.empty(M, n, dtype=bf, device="cuda")
        fn = lambda A=A, W=W, C=C, n=n, k=k, bias=bias: ops.gemm_nt(A, W, C, M, n, k, bias=bias, flags=L.EPI_SCALE_N, scale=0.18, scale_ncols=D)
    else:
        C = torch.empty(M, n, dtype=bf, device="cuda")
        fn = lambda A=A, W=W, C=C, n=n, k=k: ops.gemm_nt(A, W, C, M, n, k)
    # the three tile choices INTERLEAVED, three rounds (the ablation build reads DICOW_NT_VARIANT per call)
    res = {v: [] for v in ("0", "21", "22")}
    for rnd_ in range(3):
        for v in res:
            os.environ["DICOW_NT_VARIANT"] = v
            res[v].append(timeit(fn))
    os.environ["DICOW_NT_VARIANT"] = "0"
    fl = 2.0 * M * n * k
    print(f"{name:40s} shipped " + " ".join(f"{t:6.1f}" for t in res["0"]) + "   256x256 " + " ".join(f"{t:6.1f}" for t in res["21"]) +
          "   192x320 " + " ".join(f"{t:6.1f}" for t in res["22"]) + f"   us   ({fl / min(res['0']) / 1e6:5.0f} TF shipped, best round)", flush=True)
    del A, W, C

if True:
    # table b: FDDT(next layer) + LayerNorm behind fc2 -- (1) separate staged row kernel on h (training form), (2) FDDT in the fc2 epilogue, the
    # LayerNorm-only wave kernel on h' (inference form; the training form would ALSO have to store h: + M * D * 4 bytes in that epilogue)
    A, W = rnd(M, F), rnd(D, F)
    bias = torch.randn(D, device="cuda")
    R = torch.randn(M, D, device="cuda")
    h = torch.empty(M, D, device="cuda"); y = torch.empty(M, D, dtype=bf, device="cuda")
    fw = [torch.randn(D, device="cuda") for _ in range(4)]; fb = [torch.randn(D, device="cuda") * 0.1 for _ in range(4)]
    rowmask = torch.rand(M, 4, device="cuda")
    g, b = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
    t_plain = timeit(lambda: ops.gemm_nt(A, W, h, M, D, F, bias=bias, residual=R))
    t_fddt = timeit(lambda: ops.gemm_nt(A, W, h, M, D, F, bias=bias, residual=R, fddt=(fw, fb, rowmask)))
    extra = torch.empty(M, D, device="cuda")
    t_copy = timeit(lambda: extra.copy_(h))                   # what a second fp32 output of the epilogue costs at the very least: its bytes
    print(f"fc2 + residual                         {t_plain:7.1f} us")
    print(f"fc2 + residual + next FDDT (epilogue)  {t_fddt:7.1f} us   (+{t_fddt - t_plain:.1f})")
    print(f"one more [M, D] fp32 stream (h beside h'): a device copy of it takes {t_copy:.1f} us (read + write; the write alone ~{t_copy / 2:.0f})")
