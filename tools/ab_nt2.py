"""A/B library builds on the encoder layer's NT GEMM shapes with the epilogues the step uses (large-v3-turbo, B = 16: M = 24000),
inside ONE process: the ctypes binding is pointed at each build in turn, interleaved rounds, a 256 MB copy between timed launches
(operands evicted from L2 / MALL, clock at in-step conditions).
   python tools/ab_nt2.py label=lib.so ...        (ROUNDS, default 5; GEMM_M, default 24000)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg
amd_pkg.load()
from ts_asr_whisper_amd import ops, _lib as L

specs = [a.split("=", 1) for a in sys.argv[1:]]
ROUNDS = int(os.environ.get("ROUNDS", "5"))
M = int(os.environ.get("GEMM_M", "24000"))
bf = torch.bfloat16
dev = "cuda"
torch.manual_seed(0)


def rnd(*s, sc=0.5):
    return (torch.randn(*s, device=dev) * sc).to(bf)


SHAPES = [("qkv fwd bias+qscale", 3840, 1280, "qkv"), ("out fwd bias+res", 1280, 1280, "res"), ("fc1 fwd gelu+dgelu", 5120, 1280, "gelud"),
          ("fc1 inf gelu", 5120, 1280, "gelu"), ("fc2 fwd bias+res", 1280, 5120, "res"), ("fc2 dgrad xgelu'+colsum", 5120, 1280, "mulaux"),
          ("fc1 dgrad plain", 1280, 5120, "plain"), ("out dgrad plain", 1280, 1280, "plain"), ("qkv dgrad plain", 1280, 3840, "plain")]
evict_a = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
evict_b = torch.empty(1 << 28, dtype=torch.uint8, device=dev)


def use(path):
    L.LIB_PATH = os.path.abspath(path)
    L._lib = None
    L.lib()


def make(N, K, epi):
    A, W = rnd(M, K), rnd(N, K, sc=0.03)
    bias = torch.randn(N, device=dev) * 0.1
    Cb = torch.empty(M, N, dtype=bf, device=dev)
    if epi == "qkv":
        return lambda: ops.gemm_nt(A, W, Cb, M, N, K, bias=bias, flags=L.EPI_SCALE_N, scale=0.125, scale_ncols=N // 3)
    if epi == "res":
        Cf = torch.empty(M, N, device=dev); res = torch.randn(M, N, device=dev)
        return lambda: ops.gemm_nt(A, W, Cf, M, N, K, bias=bias, residual=res)
    if epi == "gelud":
        aux = torch.empty(M, N, dtype=bf, device=dev)
        return lambda: ops.gemm_nt(A, W, Cb, M, N, K, bias=bias, aux=aux, flags=L.EPI_GELU | L.EPI_GELU_DAUX)
    if epi == "gelu":
        return lambda: ops.gemm_nt(A, W, Cb, M, N, K, bias=bias, flags=L.EPI_GELU)
    if epi == "mulaux":
        aux = rnd(M, N); cs = torch.zeros(N, device=dev)
        return lambda: ops.gemm_nt(A, W, Cb, M, N, K, aux=aux, flags=L.EPI_MUL_AUX, colsum_out=cs)
    return lambda: ops.gemm_nt(A, W, Cb, M, N, K)


def timed(fn, n=6):
    ev = []
    for _ in range(n):
        evict_a.copy_(evict_b)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        ev.append((s, e))
    torch.cuda.synchronize()
    return statistics.median(s.elapsed_time(e) for s, e in ev) * 1e3


print(f"{'shape':26s} " + " ".join(f"{k:>10s}" for k, _ in specs) + "    (us, median of interleaved rounds; TFLOP/s of the first / last build)", flush=True)
for name, N, K, epi in SHAPES:
    fn = make(N, K, epi)
    res = {k: [] for k, _ in specs}
    for r in range(ROUNDS):
        for label, path in specs:
            use(path)
            fn(); fn()
            res[label].append(timed(fn))
    med = {k: statistics.median(v) for k, v in res.items()}
    fl = 2.0 * M * N * K
    first, last = specs[0][0], specs[-1][0]
    print(f"{name:26s} " + " ".join(f"{med[k]:10.1f}" for k, _ in specs) + f"    {fl / med[first] / 1e6:6.0f} / {fl / med[last] / 1e6:6.0f}", flush=True)
