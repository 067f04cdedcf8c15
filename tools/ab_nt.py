"""A/B builds (and DICOW_NT_VARIANT settings) of the NT GEMM at the layer's shapes, in-situ timing (a 256 MB copy between
launches), interleaved rounds of subprocess runs:  python tools/ab_nt.py label=lib.so[:variant] ..."""
import os, subprocess, sys
CHILD = r'''
import sys, os, statistics, torch
sys.path.insert(0, ".")
import amd_pkg; amd_pkg.load()
from ts_asr_whisper_amd import ops, _lib as L
bf = torch.bfloat16
M = 24000
ea = torch.empty(1 << 28, dtype=torch.uint8, device="cuda"); eb = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
def timeit(fn, rounds=9):
    for _ in range(2): fn()
    ev = []
    for _ in range(rounds):
        ea.copy_(eb)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); ev.append((s, e))
    torch.cuda.synchronize()
    return statistics.median(s.elapsed_time(e) for s, e in ev) * 1e3
out = []
for (N, K, epi) in ((3840, 1280, "plain"), (1280, 1280, "plain"), (1280, 5120, "plain"), (5120, 1280, "gelu"), (1280, 1280, "res"), (1280, 5120, "res"), (5120, 1280, "mulaux")):
    A = (torch.randn(M, K, device="cuda") * 0.5).to(bf); W = (torch.randn(N, K, device="cuda") * 0.03).to(bf)
    Cb = torch.empty(M, N, dtype=bf, device="cuda"); bias = torch.randn(N, device="cuda") * 0.1
    if epi == "plain": fn = lambda: ops.gemm_nt(A, W, Cb, M, N, K)
    elif epi == "gelu":
        aux = torch.empty(M, N, dtype=bf, device="cuda")
        fn = lambda: ops.gemm_nt(A, W, Cb, M, N, K, bias=bias, aux=aux, flags=L.EPI_GELU | L.EPI_GELU_DAUX)
    elif epi == "res":
        Cf = torch.empty(M, N, device="cuda"); res = torch.randn(M, N, device="cuda")
        fn = lambda: ops.gemm_nt(A, W, Cf, M, N, K, bias=bias, residual=res)
    else:
        aux = (torch.randn(M, N, device="cuda")).to(bf); cs = torch.zeros(N, device="cuda")
        fn = lambda: ops.gemm_nt(A, W, Cb, M, N, K, aux=aux, flags=L.EPI_MUL_AUX, colsum_out=cs)
    t = timeit(fn)
    out.append(f"{2*M*N*K/t/1e6:5.0f}")
print(" ".join(out))
'''
specs = []
for a in sys.argv[1:]:
    label, rest = a.split("=", 1)
    lib, _, var = rest.partition(":")
    specs.append((label, lib, var or "0"))
print(f"{'TF (in-situ)':26s} qkvP  outP  fc2P  fc1G  outR  fc2R  dfc2M")
for rep in range(int(os.environ.get("REPS", "2"))):
    for label, lib, var in specs:
        env = dict(os.environ, DICOW_HIP_LIB=os.path.abspath(lib), DICOW_NT_VARIANT=var)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(f"{label:26s}", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
