"""Timing only (no output comparison): the deferred-epilogue kernel variants against the ring kernel, light epilogues, in-situ.
python tools/ntd_time.py label=lib.so:DEFER ..."""
import os, subprocess, sys
CHILD = r'''
import sys, statistics, torch
sys.path.insert(0, ".")
import amd_pkg; amd_pkg.load()
from ts_asr_whisper_amd import ops, _lib as L
bf = torch.bfloat16
ea = torch.empty(1 << 28, dtype=torch.uint8, device="cuda"); eb = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
def insitu(fn, rounds=9):
    for _ in range(2): fn()
    ev = []
    for _ in range(rounds):
        ea.copy_(eb)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); ev.append((s, e))
    torch.cuda.synchronize()
    return statistics.median(s.elapsed_time(e) for s, e in ev) * 1e3
M = 24000
out = []
for (N, K) in ((1280, 1280), (3840, 1280), (1280, 5120)):
    A = (torch.randn(M, K, device="cuda") * 0.5).to(bf); W = (torch.randn(N, K, device="cuda") * 0.03).to(bf)
    bias = torch.randn(N, device="cuda") * 0.1; res = torch.randn(M, N, device="cuda")
    Cp = torch.empty(M, N, dtype=bf, device="cuda"); Cr = torch.empty(M, N, device="cuda")
    out.append(f"{insitu(lambda: ops.gemm_nt(A, W, Cp, M, N, K)):7.1f} {insitu(lambda: ops.gemm_nt(A, W, Cr, M, N, K, bias=bias, residual=res)):7.1f}")
print("   ".join(out))
'''
print(f"{'us in-situ: plain resid':30s} N1280K1280        N3840K1280        N1280K5120")
for rep in range(2):
    for a in sys.argv[1:]:
        label, rest = a.split("=", 1)
        lib, _, mode = rest.partition(":")
        env = dict(os.environ, DICOW_HIP_LIB=os.path.abspath(lib), DICOW_NT_DEFER=mode or "0")
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(f"{label:30s}", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
