"""A/B builds of the TN (weight-gradient) GEMM at the layer's shapes, in-situ timing:  python tools/ab_tn.py label=lib.so ..."""
import os, subprocess, sys
CHILD = r'''
import sys, os, statistics, torch
sys.path.insert(0, ".")
import amd_pkg; amd_pkg.load()
from ts_asr_whisper_amd import ops
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
for (N1, N2) in ((5120, 1280), (1280, 5120), (3840, 1280), (1280, 1280)):
    A = (torch.randn(M, N1, device="cuda") * 0.5).to(bf); B = (torch.randn(M, N2, device="cuda") * 0.5).to(bf)
    C = torch.zeros(N1, N2, device="cuda")
    t = timeit(lambda: ops.gemm_tn(A, B, C, M, N1, N2))
    out.append(f"{2*M*N1*N2/t/1e6:5.0f}")
print(" ".join(out))
'''
specs = [a.split("=", 1) for a in sys.argv[1:]]
print(f"{'TF (in-situ, incl. split reduce)':34s} 5120x1280 1280x5120 3840x1280 1280x1280")
for rep in range(int(os.environ.get("REPS", "2"))):
    for label, lib in specs:
        env = dict(os.environ, DICOW_HIP_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(f"{label:34s}", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
