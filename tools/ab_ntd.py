"""Deferred-epilogue fc1 kernel (gemm_ntd.hip) against the ring kernel: bit-identity of the outputs and timing, one child process per
DICOW_NT_DEFER setting (the switch is read once per process).   python tools/ab_ntd.py [lib.so]"""
import os, subprocess, sys, tempfile
CHILD = r'''
import sys, os, statistics, torch
sys.path.insert(0, ".")
import amd_pkg; amd_pkg.load()
from ts_asr_whisper_amd import ops, _lib as L
bf = torch.bfloat16
out = {}
ea = torch.empty(1 << 28, dtype=torch.uint8, device="cuda"); eb = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
def hot(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
def insitu(fn, rounds=9):
    ev = []
    for _ in range(rounds):
        ea.copy_(eb)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); ev.append((s, e))
    torch.cuda.synchronize()
    return statistics.median(s.elapsed_time(e) for s, e in ev) * 1e3
for (M, N, K) in ((24000, 5120, 1280), (23900, 1600, 1024), (12100, 3520, 1344)):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(bf); W = (torch.randn(N, K, device="cuda", generator=g) * 0.03).to(bf)
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    C1 = torch.full((M, N), float("nan"), dtype=bf, device="cuda"); C2 = torch.full((M, N), float("nan"), dtype=bf, device="cuda")
    U2 = torch.full((M, N), float("nan"), dtype=bf, device="cuda")
    before = ops.gemm_dispatch_log()
    f1 = lambda: ops.gemm_nt(A, W, C1, M, N, K, bias=bias, flags=L.EPI_GELU)
    f2 = lambda: ops.gemm_nt(A, W, C2, M, N, K, bias=bias, aux=U2, flags=L.EPI_GELU | L.EPI_GELU_DAUX)
    f1(); f2()
    after = ops.gemm_dispatch_log()
    kern = ",".join(k for k in after if after[k] != before.get(k, 0))
    torch.cuda.synchronize()
    out[("gelu", M, N, K)] = (C1.cpu(), C2.cpu(), U2.cpu())
    line = f"gelu  M={M} N={N} K={K}: [{kern}]"
    if M == 24000:
        line += f"  gelu hot {hot(f1):.1f} in-situ {insitu(f1):.1f} us | gelu+daux hot {hot(f2):.1f} in-situ {insitu(f2):.1f} us"
    print(line, flush=True)
# light epilogues: plain / bias / bias + q-scale (bf16 out), bias + fp32 residual (fp32 out)
for (M, N, K) in ((24000, 1280, 1280), (24000, 1280, 5120), (24000, 3840, 1280), (24000, 1280, 3840), (23808, 1600, 1024), (11520, 3520, 1344), (23900, 1600, 1024)):
    g = torch.Generator(device="cuda").manual_seed(M + N + K + 1)
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(bf); W = (torch.randn(N, K, device="cuda", generator=g) * 0.03).to(bf)
    bias = torch.randn(N, device="cuda", generator=g) * 0.1; res = torch.randn(M, N, device="cuda", generator=g)
    Cp = torch.full((M, N), float("nan"), dtype=bf, device="cuda"); Cb = torch.full((M, N), float("nan"), dtype=bf, device="cuda")
    Cs = torch.full((M, N), float("nan"), dtype=bf, device="cuda"); Cr = torch.full((M, N), float("nan"), device="cuda")
    nq = (N // 12) * 4
    before = ops.gemm_dispatch_log()
    fp = lambda: ops.gemm_nt(A, W, Cp, M, N, K)
    fb = lambda: ops.gemm_nt(A, W, Cb, M, N, K, bias=bias)
    fs = lambda: ops.gemm_nt(A, W, Cs, M, N, K, bias=bias, flags=L.EPI_SCALE_N, scale=0.125, scale_ncols=nq)
    fr = lambda: ops.gemm_nt(A, W, Cr, M, N, K, bias=bias, residual=res)
    fp(); fb(); fs(); fr()
    after = ops.gemm_dispatch_log()
    kern = ",".join(k for k in after if after[k] != before.get(k, 0))
    torch.cuda.synchronize()
    out[("light", M, N, K)] = (Cp.cpu(), Cb.cpu(), Cs.cpu(), Cr.cpu())
    line = f"light M={M} N={N} K={K}: [{kern}]"
    if M == 24000:
        line += f"\n      in-situ us: plain {insitu(fp):.1f}  bias {insitu(fb):.1f}  bias+scale {insitu(fs):.1f}  bias+residual {insitu(fr):.1f}   | hot: plain {hot(fp):.1f} scale {hot(fs):.1f} residual {hot(fr):.1f}"
    print(line, flush=True)
torch.save(out, sys.argv[1])
'''
lib = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else None
MODE = sys.argv[2] if len(sys.argv) > 2 else "15"
files = {}
for rep in range(2):
    for mode in ("0", MODE):
        f = tempfile.mktemp(suffix=f"_ntd{mode}.pt")
        env = dict(os.environ, DICOW_NT_DEFER=mode)
        if lib: env["DICOW_HIP_LIB"] = lib
        r = subprocess.run([sys.executable, "-c", CHILD, f], env=env, capture_output=True, text=True)
        print(f"--- DICOW_NT_DEFER={mode}\n" + (r.stdout.strip() if r.returncode == 0 else r.stdout + r.stderr[-1500:]), flush=True)
        files.setdefault(mode, f)
import torch
a, b = torch.load(files["0"]), torch.load(files[MODE])
for k in a:
    names = ("gelu C", "gelu+daux C", "gelu+daux aux") if k[0] == "gelu" else ("plain", "bias", "bias+scale", "bias+residual")
    for name, x, y in zip(names, a[k], b[k]):
        same = torch.equal(x, y)
        nan = int(torch.isnan(y.float()).sum())
        d = (x.float() - y.float()).abs()
        print(f"{k} {name:14s} bit-identical={same}  nan={nan}  max|d|={float(d.nan_to_num(1e9).max()):.4g}  differing={int((x != y).sum())}")
