"""Diagnostic: per-workgroup timeline of the 4-wave NT kernel (DICOW_HIP_LIB=<libdicow_hip.so built with -DNTW_PROFILE>).
   NOTE: the -DNTW_PROFILE build of the current kernel crashes the ROCm 7.2 compiler (inliner segfault); the numbers quoted in
   gemm.hip / DESIGN.md come from the commit that introduced the persistent kernel.  tools/sweep_k.py and tools/probe_dma.hip
   give the same decomposition (k-loop slope, prologue + epilogue intercept, L2 -> LDS ceiling) without instrumentation.
   python tools/profile_ksteps.py M N K"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg; amd_pkg.load()
from ts_asr_whisper_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4])
bf = torch.bfloat16
A = (torch.randn(M, K, device="cuda") * 0.5).to(bf); W = (torch.randn(N, K, device="cuda") * 0.5).to(bf)
C = torch.empty(M, N, dtype=bf, device="cuda")
nblk = ((M + 255) // 256) * ((N + 255) // 256)
prof = torch.zeros(nblk * 14, dtype=torch.int64, device="cuda")
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(5):
    s.record(); ops.gemm_nt(A, W, C, M, N, K, aux=prof); e.record()
torch.cuda.synchronize()
p = prof[:nblk * 6].view(-1, 6).double().cpu()
fine = prof[nblk * 6:].view(-1, 8).double().cpu() / 100
cyc, wall = p[:, 0], p[:, 1]
t0 = p[:, 2].min()
st, ks, ke, en = (p[:, 2] - t0) / 100, (p[:, 3] - t0) / 100, (p[:, 4] - t0) / 100, (p[:, 5] - t0) / 100   # us
nk = K // 64
print(f"M{M} N{N} K{K}: kernel {s.elapsed_time(e)*1e3:.1f} us ({2*M*N*K/s.elapsed_time(e)/1e9:.0f} TF), {nblk} workgroups")
print(f"  k-loop {cyc.mean():.0f} cycles = {cyc.mean()/nk:.0f}/k-step (2048 = MFMA-bound); shader clk {(cyc/(wall*10e-9)).mean()*1e-9:.3f} GHz; k-step {wall.mean()*10/nk:.0f} ns")
print(f"  per tile: k-loop {(ke-ks).mean():.2f} us, epilogue {(en-ke).mean():.2f} us")
order = torch.argsort(st)
print(f"  start times (us) sorted: first {st[order[:3]].tolist()} .. #256 {st[order[min(255,nblk-1)]]:.1f}, #257 {st[order[min(256,nblk-1)]]:.1f}; last end {en.max():.1f}")
first = order[:256]; rest = order[256:]
if len(rest):
    print(f"  round 1: end mean {en[first].mean():.1f} (min {en[first].min():.1f} max {en[first].max():.1f}); round 2 start mean {st[rest].mean():.1f}, k-loop {(ke-ks)[rest].mean():.1f} vs {(ke-ks)[first].mean():.1f} us")
print(f"  epilogue split (us): coords+DMA issue {fine[:,0].mean():.2f}, barrier {fine[:,1].mean():.2f}, passes j0..3 " + ", ".join(f"{fine[:,2+j].mean():.2f}" for j in range(4)))
kl, ep = (ke - ks), (en - ke)
print(f"  k-loop us: min {kl.min():.1f} mean {kl.mean():.1f} max {kl.max():.1f} std {kl.std():.1f}; epilogue us: min {ep.min():.1f} mean {ep.mean():.1f} max {ep.max():.1f}")
ncu = min(256, nblk)
xcd = torch.arange(nblk) % ncu % 8
print("  k-loop mean per XCD:", [round(float(kl[xcd == x].mean()), 1) for x in range(8)])
print("  epilogue mean per XCD:", [round(float(ep[xcd == x].mean()), 1) for x in range(8)])
cu = torch.arange(nblk) % ncu
percu = torch.stack([kl[cu == c].mean() for c in range(ncu)])
print(f"  k-loop mean per CU: min {percu.min():.1f} max {percu.max():.1f}; slowest CUs {torch.argsort(percu)[-6:].tolist()}")
