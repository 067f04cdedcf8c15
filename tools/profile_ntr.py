"""Per-tile timeline of the ring NT GEMM (diagnostic build -DNTR_PROFILE: tools/build_variants.sh prof "-DNTR_PROFILE").
   DICOW_HIP_LIB=tools/libv_prof.so python tools/profile_ntr.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg; amd_pkg.load()
from ts_asr_whisper_amd import ops, _lib as L
bf = torch.bfloat16
M = 24000
ea = torch.empty(1 << 28, dtype=torch.uint8, device="cuda"); eb = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
def run(N, K, epi):
    A = (torch.randn(M, K, device="cuda") * 0.5).to(bf); W = (torch.randn(N, K, device="cuda") * 0.03).to(bf)
    bias = torch.randn(N, device="cuda") * 0.1
    ntiles = max(((M + 255) // 256) * ((N + 255) // 256), ((M + 191) // 192) * ((N + 319) // 320), ((M + 127) // 128) * ((N + 255) // 256))     # whichever tile shape is dispatched
    dbg = torch.zeros(ntiles * 6, dtype=torch.int64, device="cuda")
    a = L.GemmArgs()
    a.A, a.B = A.data_ptr(), W.data_ptr()
    a.M, a.N, a.K, a.lda, a.ldb, a.ldc, a.ldr, a.ldaux, a.batch = M, N, K, K, K, N, N, N, 1
    keep = []
    if epi == "plain":
        C = torch.empty(M, N, dtype=bf, device="cuda"); a.flags = 0
    elif epi == "res":
        C = torch.empty(M, N, device="cuda"); r = torch.randn(M, N, device="cuda"); keep.append(r)
        a.bias, a.residual, a.flags = bias.data_ptr(), r.data_ptr(), L.EPI_BIAS | L.EPI_RESIDUAL | L.EPI_OUT_F32
    elif epi == "gelu":
        C = torch.empty(M, N, dtype=bf, device="cuda"); x = torch.empty(M, N, dtype=bf, device="cuda"); keep.append(x)
        a.bias, a.aux, a.flags = bias.data_ptr(), x.data_ptr(), L.EPI_BIAS | L.EPI_GELU | L.EPI_GELU_DAUX
    elif epi == "geluinf":
        C = torch.empty(M, N, dtype=bf, device="cuda")
        a.bias, a.flags = bias.data_ptr(), L.EPI_BIAS | L.EPI_GELU
    a.C = C.data_ptr()
    a.colsum_ws = dbg.data_ptr()
    for _ in range(3):
        ea.copy_(eb)
        dbg.zero_()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); L.call_struct("dicow_gemm_nt", a); e.record()
        torch.cuda.synchronize()
    d = dbg.view(ntiles, 6).cpu()
    d = d[d[:, 0] > 0]
    ntiles = d.shape[0]
    t0 = int(d[:, 0][d[:, 0] > 0].min())
    kl = (d[:, 1] - d[:, 0]).float() / 100.0          # us
    ep = (d[:, 2] - d[:, 1]).float() / 100.0
    clk = d[:, 3].float()
    end = (int(d[:, 2].max()) - t0) / 100.0
    nk = K // 64
    # per workgroup: its tiles in time order
    per = {}
    clk_of = {}
    for i in range(ntiles):
        per.setdefault(int(d[i, 4]), []).append((int(d[i, 0]), int(d[i, 1]), int(d[i, 2])))
        clk_of[(int(d[i, 4]), int(d[i, 0]))] = float(d[i, 3])
    gaps = []
    by_order = {}
    for b, ts in per.items():
        ts.sort()
        for k, (a0, a1, a2) in enumerate(ts):
            by_order.setdefault(k, []).append(((a1 - a0) / 100.0, (a2 - a1) / 100.0, clk_of[(b, a0)]))
        for (a0, a1, a2), (b0, b1, b2) in zip(ts, ts[1:]):
            gaps.append((b0 - a2) / 100.0)
    print(f"N{N} K{K} {epi:8s}: kernel {s.elapsed_time(e)*1e3:7.1f} us (stamps span {end:7.1f}) | k-loop {kl.mean():6.2f} us/tile "
          f"({kl.mean()/nk*1e3:5.0f} ns/step, {clk.mean()/nk:5.0f} clk/step, min {kl.min():.1f} max {kl.max():.1f}) | epilogue {ep.mean():5.2f} us "
          f"(min {ep.min():.1f} max {ep.max():.1f}) | inter-tile gap {statistics.mean(gaps) if gaps else 0:.2f} us | {len(per)} WGs x {ntiles/len(per):.1f} tiles", flush=True)
    print("      k-loop / epilogue us by the tile's position in its workgroup's walk: " +
          "  ".join(f"#{k}: {statistics.mean(x[0] for x in v):.1f}/{statistics.mean(x[1] for x in v):.1f}" for k, v in sorted(by_order.items())), flush=True)
    # the same k-loops in SHADER CLOCKS (s_memtime) and the clock rate they imply: a slow first tile with the same clock count is the
    # board's clock, not the memory system
    print("      k-loop shader clocks per step / implied MHz by position: " +
          "  ".join(f"#{k}: {statistics.mean(x[2] for x in v) / nk:.0f}/{statistics.mean(x[2] for x in v) / statistics.mean(x[0] for x in v):.0f}"
                    for k, v in sorted(by_order.items())), flush=True)
for N, K, epi in ((3840, 1280, "plain"), (1280, 1280, "plain"), (1280, 5120, "plain"), (1280, 1280, "res"), (1280, 5120, "res"),
                  (5120, 1280, "gelu"), (5120, 1280, "geluinf"), (5120, 1280, "plain")):
    run(N, K, epi)
