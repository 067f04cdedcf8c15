"""Attention micro-benchmark at the encoder self-attention shape (B=16, H=20, L=1500, head_dim 64), in-situ-like:
   a memory-bound copy between launches; fwd and bwd (delta + dq + dkv) timed separately.
   python tools/bench_attn.py [B H Lq Lk causal]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg; amd_pkg.load()
from ts_asr_whisper_amd import ops
B, H, Lq, Lk, causal = (int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (16, 20, 1500, 1500, 0)))
D = H * 64
bf = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.randn(B, Lq, 3 * D, device="cuda", generator=g)).to(bf)
qkv[:, :, :D] *= 0.125
kv = qkv if Lk == Lq else torch.randn(B, Lk, 3 * D, device="cuda", generator=g).to(bf)
q, k, v = qkv[:, :, :D].view(B, Lq, H, 64), kv[:, :, D:2 * D].view(B, Lk, H, 64), kv[:, :, 2 * D:].view(B, Lk, H, 64)
o = torch.empty(B, Lq, H, 64, dtype=bf, device="cuda"); lse = torch.empty(B, H, Lq, device="cuda")
d_o = (torch.randn(B, Lq, H, 64, device="cuda", generator=g) * 0.01).to(bf)
dqkv = torch.empty(B, Lq, 3 * D, dtype=bf, device="cuda"); dkv = dqkv if Lk == Lq else torch.empty(B, Lk, 3 * D, dtype=bf, device="cuda")
dq, dk, dv = dqkv[:, :, :D].view(B, Lq, H, 64), dkv[:, :, D:2 * D].view(B, Lk, H, 64), dkv[:, :, 2 * D:].view(B, Lk, H, 64)
delta = torch.empty(2, B, H, Lq, device="cuda")
x1, x2 = torch.empty(1 << 28, dtype=torch.uint8, device="cuda"), torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize(); evs = []
    for _ in range(iters):
        x1.copy_(x2)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); evs.append((s, e))
    torch.cuda.synchronize()
    return sum(s.elapsed_time(e) for s, e in evs) / iters
fl = 4.0 * B * H * Lq * Lk * 64 * (0.5 if causal else 1.0)
LOG2 = os.environ.get("ATTN_LOG2") == "1"          # the encoder's mode: q carries log2(e)
t = timeit(lambda: ops.attn_fwd(q, k, v, o, lse, causal=bool(causal), q_log2=LOG2))
print(f"attn_fwd B{B} H{H} Lq{Lq} Lk{Lk} causal{causal}: {t*1e3:.1f} us  {fl/t/1e9:.0f} TF")
for rep in range(int(os.environ.get("ATTN_BWD_REPS", "2"))):          # interleaved A/B: two-kernel form | fused form (dense, large problems only)
    t = timeit(lambda: ops.attn_bwd(q, k, v, o, d_o, lse, delta, dq, dk, dv, causal=bool(causal), dq_scale=0.125, q_log2=LOG2, fused=False))
    print(f"attn_bwd (dq+dkv): {t*1e3:.1f} us  {3.5*fl/t/1e9:.0f} TF executed (7 matmul passes), {2.5*fl/t/1e9:.0f} TF algorithmic")
    if not causal:
        t = timeit(lambda: ops.attn_bwd(q, k, v, o, d_o, lse, delta, dq, dk, dv, causal=False, dq_scale=0.125, q_log2=LOG2, fused=True))
        print(f"attn_bwd (fused, incl. stats kernel): {t*1e3:.1f} us  {2.5*fl/t/1e9:.0f} TF algorithmic (5 matmul passes), status {ops.attn_bwd_fused_status()}")
if os.environ.get("ATTN_PROFILE_FUSED"):     # only with a -DATTN_FUSED_PROFILE=1 build (tools/libv_fprof.so): per-workgroup cycle accounting of the fused backward
    ops.attn_bwd(q, k, v, o, d_o, lse, delta, dq, dk, dv, causal=False, dq_scale=0.125, q_log2=LOG2, fused=True); torch.cuda.synchronize()
    ws = ops.attn_bwd_fused_ws(B, H, Lq, Lk, q.device)
    nt, nkb = (Lq + 63) // 64, (Lk + 127) // 128
    nfl = B * H * nt * 4
    off = 4096 + ((nfl * 4 + 4095) // 4096) * 4096 + nfl * 4096
    nwg = 8 * ((B * H + 7) // 8) * nkb
    rec = ws[off:off + nwg * 128].view(torch.int64).view(nwg, 16).double().cpu()
    names = ["loop top, flag poll issue", "A: q block 0", "wait in the middle (store acks, poll)", "publish, flag check, sum request, q block 1", "wait + barrier X", "dS fragment reads, DMA issue, barrier Y", "dQ product", "wait for the sum (slow path: flag)", "LDS read + add, stores, bookkeeping"]
    for sel, lab in ((rec[:, 11] == 0, "key block 0"), ((rec[:, 11] > 0) & (rec[:, 11] < nkb - 1), "middle key blocks"), (rec[:, 11] == nkb - 1, "last key block")):
        r = rec[sel]
        print(f"fused bwd, {lab} ({int(sel.sum())} workgroups), cycles per 64-query tile (wave 0): " +
              ", ".join(f"{n} {r[:, i].mean() / nt:.0f}" for i, n in enumerate(names)) +
              f"; slow path in {r[:, 13].mean() / nt * 100:.0f} % of the tiles; loop total {r[:, 9].mean() / nt:.0f}/tile, {r[:, 10].mean() / 100:.1f} us per workgroup = {r[:, 9].mean() / r[:, 10].mean() * 0.1:.2f} GHz")
    t0 = rec[:, 12].min()
    print("fused bwd: workgroup start times (us after the first), percentiles 10/50/90/100:",
          [round(float(x) / 100, 1) for x in torch.quantile(rec[:, 12] - t0, torch.tensor([0.1, 0.5, 0.9, 1.0], dtype=torch.float64))],
          "; end of the last:", round(float((rec[:, 12] + rec[:, 10]).max() - t0) / 100, 1))
if os.environ.get("ATTN_PROFILE"):
    ops.attn_fwd(q, k, v, o, lse, causal=bool(causal)); torch.cuda.synchronize()
    nblk = B * H * ((Lq + 127) // 128)
    pr = lse.view(-1).view(torch.int64)[:nblk * 8].view(-1, 8).double().cpu()
    nt = pr[:, 6].mean()
    names = ["stage+wait+barrier", "QK mfma + V tr issue", "softmax", "PV", "lgkm+end barrier"]
    print(f"per k-tile cycles (wave 0, mean over {nblk} workgroups, {nt:.1f} tiles): " +
          ", ".join(f"{n} {pr[:, i].mean() / nt:.0f}" for i, n in enumerate(names)) + f"; loop total {pr[:,5].mean()/nt:.0f}/tile; epilogue {pr[:,7].mean():.0f}")
if os.environ.get("ATTN_PROFILE_DKV"):      # only with a build of the stamped dkv kernel (commit 948e096: -DATTN_PROFILE)
    ops.attn_bwd(q, k, v, o, d_o, lse, delta, dq, dk, dv, causal=bool(causal), dq_scale=0.125); torch.cuda.synchronize()
    nblk = B * H * ((Lk + 127) // 128)
    rows = (nblk + 39) // 40
    raw = dqkv.view(-1, 3 * D)[:rows, :D].contiguous().view(torch.int64).view(-1, 8)[:nblk].double().cpu()   # 64-byte records in the dq columns
    nt = raw[:, 3].mean()
    names = ["vmcnt wait + barrier", "q-block 0", "q-block 1"]
    print(f"dkv per q-tile cycles (wave 0, {nt:.1f} tiles): " + ", ".join(f"{n} {raw[:, i].mean() / nt:.0f}" for i, n in enumerate(names)) +
          f"; sum {raw[:, :3].sum(1).mean() / nt:.0f}/tile; loop total {raw[:, 4].mean():.0f} cycles in {raw[:, 5].mean() / 100:.1f} us "
          f"(100 MHz s_memrealtime) = {raw[:, 4].mean() / raw[:, 5].mean() * 0.1:.2f} GHz")
