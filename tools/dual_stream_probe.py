"""Probe: encoder forward of B=16 as ONE stream vs TWO half-batches on two HIP streams with the persistent GEMM limited to
half the CUs each (do memory-bound phases of one half hide under MFMA phases of the other?).  python tools/dual_stream_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg
pkg = amd_pkg.load()
from ts_asr_whisper_amd import ops
from ts_asr_whisper_amd.data import synthetic_batch
cfg = pkg.DiCoWConfig.preset("whisper-large-v3-turbo", use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True,
                             fddt_init="suppressive", non_target_fddt_value=0.5)
torch.manual_seed(0)
model = pkg.DiCoWForConditionalGeneration(cfg).cuda()
enc = model.model.encoder
b = synthetic_batch(cfg, 16, 128, seed=1)
x, st = b["input_features"], b["stno_mask"]
def fwd(xx, ss):
    with torch.no_grad():
        return enc(xx, stno_mask=ss).last_hidden_state
def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
ref = fwd(x, st)
print(f"single stream B=16: {timed(lambda: fwd(x, st)):.2f} ms")
# round 6: two normal-priority pool streams may share ONE hardware queue with each other / the default stream (kernels of a queue never
# overlap: profiles/r06_dp_emulated.txt) -- STREAM_PRIO=1 makes the second stream a high-priority one (a queue of its own)
s1, s2 = torch.cuda.Stream(), (torch.cuda.Stream(priority=-1) if os.environ.get("STREAM_PRIO") == "1" else torch.cuda.Stream())
print("second stream:", "high priority (own hardware queue)" if os.environ.get("STREAM_PRIO") == "1" else "normal priority (pool stream)")
xs, sts = (x[:8].contiguous(), x[8:].contiguous()), (st[:8].contiguous(), st[8:].contiguous())
for cus in (118, 128, 0):
    ops.set_gemm_cus(cus)
    outs = [None, None]
    def dual():
        cur = torch.cuda.current_stream()
        for i, s in enumerate((s1, s2)):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs[i] = fwd(xs[i], sts[i])
        cur.wait_stream(s1); cur.wait_stream(s2)
    t = timed(dual)
    err = float((torch.cat(outs) - ref).abs().max())
    print(f"two streams B=8+8, gemm_cus={cus}: {t:.2f} ms   (max |diff| vs single {err:.3g})")
ops.set_gemm_cus(0)
print(f"single stream B=8 alone x2 sequential: {2 * timed(lambda: fwd(xs[0], sts[0])):.2f} ms")
