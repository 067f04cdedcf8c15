"""Time one CTC prefix-scoring call at large-v3-turbo decoding sizes (B hypotheses x 500 candidates x 375 frames) and the
torch formulation of the reference's frame loop beside it.  usage: python tools/bench_ctc_prefix.py [B] [T]"""
import sys
import torch
sys.path.insert(0, ".")
import amd_pkg
amd_pkg.load()
from ts_asr_whisper_amd import ctc_decoding as cd

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
Tn = int(sys.argv[2]) if len(sys.argv) > 2 else 375
V1, C, eos = 51867, 500, 50257
g = torch.Generator().manual_seed(0)
logits = (torch.randn(B, Tn, V1, generator=g) * 3).to(torch.bfloat16).cuda()
sc = cd.CtcPrefixScorer(logits, V1 - 1, eos)
r0 = sc.initial_state()
cs = torch.stack([torch.randperm(50364, generator=g)[:C] for _ in range(B)]).cuda()
dl, last, rows = torch.zeros(B, dtype=torch.int32).cuda(), torch.full((B,), V1 - 1, dtype=torch.int32).cuda(), torch.arange(B).cuda()


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


t = timeit(lambda: sc(rows, cs, dl, last, r0))
out_bytes = B * Tn * 2 * C * 4
print(f"dicow_ctc_prefix_score B={B} C={C} T={Tn}: {t*1e3:.0f} us ({out_bytes/t/1e6:.0f} GB/s of state written; {B*C*Tn/t/1e6:.1f} G label-frames/s)")

# the reference's formulation: a Python loop over frames of torch ops (decoding.py:95-107), on the same device
x = torch.log_softmax(logits.float(), -1)
xs = torch.gather(x, 2, cs[:, None, :].expand(-1, Tn, -1))
xb = x[..., V1 - 1]
rsum = torch.logaddexp(r0[..., 0], r0[..., 1])


def ref_loop():
    r = torch.full((B, Tn, 2, C), -1e10, device="cuda")
    r[:, 0, 0] = xs[:, 0]
    phi = rsum[..., None].expand(-1, -1, C)
    for t_ in range(1, Tn):
        r[:, t_, 0] = torch.logaddexp(r[:, t_ - 1, 0], phi[:, t_ - 1]) + xs[:, t_]
        r[:, t_, 1] = torch.logaddexp(r[:, t_ - 1, 0], r[:, t_ - 1, 1]) + xb[:, t_][:, None]
    return r


t2 = timeit(ref_loop, 3)
print(f"frame loop of torch ops (reference formulation) on the same GPU: {t2:.1f} ms -> {t2/t:.0f}x")
