"""Micro-benchmarks of the HIP kernels at whisper-large-v3-turbo B=16 shapes (run on the GPU box)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg
amd_pkg.load()
from ts_asr_whisper_amd import ops, _lib as L

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters   # ms

B, T, D, F, H = 16, 1500, 1280, 5120, 20
M = B * T
res = {}
bf = torch.bfloat16
def rnd(*s): return (torch.randn(*s, device="cuda") * 0.5).to(bf)

for name, (m, n, k) in {"qkv": (M, 3 * D, D), "out": (M, D, D), "fc1": (M, F, D), "fc2": (M, D, F)}.items():
    A, W = rnd(m, k), rnd(n, k)
    C = torch.empty(m, n, dtype=bf, device="cuda")
    bias = torch.randn(n, device="cuda")
    t = timeit(lambda: ops.gemm_nt(A, W, C, m, n, k, bias=bias))
    t2 = timeit(lambda: torch.matmul(A, W.t(), out=C))
    res["nt_" + name] = {"ms": t, "TF": 2 * m * n * k / t / 1e9, "torch_ms": t2, "torch_TF": 2 * m * n * k / t2 / 1e9}
    print(name, res["nt_" + name], flush=True)

for name, (mk, n1, n2) in {"w_qkv": (M, 3 * D, D), "w_fc1": (M, F, D), "w_fc2": (M, D, F)}.items():
    A, Bm = rnd(mk, n1), rnd(mk, n2)
    C = torch.zeros(n1, n2, device="cuda")
    t = timeit(lambda: ops.gemm_tn(A, Bm, C, mk, n1, n2))
    t2 = timeit(lambda: torch.matmul(A.t(), Bm))
    res["tn_" + name] = {"ms": t, "TF": 2 * mk * n1 * n2 / t / 1e9, "torch_ms": t2, "torch_TF": 2 * mk * n1 * n2 / t2 / 1e9}
    print(name, res["tn_" + name], flush=True)

qkv = rnd(B, T, 3 * D)
q, k, v = (qkv[:, :, i * D:(i + 1) * D].view(B, T, H, 64) for i in range(3))
o = torch.empty(B, T, H, 64, dtype=bf, device="cuda"); lse = torch.empty(B, H, T, device="cuda")
t = timeit(lambda: ops.attn_fwd(q, k, v, o, lse))
fl = 4 * B * H * T * T * 64
qq, kk, vv = (x.permute(0, 2, 1, 3) for x in (q, k, v))
t2 = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(qq, kk, vv, scale=1.0))
res["attn_fwd"] = {"ms": t, "TF": fl / t / 1e9, "torch_ms": t2, "torch_TF": fl / t2 / 1e9}
print("attn_fwd", res["attn_fwd"], flush=True)

h = torch.randn(M, D, device="cuda"); st = torch.softmax(torch.randn(B, 4, T, device="cuda"), 1)
w = [torch.randn(D, device="cuda") for _ in range(4)]; b = [torch.randn(D, device="cuda") for _ in range(4)]
ho = torch.empty_like(h); y = torch.empty(M, D, dtype=bf, device="cuda"); mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
lw, lb = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
t = timeit(lambda: ops.fddt_ln_fwd(h, M, D, mode=1, stno=st, T=T, w=w, b=b, h_out=ho, ln_w=lw, ln_b=lb, y_bf16=y, mean=mean, rstd=rstd))
byt = M * D * (4 + 4 + 2)
res["fddt_ln_fwd"] = {"ms": t, "GBps": byt / t / 1e6}
print("fddt_ln_fwd", res["fddt_ln_fwd"], flush=True)
dy = rnd(M, D); gres = torch.randn(M, D, device="cuda"); g0 = torch.empty_like(h); g0b = torch.empty(M, D, dtype=bf, device="cuda")
dlw = torch.zeros(D, device="cuda"); dlb = torch.zeros(D, device="cuda"); cs = torch.zeros(D, device="cuda")
dw = [torch.zeros(D, device="cuda") for _ in range(4)]; db = [torch.zeros(D, device="cuda") for _ in range(4)]
t = timeit(lambda: ops.fddt_ln_bwd(h, M, D, mode=1, stno=st, T=T, w=w, b=b, ln_w=lw, mean=mean, rstd=rstd, d_y=dy, g_res=gres,
                                   g_out=g0, g_out_bf16=g0b, dln_w=dlw, dln_b=dlb, dw=dw, db=db, colsum_out=cs))
byt = M * D * (4 + 2 + 4 + 4 + 2)
res["fddt_ln_bwd"] = {"ms": t, "GBps": byt / t / 1e6}
print("fddt_ln_bwd", res["fddt_ln_bwd"], flush=True)
# ---- the log-mel front end (row A1): 16 clips x 30 s of 16 kHz audio -> [16, 128, 3000] fp32
from ts_asr_whisper_amd.features import log_mel, N_SAMPLES
from ts_asr_whisper_amd.augment import BatchAugmenter
wave = torch.randn(B, N_SAMPLES, device="cuda") * 0.1
for M_ in (80, 128):
    t = timeit(lambda: log_mel(wave, M_), iters=10)
    frames = N_SAMPLES // 160
    fl = B * frames * (2.0 * 400 * 201 * 2 + 2.0 * 201 * M_)
    byt = B * (N_SAMPLES * 4 + 3 * M_ * frames * 4)          # wave read; mel written, re-read and re-written by the clamp pass
    res[f"logmel_{M_}"] = {"us": t * 1e3, "GFLOPs_fp32": fl / t / 1e6, "algorithmic_MB": byt / 1e6, "GBps": byt / t / 1e6,
                           "note": "logmel_kernel (direct DFT, fp32 VALU) + logmel_finalize_kernel"}
    print(f"logmel_{M_}", res[f"logmel_{M_}"], flush=True)
mel = torch.randn(B, 128, 3000, device="cuda").clamp_(-1.5, 1.5)
aug = BatchAugmenter(stno_segment_augment_prob=1.0, spec_aug_prob=1.0)
t = timeit(lambda: aug({"input_features": mel, "stno_mask": st}), iters=10)
res["batch_augmenter"] = {"us": t * 1e3, "note": "STNO segment augmentation + joint SpecAug forced on, host planner included"}
print("batch_augmenter", res["batch_augmenter"], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/bench_kernels.json", "w"), indent=1)
