"""The encoder layer's 8 NT GEMM shapes (B=16, whisper-large-v3-turbo): this library's fused kernels against hipBLASLt
(torch.nn.functional.linear / torch.matmul in bf16) on the same box, in the same process, interleaved.

    python tools/gemm_vs_hipblaslt.py [out.json]

Three figures per shape (TFLOP/s of the algorithmic 2MNK):
  ours_fused      the kernel the training step launches (bias / scale / GELU+gelu' / fp32 residual / x gelu' + column sums)
  ours_plain      the same tile code with no epilogue operands (bf16 store)
  lt_plain        hipBLASLt, plain bf16 GEMM (no bias)
  lt_equiv        hipBLASLt + the torch element-wise ops a library user needs for the same result (what the fused kernel replaces)
Each figure is the median of ROUNDS interleaved rounds; between timed launches a 256 MB copy evicts L2 / MALL-resident
operands and cools the clock to in-step conditions (mode "insitu"); mode "hot" is the back-to-back loop."""
import json, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import amd_pkg; amd_pkg.load()
from ts_asr_whisper_amd import ops, _lib as L

bf = torch.bfloat16
M = int(os.environ.get("GEMM_M", "24000"))
ROUNDS = int(os.environ.get("ROUNDS", "7"))
dev = "cuda"
torch.manual_seed(0)
def rnd(*s, sc=0.5): return (torch.randn(*s, device=dev) * sc).to(bf)

SHAPES = [  # name, N, K, epilogue
    ("qkv  fwd  bias+qscale", 3840, 1280, "qkv"),
    ("out  fwd  bias+res f32", 1280, 1280, "res"),
    ("fc1  fwd  bias+gelu+dgelu", 5120, 1280, "gelu"),
    ("fc2  fwd  bias+res f32", 1280, 5120, "res"),
    ("fc2  dgrad x gelu' +colsum", 5120, 1280, "mulaux"),
    ("fc1  dgrad plain", 1280, 5120, "plain"),
    ("out  dgrad plain", 1280, 1280, "plain"),
    ("qkv  dgrad plain", 1280, 3840, "plain"),
]
evict_a = torch.empty(1 << 28, dtype=torch.uint8, device=dev); evict_b = torch.empty(1 << 28, dtype=torch.uint8, device=dev)

def time_one(fn, insitu):
    if insitu:
        evict_a.copy_(evict_b)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record()
    return s, e

def run(insitu):
    rows = []
    for name, N, K, epi in SHAPES:
        A, W = rnd(M, K), rnd(N, K, sc=0.03)
        bias = torch.randn(N, device=dev) * 0.1
        Cb = torch.empty(M, N, dtype=bf, device=dev); Cf = torch.empty(M, N, device=dev)
        res = torch.randn(M, N, device=dev); aux = rnd(M, N); cs = torch.zeros(N, device=dev)
        if epi == "qkv":
            fused = lambda: ops.gemm_nt(A, W, Cb, M, N, K, bias=bias, flags=L.EPI_SCALE_N, scale=0.125, scale_ncols=N // 3)
            def equiv():
                y = F.linear(A, W, bias.to(bf)); y[:, :N // 3] *= 0.125; return y
        elif epi == "res":
            fused = lambda: ops.gemm_nt(A, W, Cf, M, N, K, bias=bias, residual=res)
            equiv = lambda: res + F.linear(A, W, bias.to(bf))
        elif epi == "gelu":
            fused = lambda: ops.gemm_nt(A, W, Cb, M, N, K, bias=bias, aux=aux, flags=L.EPI_GELU | L.EPI_GELU_DAUX)
            def equiv():
                u = F.linear(A, W, bias.to(bf)); return F.gelu(u), u          # (the derivative is not even computed here)
        elif epi == "mulaux":
            fused = lambda: ops.gemm_nt(A, W, Cb, M, N, K, aux=aux, flags=L.EPI_MUL_AUX, colsum_out=cs)
            def equiv():
                g = torch.matmul(A, W.t()) * aux; return g, g.float().sum(0)
        else:
            fused = lambda: ops.gemm_nt(A, W, Cb, M, N, K)
            equiv = lambda: torch.matmul(A, W.t())
        plain = lambda: ops.gemm_nt(A, W, Cb, M, N, K)
        lt = lambda: torch.matmul(A, W.t())
        fns = {"ours_fused": fused, "ours_plain": plain, "lt_plain": lt, "lt_equiv": equiv}
        for f in fns.values():
            for _ in range(2): f()
        torch.cuda.synchronize()
        ev = {k: [] for k in fns}
        for _ in range(ROUNDS):
            for k, f in fns.items():
                ev[k].append(time_one(f, insitu))
        torch.cuda.synchronize()
        fl = 2.0 * M * N * K
        row = {"shape": name, "M": M, "N": N, "K": K}
        for k in fns:
            ms = statistics.median(s.elapsed_time(e) for s, e in ev[k])
            row[k + "_us"] = round(ms * 1e3, 1); row[k + "_tflops"] = round(fl / ms / 1e9, 1)
        row["fused_over_lt_plain"] = round(row["lt_plain_us"] / row["ours_fused_us"], 3)
        row["plain_over_lt_plain"] = round(row["lt_plain_us"] / row["ours_plain_us"], 3)
        row["fused_over_lt_equiv"] = round(row["lt_equiv_us"] / row["ours_fused_us"], 3)
        rows.append(row)
        print(f"{'insitu' if insitu else 'hot':6s} {name:28s} N{N:5d} K{K:5d}  ours fused {row['ours_fused_tflops']:7.1f}  plain {row['ours_plain_tflops']:7.1f}"
              f"  | hipBLASLt plain {row['lt_plain_tflops']:7.1f}  equiv {row['lt_equiv_tflops']:7.1f} TF"
              f"  | fused/lt_plain {row['fused_over_lt_plain']:.2f} plain/lt_plain {row['plain_over_lt_plain']:.2f} fused/lt_equiv {row['fused_over_lt_equiv']:.2f}", flush=True)
        del A, W, Cb, Cf, res, aux
    return rows

out = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "rounds": ROUNDS,
       "lib": os.environ.get("DICOW_HIP_LIB", "libdicow_hip.so"),
       "insitu": run(True), "hot": run(False)}
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        json.dump(out, f, indent=1)
