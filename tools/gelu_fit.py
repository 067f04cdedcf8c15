"""Fit and exhaustive check of the GELU form of csrc/common.h: Phi(x) = 1 / (1 + 2^(x q(x^2))), q of degree 6 in x^2 (the
coefficients GELU_Q0..Q6), evaluated the way the kernels do (fp32, fma) over every finite bf16 input.  CPU only: python tools/gelu_fit.py"""
import numpy as np, torch
from scipy.special import ndtr, log_ndtr
xs = np.linspace(1e-3, 6.0, 12000)
y = (log_ndtr(-xs) - log_ndtr(xs)) / np.log(2.0) / xs
deg = 6
A = np.stack([xs ** (2 * k) for k in range(deg + 1)], 1)
Phi = ndtr(xs)
wt = xs * Phi * (1 - Phi) * xs * np.log(2)
w = np.ones_like(xs)
for it in range(400):
    co, *_ = np.linalg.lstsq(A * (wt * w)[:, None], y * wt * w, rcond=None)
    e = np.abs((A @ co - y) * wt)
    w = w * (1 + 3 * e / e.max()); w /= w.mean()
print("coefs:", ", ".join(f"{c:.17e}" for c in co))
f32 = np.float32
def gelu32(x):
    def fma(a, b, c): return (a.astype(np.float64) * b.astype(np.float64) + np.float64(c)).astype(f32)
    with np.errstate(all="ignore"):
        s = (x.astype(np.float64) * x).astype(f32)
        q = np.full_like(x, f32(co[-1]))
        for k in range(deg - 1, -1, -1): q = fma(q, s, f32(co[k]))
        wv = (x.astype(np.float64) * q).astype(f32)
        e = np.exp2(wv.astype(np.float64)).astype(f32)
        cdf = (1.0 / (e + f32(1)).astype(np.float64)).astype(f32)
        return (x.astype(np.float64) * cdf).astype(f32), cdf
u = np.arange(65536, dtype=np.uint32)
xa = (u << 16).view(np.float32); xa = xa[np.isfinite(xa)]
g, cdf = gelu32(xa)
ref = xa.astype(np.float64) * ndtr(xa.astype(np.float64))
print("non-finite:", (~np.isfinite(g)).sum(), "max abs err all bf16:", np.abs(g - ref).max(), "cdf range", cdf.min(), cdf.max())
m = np.abs(xa) > 6
print("beyond |x|>6: max abs err", np.abs(g - ref)[m & (np.abs(xa)<1e30)].max())
x = torch.linspace(-9, 9, 256 * 64).bfloat16().float().numpy()
g, _ = gelu32(x); ref = x.astype(np.float64) * ndtr(x.astype(np.float64))
gb = torch.from_numpy(g).bfloat16().float().numpy(); rb = torch.from_numpy(ref.astype(f32)).bfloat16().float().numpy()
for thr in (0, 1e-6, 1e-4):
    mm = np.abs(ref) > thr
    print("grid: |ref|>", thr, "agree", (gb[mm] == rb[mm]).mean())
rel = np.abs(g - ref) / np.maximum(np.abs(ref), 1e-300)
for lo, hi in ((-6,-5),(-5,-4),(-4,-3),(-3,-2),(-2,0),(0,9)):
    mm = (x >= lo) & (x < hi); print(lo, hi, "max rel", rel[mm].max(), "max abs", np.abs(g - ref)[mm].max())
