"""LayerNorm folded into the GEMMs on either side of it (DICOW_EPI_LNSTAT / DICOW_EPI_LNFOLD, include/dicow_hip.h).

Replaces HF WhisperEncoderLayer's `self_attn_layer_norm` -> q/k/v and `final_layer_norm` -> fc1 (HF modeling_whisper.py:392-405,
reached from the reference's encoder.py:216-221) without the LayerNorm launch.  Kernel level: the producer's bf16 copy and row
partials against torch, the consumer against torch's LayerNorm + Linear (fp32 and with the fold's own rounding points).
EXPERIMENTAL: the fold measured slower (profiles/r04_lnfold.txt) and is not in the stable ABI; these tests load the experiments
library (ts-asr-whisper_amd/libdicow_hip_exp.so: csrc/build.sh --exp, built by __graft_entry__.build()).
Run with `pytest -m gpu`."""
import os

import pytest
import torch

import amd_pkg

pytestmark = pytest.mark.gpu
amd_pkg.load()


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ts_asr_whisper_amd import ops as _ops, _lib as L
    exp = os.path.join(os.path.dirname(os.path.abspath(L.__file__)), "libdicow_hip_exp.so")
    if not os.path.exists(exp):
        pytest.skip("experiments library missing: ts-asr-whisper_amd/csrc/build.sh --exp")
    old = (L.LIB_PATH, L._lib)
    L.LIB_PATH, L._lib = exp, None
    assert L.has_experimental()
    yield _ops
    L.LIB_PATH, L._lib = old


def _bf(t):
    return t.to(torch.bfloat16).float()


def _rows_stats(stats, nslots):
    s = stats[:, :nslots, :].double().sum(1)
    return s[:, 0], s[:, 1]


@pytest.mark.parametrize("M,N,K,with_fddt", [(24000, 1280, 1280, False), (11001, 1280, 320, False), (24000, 1280, 5120, True),
                                             (17300, 640, 256, False), (12288, 1280, 128, True)])
def test_producer_stores_bf16_copy_and_row_partials(ops, M, N, K, with_fddt):
    """out-proj / fc2 epilogue with LNSTAT: the fp32 result is bit-identical to the epilogue without it, the bf16 copy is its
    rounding, and the 4 * N / 320 partial slots of every row add up to (sum, sum of squares) of the fp32 row."""
    from ts_asr_whisper_amd import _lib as L
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g)).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = (torch.randn(M, N, generator=g) * 2 + 0.3 * torch.randn(M, 1, generator=g)).cuda()
    fd = None
    if with_fddt:
        Tn = 1500 if M % 1500 == 0 else M // 4
        Bn = M // Tn
        st = torch.softmax(torch.randn(Bn, 4, Tn, generator=g), 1).cuda()
        w = [(1 + 0.1 * torch.randn(N, generator=g)).cuda() for _ in range(4)]
        b = [(0.1 * torch.randn(N, generator=g)).cuda() for _ in range(4)]
        rowmask = torch.zeros((M + 191) // 192 * 192 + 64, 4, device="cuda")
        rowmask[:Bn * Tn].view(Bn, Tn, 4).copy_(st.permute(0, 2, 1))
        fd = (w, b, rowmask)
    C0 = torch.empty(M, N, device="cuda")
    assert ops.gemm_nt(A, W, C0, M, N, K, bias=bias, residual=res, query_lnstat=True)
    ops.gemm_nt(A, W, C0, M, N, K, bias=bias, residual=res, fddt=fd)
    C1 = torch.full((M, N), float("nan"), device="cuda")
    hb = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    stats = torch.zeros(M, L.LN_SLOTS, 2, device="cuda")
    ops.gemm_nt(A, W, C1, M, N, K, bias=bias, residual=res, fddt=fd, ln_stat=(hb, stats))
    torch.cuda.synchronize()
    assert torch.equal(C1, C0)
    assert torch.equal(hb, C0.to(torch.bfloat16))
    nsl = 4 * (N // 320)
    S, Q = _rows_stats(stats, nsl)
    assert float(stats[:, nsl:].abs().max()) == 0.0 if nsl < L.LN_SLOTS else True
    Sr, Qr = C0.double().sum(1), (C0.double() ** 2).sum(1)
    assert float(((S - Sr).abs() / (Qr.sqrt() * N ** 0.5 + 1e-30)).max()) < 1e-6
    assert float(((Q - Qr).abs() / Qr).max()) < 1e-6
    # a second launch writes the same bits (fixed slots, no atomics)
    stats2 = torch.zeros_like(stats)
    ops.gemm_nt(A, W, C1, M, N, K, bias=bias, residual=res, fddt=fd, ln_stat=(hb, stats2))
    assert torch.equal(stats, stats2)


def test_producer_refuses_shapes_without_whole_320_column_tiles(ops):
    from ts_asr_whisper_amd import _lib as L
    M, N, K = 24000, 1284, 128
    A, W = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda"), torch.zeros(N, K, dtype=torch.bfloat16, device="cuda")
    C, res, bias = torch.empty(M, N, device="cuda"), torch.zeros(M, N, device="cuda"), torch.zeros(N, device="cuda")
    assert not ops.gemm_nt(A, W, C, M, N, K, bias=bias, residual=res, query_lnstat=True)
    with pytest.raises(L.DicowError):
        ops.gemm_nt(A, W, C, M, N, K, bias=bias, residual=res,
                    ln_stat=(torch.empty(M, N, dtype=torch.bfloat16, device="cuda"), torch.zeros(M, 16, 2, device="cuda")))
    # below the persistent kernel's threshold: refused, not silently un-normalised
    assert not ops.gemm_nt(A[:500], W[:1280], C, 500, 1280, K, bias=bias[:1280], residual=res, query_lnstat=True)


@pytest.mark.parametrize("M,N,kind", [(24000, 3840, "qkv"), (24000, 5120, "gelu"), (24000, 5120, "gelu_daux"), (12300, 2560, "qkv"),
                                      (12301, 5120, "gelu")])
def test_fold_pair_equals_layernorm_then_linear(ops, M, N, kind):
    """producer (out-proj-like, K = 320) -> consumer with the folded weight: the consumer's output equals LayerNorm(h) @ W^T + b
    computed by torch from the producer's fp32 h -- (a) with the fold's own rounding points (bf16(h), bf16(gamma W)) to fp32
    accumulation accuracy, (b) against the reference's AMP rounding points (bf16(LN(h)), bf16(W)) within bf16 rounding noise."""
    from ts_asr_whisper_amd import _lib as L
    D, K0, eps = 1280, 320, 1e-5
    g = torch.Generator().manual_seed(M + N)
    A0 = torch.randn(M, K0, generator=g).to(torch.bfloat16).cuda()
    W0 = (torch.randn(D, K0, generator=g) * K0 ** -0.5).to(torch.bfloat16).cuda()
    b0 = torch.randn(D, generator=g).cuda()
    res = (torch.randn(M, D, generator=g) + 0.5 * torch.randn(M, 1, generator=g)).cuda()
    res[:, 7] += 20.0                                                      # an outlier channel, like Whisper's residual stream
    h = torch.empty(M, D, device="cuda")
    hb = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    stats = torch.zeros(M, L.LN_SLOTS, 2, device="cuda")
    ops.gemm_nt(A0, W0, h, M, D, K0, bias=b0, residual=res, ln_stat=(hb, stats))
    gamma, beta = (1 + 0.2 * torch.randn(D, generator=g)).cuda(), (0.2 * torch.randn(D, generator=g)).cuda()
    W = (torch.randn(N, D, generator=g) * D ** -0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    Wf = torch.empty(N, D, dtype=torch.bfloat16, device="cuda")
    c, bf = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    ops.lnfold_prep(W, gamma, beta, bias, Wf, c, bf)
    assert torch.equal(Wf, (W * gamma).to(torch.bfloat16))
    assert float((c - Wf.float().sum(1)).abs().max()) < 1e-4
    assert float((bf - (bias + W.to(torch.bfloat16).float() @ beta)).abs().max()) < 1e-4
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    aux = None
    flags, kw = 0, {}
    if kind == "qkv":
        flags, kw = L.EPI_SCALE_N, dict(scale=0.125 * ops.LOG2E, scale_ncols=D)
    elif kind == "gelu":
        flags = L.EPI_GELU
    else:
        flags = L.EPI_GELU | L.EPI_GELU_DAUX
        aux = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(hb, Wf, out, M, N, D, bias=bf, aux=aux, flags=flags, ln_fold=(stats, c, D, eps), **kw)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out.float()).all())
    hd = h.double()
    mean, var = hd.mean(1, keepdim=True), hd.var(1, unbiased=False, keepdim=True)
    rstd = (var + eps).rsqrt()

    def act(y):
        if kind == "qkv":
            y = y.clone()
            y[:, :D] *= 0.125 * ops.LOG2E
            return y
        return torch.nn.functional.gelu(y.float().to(torch.bfloat16).float().double())       # GELU of the bf16-rounded pre-activation

    # (a) the fold's own rounding points, fp64 accumulation
    own = rstd * (hb.double() @ Wf.double().t()) - rstd * mean * c.double() + bf.double()
    ref_a = act(own)
    err_a = (out.double() - ref_a).abs()
    tol_a = 2.0 ** -7 * ref_a.abs() + 3e-3                                 # bf16 output rounding (+ a flipped rounding of the pre-activation)
    assert bool((err_a <= tol_a).float().mean() > 0.9995) and float((err_a / (ref_a.abs() + 1.0)).max()) < 4e-2
    # (b) the reference's rounding points
    xln = ((hd - mean) * rstd * gamma.double() + beta.double()).float().to(torch.bfloat16).double()
    ref_b = act(xln @ W.to(torch.bfloat16).double().t() + bias.double())
    exact = act(((hd - mean) * rstd * gamma.double() + beta.double()) @ W.double().t() + bias.double())
    e_fold = float((out.double() - exact).norm() / exact.norm())
    e_ref = float((ref_b.float().to(torch.bfloat16).double() - exact).norm() / exact.norm())
    print(f"{kind} M={M} N={N}: rel-L2 error vs exact fp64: fold {e_fold:.3e}, reference AMP rounding {e_ref:.3e}")
    assert e_fold < 1.5 * e_ref + 1e-4
    if aux is not None:                                                    # saved derivative, as the un-folded training epilogue stores it
        x = own.float().to(torch.bfloat16).double()
        d = 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-x * x / 2) / (2 * torch.pi) ** 0.5
        assert float((aux.double() - d).abs().max()) < 2e-2
