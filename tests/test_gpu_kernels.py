"""Kernel-level parity on the MI355X: each HIP entry point (through the C ABI) vs the CPU oracle /
golden fixtures.  Run with `pytest -m gpu`."""
import math

import pytest
import torch

import amd_pkg
from oracle import dicow_oracle as O
from tests.util import load_golden, T, maxdiff

pytestmark = pytest.mark.gpu

pkg = amd_pkg.load()


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ts_asr_whisper_amd import ops as _ops
    return _ops


def dev(t, dtype=None):
    t = t.cuda()
    return t.to(dtype) if dtype is not None else t


CLS = ["silence_linear", "target_linear", "non_target_linear", "overlap_linear"]     # stno channel order S,T,N,O


# ------------------------------------------------------------------------------------------------ FDDT (golden F3)
@pytest.mark.parametrize("vn", ["diag", "diag_no_sil_ovl", "bias"])
def test_fddt_fwd_bit_exact_and_bwd(ops, vn):
    z = load_golden("f3_fddt")
    h, st, go = T(z, vn + ".h"), T(z, vn + ".stno"), T(z, vn + ".gout")
    B, Tn, D = h.shape
    mode = ops.MODE_BIAS if vn == "bias" else ops.MODE_DIAG
    w, b = [], []
    for c in CLS:
        if vn == "bias":
            key = f"{vn}.p.{c}"
            w.append(None)
            b.append(dev(T(z, key)) if key in z.files else None)
        else:
            kw = f"{vn}.p.{c}.weight"
            w.append(dev(T(z, kw)) if kw in z.files else None)
            b.append(dev(T(z, f"{vn}.p.{c}.bias")) if kw in z.files else None)
    hd, std = dev(h).contiguous(), dev(st).contiguous()
    out = torch.empty_like(hd)
    ops.fddt_ln_fwd(hd, B * Tn, D, mode=mode, stno=std, T=Tn, w=w, b=b, h_out=out)
    ref = T(z, vn + ".out")
    # bit-exact mask indexing AND arithmetic: the kernel evaluates the reference's fp32 expression order
    assert torch.equal(out.cpu(), ref), f"{vn}: max diff {maxdiff(out.cpu(), ref)}"
    # backward
    gd = dev(go).contiguous()
    g0 = torch.empty_like(hd)
    g0b = torch.empty(B * Tn, D, dtype=torch.bfloat16, device="cuda")
    dw = [torch.zeros(D, device="cuda") if x is not None else None for x in w]
    db = [torch.zeros(D, device="cuda") if x is not None else None for x in b]
    cs = torch.zeros(D, device="cuda")
    ops.fddt_ln_bwd(hd, B * Tn, D, mode=mode, stno=std, T=Tn, w=w, b=b, g_res=gd, g_out=g0, g_out_bf16=g0b, dw=dw, db=db,
                    colsum_out=cs)
    gh = T(z, vn + ".gh")
    assert maxdiff(g0.cpu(), gh) < 1e-5
    assert maxdiff(g0b.float().cpu().view_as(gh), gh) < 2e-2 * float(gh.abs().max())
    assert maxdiff(cs.cpu(), gh.sum((0, 1))) < 1e-3
    for i, c in enumerate(CLS):
        if vn == "bias":
            if db[i] is not None:
                ref_g = T(z, f"{vn}.g.{c}")
                assert maxdiff(db[i].cpu(), ref_g) < 1e-4 * max(1.0, float(ref_g.abs().max()))
        elif dw[i] is not None:
            for t_, nm in ((dw[i], "weight"), (db[i], "bias")):
                ref_g = T(z, f"{vn}.g.{c}.{nm}")
                assert maxdiff(t_.cpu(), ref_g) < 2e-4 * max(1.0, float(ref_g.abs().max())), (c, nm)


@pytest.mark.parametrize("D,rows,Tn", [(384, 1500, 1500), (1280, 777, 777), (512, 64, 16)])
def test_fddt_ln_fused_fwd_bwd_vs_oracle(ops, D, rows, Tn):
    g = torch.Generator().manual_seed(D + rows)
    B = rows // Tn
    h = torch.randn(B, Tn, D, generator=g)
    st = torch.softmax(torch.randn(B, 4, Tn, generator=g), 1)
    cfg = O.OracleConfig(d_model=D)
    p = {}
    for c in CLS:
        p[c + ".weight"] = 1 + 0.1 * torch.randn(D, generator=g)
        p[c + ".bias"] = 0.1 * torch.randn(D, generator=g)
    pos = torch.randn(Tn, D, generator=g)
    lw, lb = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    dy = torch.randn(B, Tn, D, generator=g).bfloat16().float()
    gres = torch.randn(B, Tn, D, generator=g)
    # oracle
    hp = h.clone().requires_grad_(True)
    pp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    posr, lwr, lbr = pos.clone().requires_grad_(True), lw.clone().requires_grad_(True), lb.clone().requires_grad_(True)
    x = O.fddt(hp, st, pp, "", cfg) + posr
    y = O.layer_norm(x, lwr, lbr)
    ((y * dy).sum() + (x * gres).sum()).backward()
    # HIP
    hd, std = dev(h).view(rows, D), dev(st)
    w = [dev(p[c + ".weight"]) for c in CLS]
    b = [dev(p[c + ".bias"]) for c in CLS]
    x_d = torch.empty(rows, D, device="cuda")
    y_bf = torch.empty(rows, D, dtype=torch.bfloat16, device="cuda")
    y_f = torch.empty(rows, D, device="cuda")
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    ops.fddt_ln_fwd(hd, rows, D, mode=ops.MODE_DIAG, stno=std, T=Tn, w=w, b=b, pos=dev(pos), h_out=x_d, ln_w=dev(lw),
                    ln_b=dev(lb), y_bf16=y_bf, y_f32=y_f, mean=mean, rstd=rstd)
    assert maxdiff(x_d.cpu().view_as(x), x.detach()) < 1e-6
    assert maxdiff(y_f.cpu().view_as(y), y.detach()) < 2e-5
    assert maxdiff(y_bf.float().cpu().view_as(y), y.detach()) < 4e-2
    g0 = torch.empty(rows, D, device="cuda")
    dlw, dlb, cs = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    dw = [torch.zeros(D, device="cuda") for _ in CLS]
    db = [torch.zeros(D, device="cuda") for _ in CLS]
    ops.fddt_ln_bwd(hd, rows, D, mode=ops.MODE_DIAG, stno=std, T=Tn, w=w, b=b, pos=dev(pos), ln_w=dev(lw), mean=mean,
                    rstd=rstd, d_y=dev(dy, torch.bfloat16).view(rows, D), g_res=dev(gres).view(rows, D), g_out=g0,
                    dln_w=dlw, dln_b=dlb, dw=dw, db=db, colsum_out=cs)
    assert maxdiff(g0.cpu().view_as(h), hp.grad) < 2e-4
    scale = math.sqrt(rows)
    assert maxdiff(dlw.cpu(), lwr.grad) < 2e-4 * scale
    assert maxdiff(dlb.cpu(), lbr.grad) < 2e-4 * scale
    for i, c in enumerate(CLS):
        assert maxdiff(dw[i].cpu(), pp[c + ".weight"].grad) < 3e-4 * scale, c
        assert maxdiff(db[i].cpu(), pp[c + ".bias"].grad) < 3e-4 * scale, c
    assert maxdiff(cs.cpu(), hp.grad.sum((0, 1))) < 3e-4 * scale


@pytest.mark.parametrize("D,B,Tn", [(512, 3, 101), (768, 2, 250), (1024, 1, 333), (1280, 5, 201)])
def test_wave_per_row_layernorm_and_initial_fddt(ops, D, B, Tn):
    """The wave-per-row row kernels at every instantiated width (D = 256 NC, NC = 2..5; ragged row counts): LayerNorm forward
    (bf16 + fp32 outputs, mean, rstd), LayerNorm backward with residual gradient and its three column sums, and the encoder's
    initial FDDT + positions from bf16 rows (bit-exact against the column-owner arithmetic restated in torch: separate fp32
    multiplies / adds in the reference's order)."""
    g = torch.Generator().manual_seed(D + B)
    rows = B * Tn
    x = torch.randn(rows, D, generator=g) * 2 + 0.3
    lw, lb = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    # ---- LayerNorm forward
    xd = dev(x)
    yb = torch.empty(rows, D, dtype=torch.bfloat16, device="cuda"); yf = torch.empty(rows, D, device="cuda")
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    ops.fddt_ln_fwd(xd, rows, D, mode=ops.MODE_NONE, ln_w=dev(lw), ln_b=dev(lb), y_bf16=yb, y_f32=yf, mean=mean, rstd=rstd)
    xr = x.double()
    mu = xr.mean(1, keepdim=True); var = ((xr - mu) ** 2).mean(1, keepdim=True)
    ref = ((xr - mu) / torch.sqrt(var + 1e-5)) * lw.double() + lb.double()
    assert maxdiff(yf.cpu(), ref) < 2e-5 * float(ref.abs().max())
    assert maxdiff(yb.float().cpu(), ref) < 1e-2 * float(ref.abs().max())
    assert maxdiff(mean.cpu(), mu[:, 0]) < 1e-5 and maxdiff(rstd.cpu() * torch.sqrt(var[:, 0] + 1e-5).float(), torch.ones(rows)) < 1e-5
    # ---- LayerNorm backward (+ residual gradient, column sums)
    dy = torch.randn(rows, D, generator=g).bfloat16()
    gres = torch.randn(rows, D, generator=g)
    xa = x.clone().double().requires_grad_(True); lwa = lw.clone().double().requires_grad_(True); lba = lb.clone().double().requires_grad_(True)
    mu_a = xa.mean(1, keepdim=True); var_a = ((xa - mu_a) ** 2).mean(1, keepdim=True)
    ya = ((xa - mu_a) / torch.sqrt(var_a + 1e-5)) * lwa + lba
    ya.backward(dy.double())
    gref = xa.grad + gres.double()
    g0 = torch.empty(rows, D, device="cuda"); g0b = torch.empty(rows, D, dtype=torch.bfloat16, device="cuda")
    dlw, dlb, cs = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda"), torch.full((D,), 0.25, device="cuda")
    ops.fddt_ln_bwd(xd, rows, D, mode=ops.MODE_NONE, ln_w=dev(lw), mean=mean, rstd=rstd, d_y=dev(dy, torch.bfloat16), g_res=dev(gres),
                    g_out=g0, g_out_bf16=g0b, dln_w=dlw, dln_b=dlb, colsum_out=cs)
    sc = float(gref.abs().max())
    assert maxdiff(g0.cpu(), gref) < 2e-5 * sc
    assert maxdiff(g0b.float().cpu(), gref) < 1e-2 * sc
    assert maxdiff(dlw.cpu(), lwa.grad) < 3e-4 * max(1.0, float(lwa.grad.abs().max()))
    assert maxdiff(dlb.cpu(), lba.grad) < 3e-4 * max(1.0, float(lba.grad.abs().max()))
    assert maxdiff(cs.cpu(), 0.25 + gref.sum(0)) < 3e-4 * max(1.0, float(gref.sum(0).abs().max()))
    # ---- initial FDDT (diag) + positions from bf16 rows: bit-exact
    xb = (torch.randn(rows, D, generator=g)).bfloat16()
    st = torch.softmax(torch.randn(B, 4, Tn, generator=g), 1)
    w = [1 + 0.1 * torch.randn(D, generator=g) for _ in range(4)]
    b = [0.1 * torch.randn(D, generator=g) for _ in range(4)]
    pos = torch.randn(Tn, D, generator=g)
    out = torch.full((rows, D), float("nan"), device="cuda")
    ops.fddt_ln_fwd(dev(xb, torch.bfloat16), rows, D, mode=ops.MODE_DIAG, stno=dev(st), T=Tn, w=[dev(t) for t in w], b=[dev(t) for t in b],
                    pos=dev(pos), h_out=out)
    hf = xb.float().view(B, Tn, D)
    t = [(hf * w[c] + b[c]) * st[:, c, :, None] for c in range(4)]
    refi = (((t[0] + t[1]) + t[2]) + t[3]) + pos[None]
    assert torch.equal(out.cpu().view(B, Tn, D), refi), maxdiff(out.cpu().view(B, Tn, D), refi)


# ------------------------------------------------------------------------------------------------ GEMM
def _bf(x):
    return x.bfloat16().float()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 384, 128), (1500, 1152, 384), (333, 516, 1280), (3000, 5120, 1280),
                                   (16, 1280, 1280), (1, 51968, 1280), (7, 36, 5120)])          # last three: the skinny decode-step kernel
def test_gemm_nt_plain(ops, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    A, B = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g) * K ** -0.5)
    ref = A.double() @ B.double().t()
    Cf = torch.empty(M, N, device="cuda")
    ops.gemm_nt(dev(A, torch.bfloat16), dev(B, torch.bfloat16), Cf, M, N, K)
    assert maxdiff(Cf.cpu(), ref) < 1e-4 * max(1.0, float(ref.abs().max()))
    Cb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(dev(A, torch.bfloat16), dev(B, torch.bfloat16), Cb, M, N, K)
    assert maxdiff(Cb.float().cpu(), ref) < 1e-2 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("M,N,K", [(1024, 512, 512), (1024, 1536, 512), (1024, 2048, 512), (1024, 512, 2048), (1000, 500, 128),
                                   (70, 68, 448)])
def test_gemm_nt64_decoder_shapes(ops, M, N, K):
    """The 64 x 64 deep-ring kernel (few 128 x 128 tiles: the decoder's GEMMs at training time): it is the kernel that runs, its
    result equals the fp64 product to fp32-accumulation accuracy, ragged edges and the fp32-residual epilogue included."""
    g = torch.Generator().manual_seed(M * 7 + N + K)
    A, B = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g) * K ** -0.5)
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = A.double() @ B.double().t()
    Ad, Bd = dev(A, torch.bfloat16), dev(B, torch.bfloat16)
    before = ops.gemm_dispatch_log()
    Cf = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_nt(Ad, Bd, Cf, M, N, K)
    after = ops.gemm_dispatch_log()
    ran = [k for k in after if after[k] != before.get(k, 0)]
    assert len(ran) == 1 and ran[0].startswith("gemm_nt64_kernel"), ran
    assert maxdiff(Cf.cpu(), ref) < 1e-4 * max(1.0, float(ref.abs().max()))
    C3 = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_nt(Ad, Bd, C3, M, N, K, bias=dev(bias), residual=dev(res))
    assert maxdiff(C3.cpu(), _bf((ref + bias).float()) + res) < 3e-2
    # same bits as the 128 x 128 kernel's order of additions is NOT required; determinism is: two runs agree exactly
    C4 = torch.empty(M, N, device="cuda")
    ops.gemm_nt(Ad, Bd, C4, M, N, K)
    assert torch.equal(C4, Cf)


@pytest.mark.parametrize("M,N,K", [(12000, 512, 512), (12000, 512, 2048), (12000, 512, 1536), (12100, 500, 128), (16000, 768, 64)])
def test_gemm_nt_mid_size_shapes(ops, M, N, K):
    """Mid-size problems (whisper-base B = 8: M = 12000 rows against the model width -- too few 256 x 256 tiles for the persistent
    kernel; 128 x 128 tiles, or the 128 x 256 ring tiles of a -DNT128W=1 build): the result equals the fp64 product to
    fp32-accumulation accuracy (ragged rows / columns, a single k-step, the fp32-residual epilogue), and two runs agree exactly."""
    g = torch.Generator().manual_seed(M * 3 + N + K)
    A, B = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g) * K ** -0.5)
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = A.double() @ B.double().t()
    Ad, Bd = dev(A, torch.bfloat16), dev(B, torch.bfloat16)
    before = ops.gemm_dispatch_log()
    Cf = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_nt(Ad, Bd, Cf, M, N, K)
    after = ops.gemm_dispatch_log()
    ran = [k for k in after if after[k] != before.get(k, 0)]
    assert len(ran) == 1 and ran[0] in ("gemm_nt128t_kernel", "gemm_nt128w_kernel", "gemm_nt_kernel<2, false>"), ran
    assert maxdiff(Cf.cpu(), ref) < 1e-4 * max(1.0, float(ref.abs().max()))
    C3 = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_nt(Ad, Bd, C3, M, N, K, bias=dev(bias), residual=dev(res))
    assert maxdiff(C3.cpu(), _bf((ref + bias).float()) + res) < 4e-2       # one bf16 step of the rounded Linear output at |x| < 8
    Cb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(Ad, Bd, Cb, M, N, K)
    assert maxdiff(Cb.float().cpu(), ref) < 1e-2 * max(1.0, float(ref.abs().max()))
    C4 = torch.empty(M, N, device="cuda")
    ops.gemm_nt(Ad, Bd, C4, M, N, K)
    assert torch.equal(C4, Cf)


@pytest.mark.parametrize("M,N,K,Tn", [(6100, 2244, 128, 305), (12200, 1284, 192, 1525), (24000, 1280, 128, 1500)])
def test_gemm_nt_fddt_epilogue_bit_exact(ops, M, N, K, Tn):
    """DICOW_EPI_FDDT: the next layer's diagonal FDDT applied in the fp32-residual epilogue of the persistent GEMM equals, bit for
    bit, the same GEMM without it followed by the FDDT row kernel (ragged rows / columns, tail column block included); problems
    below the persistent kernel's threshold are refused."""
    from ts_asr_whisper_amd import _lib as L
    g = torch.Generator().manual_seed(M + N)
    B = M // Tn
    assert B * Tn == M
    A, Bm = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g) * K ** -0.5)
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    st = torch.softmax(torch.randn(B, 4, Tn, generator=g), 1)
    w = [dev(1 + 0.1 * torch.randn(N, generator=g)) for _ in range(4)]
    b = [dev(0.1 * torch.randn(N, generator=g)) for _ in range(4)]
    Ad, Bd, biasd, resd, std = dev(A, torch.bfloat16), dev(Bm, torch.bfloat16), dev(bias), dev(res), dev(st)
    assert ops.gemm_nt(Ad, Bd, resd, M, N, K, bias=biasd, residual=resd, query_persistent=True)
    rowmask = torch.zeros((M + 191) // 192 * 192 + 64, 4, device="cuda")
    rowmask[:M].view(B, Tn, 4).copy_(std.permute(0, 2, 1))
    C1 = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_nt(Ad, Bd, C1, M, N, K, bias=biasd, residual=resd, fddt=(w, b, rowmask))
    C0 = torch.empty(M, N, device="cuda")
    ops.gemm_nt(Ad, Bd, C0, M, N, K, bias=biasd, residual=resd)
    ref = torch.empty(M, N, device="cuda")
    ops.fddt_ln_fwd(C0, M, N, mode=ops.MODE_DIAG, stno=std, T=Tn, w=w, b=b, h_out=ref)
    assert torch.equal(C1, ref), float((C1 - ref).abs().max())
    with pytest.raises(L.DicowError):
        small = torch.empty(300, N, device="cuda")
        ops.gemm_nt(Ad[:300], Bd, small, 300, N, K, bias=biasd, residual=resd[:300], fddt=(w, b, rowmask))


@pytest.mark.parametrize("R,C,ld,ld_t", [(1280, 1280, None, None), (3840, 1280, None, None), (200, 136, 144, 208), (77, 130, None, None)])
def test_cast_transpose(ops, R, C, ld, ld_t):
    """AMP weight copies: bf16 [R,C] and bf16 [C,R] of an fp32 matrix (vectorised kernel and the scalar fallback)."""
    g = torch.Generator().manual_seed(R + C)
    w = torch.randn(R, C, generator=g)
    out = torch.zeros(R, ld or C, dtype=torch.bfloat16, device="cuda")
    out_t = torch.zeros(C, ld_t or R, dtype=torch.bfloat16, device="cuda")
    ops.cast_transpose_bf16(dev(w), out, out_t, ld=ld, ld_t=ld_t)
    ref = w.to(torch.bfloat16)
    assert torch.equal(out[:, :C].cpu(), ref) and torch.equal(out_t[:, :R].cpu(), ref.t())
    if ld: assert float(out[:, C:].float().abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K", [(300, 256, 192), (6100, 2244, 128), (12200, 1284, 64),     # 128x128 / persistent 256x256 / 192x320
                                   (16, 1284, 1280), (5, 260, 192),                            # skinny (M <= 16)
                                   (12000, 508, 192)])                                         # 128x256 ring tiles (mid-size)
def test_gemm_nt_epilogues(ops, M, N, K):
    from ts_asr_whisper_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    A, B = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g) * K ** -0.5)
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    Ad, Bd = dev(A, torch.bfloat16), dev(B, torch.bfloat16)
    base = A @ B.t() + bias
    nq = (N // 8) * 4
    # bias only (bf16 out)
    C0 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(Ad, Bd, C0, M, N, K, bias=dev(bias))
    assert maxdiff(C0.float().cpu(), base) < 3e-2
    # bias + q-scale on the first nq columns
    C1 = torch.empty(M, N, device="cuda")
    ops.gemm_nt(Ad, Bd, C1, M, N, K, bias=dev(bias), flags=L.EPI_SCALE_N, scale=0.125, scale_ncols=nq)
    ref = base.clone(); ref[:, :nq] *= 0.125
    assert maxdiff(C1.cpu(), ref) < 2e-4
    C1b = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(Ad, Bd, C1b, M, N, K, bias=dev(bias), flags=L.EPI_SCALE_N, scale=0.125, scale_ncols=nq)
    assert maxdiff(C1b.float().cpu(), ref) < 3e-2
    # bias + GELU with saved pre-activation
    C2 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    aux = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(Ad, Bd, C2, M, N, K, bias=dev(bias), aux=aux, flags=L.EPI_GELU)
    assert maxdiff(aux.float().cpu(), base) < 3e-2
    assert maxdiff(C2.float().cpu(), O.gelu_erf(_bf(base))) < 3e-2
    # bias + residual (fp32 out); the Linear output is bf16-rounded before the add (AMP)
    C3 = torch.empty(M, N, device="cuda")
    ops.gemm_nt(Ad, Bd, C3, M, N, K, bias=dev(bias), residual=dev(res))
    assert maxdiff(C3.cpu(), _bf(base) + res) < 3e-2
    # GELU backward from the saved pre-activation: C = acc * gelu'(aux)
    C4 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(Ad, Bd, C4, M, N, K, aux=aux, flags=L.EPI_GELU_BWD)
    u = aux.float().cpu().requires_grad_(True)
    O.gelu_erf(u).sum().backward()
    assert maxdiff(C4.float().cpu(), (A @ B.t()) * u.grad) < 4e-2
    # bias + GELU saving the derivative instead (MLP path), then the one-multiply backward epilogue
    C6 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    daux = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(Ad, Bd, C6, M, N, K, bias=dev(bias), aux=daux, flags=L.EPI_GELU | L.EPI_GELU_DAUX)
    ub = _bf(base).requires_grad_(True)
    O.gelu_erf(ub).sum().backward()
    assert maxdiff(C6.float().cpu(), O.gelu_erf(_bf(base))) < 3e-2
    # the pre-activation itself is only bf16-accurate, so compare gelu' where it is taken at the kernel's own rounding
    assert (daux.float().cpu() - ub.grad).abs().mean() < 2e-3 and maxdiff(daux.float().cpu(), ub.grad) < 3e-2
    C7 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(Ad, Bd, C7, M, N, K, aux=daux, flags=L.EPI_MUL_AUX)
    assert maxdiff(C7.float().cpu(), (A @ B.t()) * daux.float().cpu()) < 4e-2
    # ... and the same with the fused column sum (bias gradient of the producing layer): out += colsum(C)
    C8 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    cs = torch.full((N,), 0.5, device="cuda")
    ops.gemm_nt(Ad, Bd, C8, M, N, K, aux=daux, flags=L.EPI_MUL_AUX, colsum_out=cs)
    assert maxdiff(C8.float().cpu(), C7.float().cpu()) == 0.0
    ref_cs = 0.5 + ((A @ B.t()) * daux.float().cpu()).double().sum(0)
    assert maxdiff(cs.cpu(), ref_cs) < 2e-3 * max(1.0, float(ref_cs.abs().max())) + 0.15 * (M ** 0.5) * 2 ** -8
    # accumulate
    C5 = dev(res.clone())
    ops.gemm_nt(Ad, Bd, C5, M, N, K, flags=L.EPI_ACCUM)
    assert maxdiff(C5.cpu(), res + A @ B.t()) < 2e-4


def test_gemm_nt_cu_limit(ops):
    """dicow_set_gemm_cus: the persistent kernel on fewer workgroups than CUs (CUs left to RCCL) gives the same result."""
    M, N, K = 6100, 2244, 128
    g = torch.Generator().manual_seed(11)
    A, B = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g) * K ** -0.5)
    Ad, Bd = dev(A, torch.bfloat16), dev(B, torch.bfloat16)
    C0 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    C1 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(Ad, Bd, C0, M, N, K)
    prev = ops.set_gemm_cus(100)
    try:
        ops.gemm_nt(Ad, Bd, C1, M, N, K)
    finally:
        ops.set_gemm_cus(prev)
    assert torch.equal(C0.cpu(), C1.cpu())
    assert maxdiff(C0.float().cpu(), A @ B.t()) < 3e-2


@pytest.mark.parametrize("M,N,K,f32", [(1024, 512, 51968, False), (2048, 1280, 51968, False), (200, 384, 8192, True), (1024, 512, 51968 + 64, False)])
def test_gemm_nt_deep_contraction_split(ops, M, N, K, f32):
    """The tied LM head's dgrad shape (modeling_dicow.py:302: d_x = d_logits [B L, Vpad] . E): few output tiles, 812 k-steps.
    ops.gemm_nt hands the library a workspace and the contraction runs as equal ranges of one batched launch, summed in range
    order (dicow_gemm_nt_splitk_ws_bytes); K = 51968 + 64 has 813 = 3 x 271 k-steps (three ranges).  Against fp32 torch."""
    from ts_asr_whisper_amd import _lib as L
    import ctypes as C
    g = torch.Generator(device="cuda").manual_seed(M + K)
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    B = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
    a = L.GemmArgs()
    a.A, a.B, a.C, a.M, a.N, a.K, a.lda, a.ldb, a.ldc, a.batch = A.data_ptr(), B.data_ptr(), out.data_ptr(), M, N, K, K, K, N, 1
    a.flags = L.EPI_OUT_F32 if f32 else 0
    assert L.lib().dicow_gemm_nt_splitk_ws_bytes(C.byref(a)) > 0           # this shape IS split
    ops.gemm_nt(A, B, out, M, N, K)
    ref = A.float() @ B.float().t()
    err = float((out.float() - ref).abs().max())
    assert err < (2e-3 if f32 else 2e-2) * max(1.0, float(ref.abs().max())), err
    again = torch.empty_like(out)
    ops.gemm_nt(A, B, again, M, N, K)
    assert torch.equal(out, again)                                         # fixed summation order


def test_gelu_device_accuracy(ops):
    """The GELU of the epilogues -- x / (1 + 2^(x q(x^2))), q a degree-6 fit of the normal cdf's logit (common.h) -- against
    float64 erf on a dense grid of bf16 inputs in [-9, 9]: |err| < 1e-6 before the bf16 rounding, and the ROUNDED activation
    equals the rounded float64 one wherever |gelu| > 1e-4 (below that the absolute bound is what matters)."""
    from ts_asr_whisper_amd import _lib as L
    # identity GEMM: A = x (bf16) as a [M, 64] block against B = I (64x64) reproduces x in the accumulator
    x = _bf(torch.linspace(-9, 9, 256 * 64).view(256, 64))
    eye = torch.eye(64)
    C = torch.empty(256, 64, device="cuda")
    ops.gemm_nt(dev(x, torch.bfloat16), dev(eye, torch.bfloat16), C, 256, 64, 64, flags=L.EPI_GELU)
    xd = x.double()
    ref = xd * 0.5 * (1 + torch.erf(xd / 2 ** 0.5))
    assert maxdiff(C.cpu().double(), ref) < 1e-6
    big = ref.abs() > 1e-4
    same = (_bf(C.cpu())[big] == _bf(ref.float())[big]).float().mean()
    assert float(same) > 0.999, float(same)


def test_gemm_nt_batched_strided_conv_view(ops):
    """conv1d(k=3, s=2, p=1) as a GEMM over the overlapping time-major view (encoder.py:168)."""
    B, L, Cc, Oc = 2, 200, 128, 128
    g = torch.Generator().manual_seed(9)
    x = _bf(torch.randn(B, Cc, L, generator=g))
    w = _bf(torch.randn(Oc, Cc, 3, generator=g) * (3 * Cc) ** -0.5)
    bias = torch.randn(Oc, generator=g)
    ref = torch.nn.functional.conv1d(x, w, bias, stride=2, padding=1).permute(0, 2, 1)      # [B, L/2, O]
    xt = torch.zeros(B, L + 2, Cc)
    xt[:, 1:L + 1] = x.permute(0, 2, 1)
    wp = ops.conv_weight_pack(dev(w), 3 * Cc)
    out = torch.empty(B, L // 2, Oc, device="cuda")
    ops.gemm_nt(dev(xt, torch.bfloat16), wp, out, L // 2, Oc, 3 * Cc, lda=2 * Cc, bias=dev(bias), batch=B,
                strideA=(L + 2) * Cc, strideC=(L // 2) * Oc)
    assert maxdiff(out.cpu(), ref) < 2e-4


@pytest.mark.parametrize("Mk,N1,N2", [(64, 128, 128), (200, 256, 384), (1500, 384, 1152), (1000, 136, 72)])
def test_gemm_tn(ops, Mk, N1, N2):
    g = torch.Generator().manual_seed(Mk + N1)
    A, B = _bf(torch.randn(Mk, N1, generator=g)), _bf(torch.randn(Mk, N2, generator=g))
    init = torch.randn(N1, N2, generator=g)
    Cd = dev(init.clone())
    ops.gemm_tn(dev(A, torch.bfloat16), dev(B, torch.bfloat16), Cd, Mk, N1, N2, accumulate=True)
    ref = init.double() + A.double().t() @ B.double()
    assert maxdiff(Cd.cpu(), ref) < 2e-4 * max(1.0, float(ref.abs().max()))


# ------------------------------------------------------------------------------------------------ attention forward
LN2 = 0.6931471805599453


def _attn_ref(q, k, v, causal, q_log2=False):
    # q_log2: q carries a factor log2(e) -- the scores are base-2 exponents, i.e. natural-log scores are (q . k) ln 2
    s = torch.einsum("blhd,bmhd->bhlm", q.double(), k.double()) * (LN2 if q_log2 else 1.0)
    if causal:
        Lq, Lk = s.shape[-2:]
        s = s.masked_fill(~torch.ones(Lq, Lk, dtype=torch.bool).tril(), float("-inf"))
    p = torch.softmax(s, -1)
    return torch.einsum("bhlm,bmhd->blhd", p, v.double()), torch.logsumexp(s, -1)


@pytest.mark.parametrize("B,H,Lq,Lk,causal", [(1, 2, 128, 64, False), (2, 3, 100, 100, False), (1, 2, 1500, 1500, False),
                                             (2, 2, 77, 77, True), (1, 2, 300, 300, True), (2, 2, 50, 1500, False),
                                             (3, 4, 140, 140, True)])      # 12 (batch, head) pairs: one XCD group of 8 + a remainder
@pytest.mark.parametrize("q_log2", [False, True])
def test_attn_fwd(ops, B, H, Lq, Lk, causal, q_log2):
    g = torch.Generator().manual_seed(B * 1000 + Lq + Lk)
    D = H * 64
    # packed projection buffers with row stride 3D (q | k | v), as the model uses them
    qkv_q = _bf(torch.randn(B, Lq, 3 * D, generator=g) * 0.6)
    qkv_k = _bf(torch.randn(B, Lk, 3 * D, generator=g) * 0.6)
    q = qkv_q[:, :, :D].view(B, Lq, H, 64)
    k = qkv_k[:, :, D:2 * D].view(B, Lk, H, 64)
    v = qkv_k[:, :, 2 * D:].view(B, Lk, H, 64)
    ref_o, ref_lse = _attn_ref(q, k, v, causal, q_log2)
    dq, dk = dev(qkv_q, torch.bfloat16), dev(qkv_k, torch.bfloat16)
    qd = dq[:, :, :D].view(B, Lq, H, 64)
    kd = dk[:, :, D:2 * D].view(B, Lk, H, 64)
    vd = dk[:, :, 2 * D:].view(B, Lk, H, 64)
    o = torch.zeros(B, Lq, H, 64, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B, H, Lq, device="cuda")
    ops.attn_fwd(qd, kd, vd, o, lse, causal=causal, q_log2=q_log2)
    assert maxdiff(lse.cpu(), ref_lse) < 2e-3
    assert maxdiff(o.float().cpu(), ref_o) < 2e-2


@pytest.mark.parametrize("causal", [False, True])
def test_attn_fwd_log2_reference_exponent_moves(ops, causal):
    """q_log2 forward: p = 2^s against a reference exponent that starts at ZERO and only moves when a row maximum leaves
    [-60, 60].  Rows are built to leave it in every way: scores far ABOVE the window from the first tile on (row block 0), far
    BELOW it everywhere (block 1: without the move every probability would underflow to 0), inside it first and far above it
    from the third key tile on (block 2), and ordinary rows (the rest) sharing waves and workgroups with them."""
    B, H, L = 1, 2, 320
    g = torch.Generator().manual_seed(5)
    q = torch.randn(B, L, H, 64, generator=g) * 0.3
    k = torch.randn(B, L, H, 64, generator=g) * 0.3
    v = torch.randn(B, L, H, 64, generator=g)
    u = torch.nn.functional.normalize(torch.randn(64, generator=g), dim=0)
    k = k + 3.0 * u                                              # every key has a common component along u ...
    q[:, 0:32] += 40.0 * u                                       # ... so these rows score ~ +120 (base-2 units) on every key
    q[:, 32:64] -= 50.0 * u                                      # ~ -150 on every key
    k[:, 150:] += 40.0 * u                                       # keys of the later tiles: much larger along u
    q[:, 64:96] += 2.0 * u                                       # rows that start inside the window and meet ~ +90 from tile 2 on
    q, k, v = _bf(q), _bf(k), _bf(v)
    ref_o, ref_lse = _attn_ref(q, k, v, causal, True)
    o = torch.zeros(B, L, H, 64, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B, H, L, device="cuda")
    ops.attn_fwd(dev(q, torch.bfloat16), dev(k, torch.bfloat16), dev(v, torch.bfloat16), o, lse, causal=causal, q_log2=True)
    s2 = torch.einsum("blhd,bmhd->bhlm", q.double(), k.double())          # base-2 scores
    assert float(s2[:, :, 0:32].min()) > 70 and float(s2[:, :, 32:64].max()) < -70 and float(s2[:, :, 64:96, :128].max()) < 60 < float(s2[:, :, 64:96].max())
    assert torch.isfinite(lse).all() and torch.isfinite(o.float()).all()
    assert maxdiff(lse.cpu(), ref_lse) < 2e-3 * max(1.0, float(ref_lse.abs().max()) / 60)
    assert maxdiff(o.float().cpu(), ref_o) < 2e-2


@pytest.mark.parametrize("B,H,Lq,Lk,causal", [(1, 2, 128, 64, False), (2, 3, 100, 100, False), (1, 2, 1500, 1500, False),
                                             (2, 2, 77, 77, True), (1, 2, 300, 300, True), (2, 2, 50, 1500, False),
                                             (1, 1, 200, 130, False), (3, 4, 140, 140, False), (2, 3, 300, 260, False),
                                             (1, 4, 520, 700, False)])
@pytest.mark.parametrize("q_log2", [False, True])
@pytest.mark.parametrize("fused", [False, "force"])
def test_attn_bwd(ops, B, H, Lq, Lk, causal, q_log2, fused):
    """fused = "force": the one-kernel (5-pass) backward on every dense case, whatever its size -- one key block (first = last
    visitor of the dQ tiles), ragged last key block / query tile, an all-padding second query block, Lq != Lk."""
    if fused and causal:
        pytest.skip("the fused backward is dense-only")
    g = torch.Generator().manual_seed(B * 999 + Lq + 3 * Lk)
    D = H * 64
    qkv_q = _bf(torch.randn(B, Lq, 3 * D, generator=g) * 0.6)
    qkv_k = _bf(torch.randn(B, Lk, 3 * D, generator=g) * 0.6)
    d_o = _bf(torch.randn(B, Lq, H, 64, generator=g))
    q = qkv_q[:, :, :D].view(B, Lq, H, 64).clone().requires_grad_(True)
    k = qkv_k[:, :, D:2 * D].view(B, Lk, H, 64).clone().requires_grad_(True)
    v = qkv_k[:, :, 2 * D:].view(B, Lk, H, 64).clone().requires_grad_(True)
    s = torch.einsum("blhd,bmhd->bhlm", q.double(), k.double()) * (LN2 if q_log2 else 1.0)
    if causal:
        s = s.masked_fill(~torch.ones(Lq, Lk, dtype=torch.bool).tril(), float("-inf"))
    o_ref = torch.einsum("bhlm,bmhd->blhd", torch.softmax(s, -1), v.double())
    (o_ref * d_o.double()).sum().backward()
    # the kernel's dq is dq_scale * dS . k whatever the units of q (the gradient of the projection output BEFORE any scaling
    # when dq_scale = head_dim^-0.5); autograd through the ln 2 above returns ln 2 * dS . k for the stored (log2-scaled) q
    q_grad = q.grad / LN2 if q_log2 else q.grad
    dq_, dk_ = dev(qkv_q, torch.bfloat16), dev(qkv_k, torch.bfloat16)
    qd = dq_[:, :, :D].view(B, Lq, H, 64)
    kd = dk_[:, :, D:2 * D].view(B, Lk, H, 64)
    vd = dk_[:, :, 2 * D:].view(B, Lk, H, 64)
    o = torch.zeros(B, Lq, H, 64, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B, H, Lq, device="cuda")
    ops.attn_fwd(qd, kd, vd, o, lse, causal=causal, q_log2=q_log2)
    gq = torch.zeros(B, Lq, 3 * D, dtype=torch.bfloat16, device="cuda")
    gk = torch.zeros(B, Lk, 3 * D, dtype=torch.bfloat16, device="cuda")
    delta = torch.empty(2, B, H, Lq, device="cuda")
    csq, csv = torch.full((D,), 0.25, device="cuda"), torch.full((D,), -0.5, device="cuda")
    ops.attn_bwd(qd, kd, vd, o, dev(d_o, torch.bfloat16), lse, delta,
                 gq[:, :, :D].view(B, Lq, H, 64), gk[:, :, D:2 * D].view(B, Lk, H, 64), gk[:, :, 2 * D:].view(B, Lk, H, 64),
                 causal=causal, dq_scale=0.5, dq_colsum=csq, dv_colsum=csv, q_log2=q_log2, fused=fused)
    if fused:
        assert ops.attn_bwd_fused_status() == 0
    tol = lambda ref: 2e-2 * max(1.0, float(ref.abs().max()))
    # fused bias gradients: column sums of exactly the values that were stored (accumulated onto the initial contents)
    assert maxdiff(csq.cpu(), 0.25 + gq[:, :, :D].float().cpu().double().sum((0, 1))) < 1e-3 * (1 + B * Lq) ** 0.5
    assert maxdiff(csv.cpu(), -0.5 + gk[:, :, 2 * D:].float().cpu().double().sum((0, 1))) < 1e-3 * (1 + B * Lk) ** 0.5
    assert maxdiff(gq[:, :, :D].float().cpu().view(B, Lq, H, 64), 0.5 * q_grad) < tol(q_grad)
    assert maxdiff(gk[:, :, D:2 * D].float().cpu().view(B, Lk, H, 64), k.grad) < tol(k.grad)
    assert maxdiff(gk[:, :, 2 * D:].float().cpu().view(B, Lk, H, 64), v.grad) < tol(v.grad)


# ------------------------------------------------------------------------------------------------ log-mel (golden F2)
def test_logmel_vs_reference_golden(ops):
    import numpy as np
    from ts_asr_whisper_amd import features
    z = load_golden("f2_logmel")
    for i in range(int(z["n_cases"])):
        wave = z[f"wave_{i}"].astype(np.float32) / 32768.0
        padded, am = features.pad_to_30s([wave])
        got = features.log_mel(padded.cuda(), int(z[f"mels_{i}"])).cpu()[0]
        ref = T(z, f"feat_{i}")
        assert got.shape == ref.shape
        assert int(am.sum()) == int(z[f"attn_sum_{i}"])
        # reference: fp32 torch.stft; HF quotes 1e-5 between its own numpy/torch paths
        assert maxdiff(got, ref) < 3e-4, maxdiff(got, ref)
    # batch of two clips, second longer than 30 s -> 60 s padding
    w2 = [np.random.default_rng(0).standard_normal(16000 * 31).astype(np.float32) * 0.1, wave]
    padded, am = features.pad_to_30s(w2)
    out = features.log_mel(padded.cuda(), 80)
    assert out.shape == (2, 80, 6000) and torch.isfinite(out).all()


@pytest.mark.parametrize("B,H,Lk", [(2, 3, 1), (16, 20, 448), (3, 2, 1500)])
def test_attn_fwd_single_query_row(ops, B, H, Lk):
    """Lq = 1: the decoding step's attention over a KV cache (strided cache views, batch stride > Lk rows)."""
    g = torch.Generator().manual_seed(Lk)
    D, Lmax = H * 64, Lk + 7
    q = (torch.randn(B, 1, H, 64, generator=g) * 0.3).bfloat16()
    kc = torch.randn(B, Lmax, D, generator=g).bfloat16()
    vc = torch.randn(B, Lmax, D, generator=g).bfloat16()
    qd, kd, vd = q.cuda(), kc.cuda(), vc.cuda()
    o = torch.empty(B, 1, H, 64, dtype=torch.bfloat16, device="cuda")
    ops.attn_fwd(qd, kd[:, :Lk].view(B, Lk, H, 64), vd[:, :Lk].view(B, Lk, H, 64), o)
    k, v = kc[:, :Lk].view(B, Lk, H, 64).float(), vc[:, :Lk].view(B, Lk, H, 64).float()
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k)
    ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), v)
    assert maxdiff(o.float().cpu(), ref) < 2e-2
