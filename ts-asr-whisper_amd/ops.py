"""Tensor-level wrappers over the C ABI (no autograd here; see ``functional.py`` / ``modeling.py``).

Every wrapper validates device/dtype/contiguity, allocates outputs with torch (the library never
allocates) and enqueues the HIP kernels on the current torch stream.
"""
import ctypes as C
import os
from typing import Optional, Sequence

import torch

from . import _lib as L

BF16 = torch.bfloat16
F32 = torch.float32


def _req(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise L.DicowError(f"{name}: tensor must live on the GPU (no CPU fallback)")
    if t.dtype != dtype:
        raise L.DicowError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


def _p(t):
    return None if t is None else t.data_ptr()


_WS = {}


def workspace(nbytes: int, device) -> torch.Tensor:
    """Persistent per-device scratch for the two-stage reductions (the library never allocates).  Kernels that use
    it are stream-ordered on the current stream, so one buffer per device suffices."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    t = _WS.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device)
        _WS[key] = t
    return t


def gemm_dispatch_log() -> dict:
    """{kernel instantiation name (as rocprofv3 prints it): launches so far in this process} (dicow_gemm_dispatch_log)."""
    n = L.lib().dicow_gemm_dispatch_log(None, 0)
    buf = C.create_string_buffer(n + 64)
    L.lib().dicow_gemm_dispatch_log(buf, n + 64)
    out = {}
    for line in buf.value.decode().splitlines():
        k, _, c = line.partition("\t")
        out[k] = int(c)
    return out


_NCU = {}


def num_cus(device=None) -> int:
    """CUs the persistent GEMM grids may use on this device (all of them unless set_gemm_cus limited it)."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if dev not in _NCU:
        _NCU[dev] = torch.cuda.get_device_properties(dev).multi_processor_count
    lim = L.lib().dicow_set_gemm_cus(0)
    L.lib().dicow_set_gemm_cus(lim)
    return lim if 0 < lim < _NCU[dev] else _NCU[dev]


def set_gemm_cus(n: int) -> int:
    """CUs the persistent NT GEMM may occupy (0 = all); returns the previous setting (dicow_set_gemm_cus)."""
    return L.lib().dicow_set_gemm_cus(int(n))


def _arr4(ts):
    a = (C.c_void_p * 4)()
    for i, t in enumerate(ts):
        a[i] = None if t is None else t.data_ptr()
    return a


# ------------------------------------------------------------------------------------------------ casts / packs
def cast_bf16(src: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(src, F32, "cast_bf16.src")
    src = src.contiguous()
    if out is None:
        out = torch.empty(src.shape, dtype=BF16, device=src.device)
    L.call("dicow_cast_f32_to_bf16", src.data_ptr(), out.data_ptr(), src.numel(), L.stream())
    return out


_CAST_GROUP = None          # inside `with cast_group():` the casts are recorded and launched pooled (dicow_cast_transpose_group)


class cast_group:
    """Context: cast_transpose_bf16 calls inside are recorded (their tensors kept alive) and run as pooled launches of up to
    CAST_GROUP_MAX matrices when the context exits -- a layer's weight re-cast in one launch instead of six."""

    def __enter__(self):
        global _CAST_GROUP
        self.prev, self.items = _CAST_GROUP, []
        _CAST_GROUP = self
        return self

    def flush(self):
        while self.items:
            chunk, self.items = self.items[:L.CAST_GROUP_MAX], self.items[L.CAST_GROUP_MAX:]
            arr = (L.CastProblem * len(chunk))()
            for i, (src, out, out_t, ld, ld_t, R, Cc) in enumerate(chunk):
                arr[i].src, arr[i].dst, arr[i].dst_t = src.data_ptr(), _p(out), _p(out_t)
                arr[i].R, arr[i].C, arr[i].ld, arr[i].ld_t = R, Cc, ld, ld_t
            L.call("dicow_cast_transpose_group", arr, len(chunk), L.stream())

    def __exit__(self, *exc):
        global _CAST_GROUP
        _CAST_GROUP = self.prev
        if exc[0] is None:
            self.flush()
        return False


def cast_transpose_bf16(src: torch.Tensor, out=None, out_t=None, ld=None, ld_t=None):
    """fp32 [R,C] -> bf16 `out` [R,C] (row stride ld) and/or `out_t` [C,R] (row stride ld_t); either may be None."""
    _req(src, F32, "cast_transpose.src")
    assert src.dim() == 2 and src.is_contiguous()
    R, Cc = src.shape
    ld, ld_t = (Cc if ld is None else ld), (R if ld_t is None else ld_t)
    if _CAST_GROUP is not None:
        _CAST_GROUP.items.append((src, out, out_t, ld, ld_t, R, Cc))
        return out, out_t
    L.call("dicow_cast_transpose_f32_to_bf16", src.data_ptr(), _p(out), ld, _p(out_t), ld_t, R, Cc, L.stream())
    return out, out_t


def conv_weight_pack(w: torch.Tensor, kpad: int, out=None, out_t=None, want_t=False):
    _req(w, F32, "conv_weight_pack.w")
    O, Cc, three = w.shape
    assert three == 3 and w.is_contiguous()
    if out is None:
        out = torch.empty(O, kpad, dtype=BF16, device=w.device)
    if want_t and out_t is None:
        out_t = torch.empty(kpad, O, dtype=BF16, device=w.device)
    L.call("dicow_conv_weight_pack", w.data_ptr(), out.data_ptr(), _p(out_t), O, Cc, kpad, L.stream())
    return (out, out_t) if want_t else out


def conv_weight_unpack_grad(g_packed: torch.Tensor, g_w: torch.Tensor):
    O, Cc, _ = g_w.shape
    L.call("dicow_conv_weight_unpack_grad", g_packed.data_ptr(), g_w.data_ptr(), O, Cc, g_packed.shape[1], L.stream())


def mel_to_timemajor(mel: torch.Tensor, out=None) -> torch.Tensor:
    _req(mel, F32, "mel_to_timemajor.mel")
    mel = mel.contiguous()
    B, M, Tin = mel.shape
    if out is None:
        out = torch.empty(B, Tin + 2, M, dtype=BF16, device=mel.device)
    L.call("dicow_mel_to_timemajor", mel.data_ptr(), out.data_ptr(), B, M, Tin, L.stream())
    return out


def colsum_bf16(x: torch.Tensor, out: torch.Tensor):
    """out[N] (fp32) += column sums of bf16 x [rows, N] (row stride x.stride(0))."""
    _req(x, BF16, "colsum.x")
    assert x.dim() == 2 and x.stride(1) == 1
    ws = workspace(L.lib().dicow_colsum_ws_bytes(x.shape[0], x.shape[1]), x.device)
    L.call("dicow_colsum_bf16", x.data_ptr(), x.stride(0), out.data_ptr(), x.shape[0], x.shape[1], ws.data_ptr(), ws.numel(),
           L.stream())


def sum_over_batch(g: torch.Tensor, out: torch.Tensor):
    B = g.shape[0]
    L.call("dicow_sum_over_batch", g.data_ptr(), out.data_ptr(), B, g.numel() // B, L.stream())


# ------------------------------------------------------------------------------------------------ FDDT + LayerNorm
MODE_NONE, MODE_DIAG, MODE_BIAS = 0, 1, 2


def fddt_ln_fwd(h_in, rows, D, *, mode=MODE_NONE, stno=None, stno_bstride=None, T=0, w=(None,) * 4, b=(None,) * 4,
                pos=None, h_out=None, ln_w=None, ln_b=None, y_bf16=None, y_f32=None, mean=None, rstd=None, eps=1e-5):
    a = L.FddtLnFwdArgs()
    a.h_in, a.in_bf16, a.mode = h_in.data_ptr(), int(h_in.dtype == BF16), mode
    a.stno = _p(stno)
    a.stno_bstride = (4 * T) if stno_bstride is None else stno_bstride
    a.w, a.b = _arr4(w), _arr4(b)
    a.pos, a.h_out, a.ln_w, a.ln_b = _p(pos), _p(h_out), _p(ln_w), _p(ln_b)
    a.y_bf16, a.y_f32, a.mean, a.rstd = _p(y_bf16), _p(y_f32), _p(mean), _p(rstd)
    a.rows, a.T, a.D, a.eps = rows, T, D, eps
    L.call_struct("dicow_fddt_ln_fwd", a)


def fddt_ln_bwd(h_in, rows, D, *, mode=MODE_NONE, stno=None, stno_bstride=None, T=0, w=(None,) * 4, b=(None,) * 4,
                pos=None, ln_w=None, mean=None, rstd=None, d_y=None, g_res=None, g_out=None, g_out_bf16=None,
                dln_w=None, dln_b=None, dw=(None,) * 4, db=(None,) * 4, colsum_out=None):
    a = L.FddtLnBwdArgs()
    a.h_in, a.in_bf16, a.mode = h_in.data_ptr(), int(h_in.dtype == BF16), mode
    a.stno = _p(stno)
    a.stno_bstride = (4 * T) if stno_bstride is None else stno_bstride
    a.w, a.b = _arr4(w), _arr4(b)
    a.pos, a.ln_w, a.mean, a.rstd = _p(pos), _p(ln_w), _p(mean), _p(rstd)
    a.d_y = _p(d_y)
    a.dy_f32 = int(d_y is not None and d_y.dtype == F32)
    a.g_res, a.g_out, a.g_out_bf16 = _p(g_res), _p(g_out), _p(g_out_bf16)
    a.dln_w, a.dln_b = _p(dln_w), _p(dln_b)
    a.dw, a.db = _arr4(dw), _arr4(db)
    a.colsum_out = _p(colsum_out)
    a.rows, a.T, a.D = rows, T, D
    ws = workspace(L.lib().dicow_fddt_ln_bwd_ws_bytes(rows, D), h_in.device)
    a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
    L.call_struct("dicow_fddt_ln_bwd", a)


# ------------------------------------------------------------------------------------------------ GEMM
def gemm_nt(A, B, C_out, M, N, K, *, lda=None, ldb=None, ldc=None, bias=None, residual=None, ldr=None, aux=None,
            ldaux=None, flags=0, scale=1.0, scale_ncols=0, batch=1, strideA=0, strideB=0, strideC=0, strideAux=0, colsum_out=None,
            fddt=None, query_persistent=False, ln_stat=None, ln_fold=None, query_lnstat=False):
    """C[M,N] = epilogue(A[M,K] @ B[N,K]^T).  Pointers + leading dimensions; see include/dicow_hip.h.
    colsum_out [N] fp32: += column sums of the result (a bias gradient), fused into the epilogue.
    fddt = (w[4], b[4], rowmask): DICOW_EPI_FDDT, the next layer's diagonal FDDT applied to the fp32 result (persistent kernel only).
    query_persistent: launch nothing, return dicow_gemm_nt_is_persistent for this problem.
    LayerNorm fold (include/dicow_hip.h, DICOW_EPI_LNSTAT / LNFOLD):
      ln_stat = (hb, stats): the PRODUCER of the residual stream also stores the bf16 copy hb [M, N] (through aux) and the row
                partials stats [M, 16, 2] fp32 (query_lnstat: launch nothing, return dicow_gemm_nt_lnstat_ok);
      ln_fold = (stats, c, width, eps): the CONSUMER -- A = the producer's hb, B = the folded weight, bias = the folded bias -- scales
                rows by rstd, subtracts rstd * mean * c[n]."""
    a = L.GemmArgs()
    if ln_stat is not None:
        hb, stats = ln_stat
        assert aux is None and stats.dtype == F32 and stats.numel() >= M * 2 * L.LN_SLOTS
        aux, ldaux = hb, (hb.stride(0) if ldaux is None else ldaux)
        a.lnstat = stats.data_ptr()
        flags |= L.EPI_LNSTAT
    if ln_fold is not None:
        stats, lc, width, eps = ln_fold
        assert stats.dtype == F32 and stats.numel() >= M * 2 * L.LN_SLOTS and lc.dtype == F32 and lc.numel() >= N and width % 320 == 0
        a.lnstat, a.ln_c, a.ln_inv_dim, a.ln_eps, a.ln_nslots = stats.data_ptr(), lc.data_ptr(), 1.0 / width, eps, 4 * (width // 320)
        flags |= L.EPI_LNFOLD
    if fddt is not None:
        fw, fb, rowmask = fddt
        a.fddt_w, a.fddt_b, a.fddt_rowmask = _arr4(fw), _arr4(fb), rowmask.data_ptr()
        flags |= L.EPI_FDDT
    a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), C_out.data_ptr()
    a.bias, a.residual, a.aux = _p(bias), _p(residual), _p(aux)
    a.M, a.N, a.K = M, N, K
    a.lda = K if lda is None else lda
    a.ldb = K if ldb is None else ldb
    a.ldc = N if ldc is None else ldc
    a.ldr = N if ldr is None else ldr
    a.ldaux = N if ldaux is None else ldaux
    a.batch, a.strideA, a.strideB, a.strideC, a.strideAux = batch, strideA, strideB, strideC, strideAux
    if C_out.dtype == F32:
        flags |= L.EPI_OUT_F32
    if bias is not None:
        flags |= L.EPI_BIAS
    if residual is not None:
        flags |= L.EPI_RESIDUAL
    if colsum_out is not None:
        flags |= L.EPI_COLSUM
        ws = workspace(L.lib().dicow_gemm_nt_colsum_ws_bytes(M, N), C_out.device)
        a.colsum_out, a.colsum_ws, a.colsum_ws_bytes = colsum_out.data_ptr(), ws.data_ptr(), ws.numel()
    a.flags, a.scale, a.scale_ncols = flags, scale, scale_ncols
    if query_persistent:
        return bool(L.lib().dicow_gemm_nt_is_persistent(C.byref(a)))
    if query_lnstat:
        if not L.has_experimental():
            raise L.DicowError("the LayerNorm fold is experimental: build ts-asr-whisper_amd/csrc/build.sh --exp and load it with DICOW_HIP_LIB=.../libdicow_hip_exp.so")
        return bool(L.lib().dicow_gemm_nt_lnstat_ok(C.byref(a)))
    if K >= 8192 and colsum_out is None:             # deep contraction, small output: split ranges + ordered sum (LM-head dgrad)
        need = L.lib().dicow_gemm_nt_splitk_ws_bytes(C.byref(a))
        if need:
            ws = workspace(need, C_out.device)
            a.colsum_ws, a.colsum_ws_bytes = ws.data_ptr(), ws.numel()
    L.call_struct("dicow_gemm_nt", a)


def lnfold_prep(W, gamma, beta, bias, out_w, out_c, out_b):
    """Weights of a Linear behind a LayerNorm, folded: out_w bf16 [N, K] = bf16(gamma * W), out_c [N] = its row sums (fp32),
    out_b [N] = bias + W_bf16 @ beta.  W fp32 [N, K] contiguous; bias may be None."""
    if not L.has_experimental():
        raise L.DicowError("dicow_lnfold_prep is experimental: build ts-asr-whisper_amd/csrc/build.sh --exp and load it with "
                           "DICOW_HIP_LIB=.../libdicow_hip_exp.so")
    N, K = W.shape
    assert W.is_contiguous() and W.dtype == F32 and out_w.dtype == torch.bfloat16 and out_w.stride(1) == 1
    L.call("dicow_lnfold_prep", W.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _p(bias), out_w.data_ptr(), out_w.stride(0),
           out_c.data_ptr(), out_b.data_ptr(), N, K, L.stream())


def gemm_tn(A, B, C_out, Mk, N1, N2, *, lda=None, ldb=None, ldc=None, batch=1, strideA=0, strideB=0, accumulate=True,
            C_seg=None, seg_rows=0):
    """C[N1,N2] (+)= sum_m A[m,N1] * B[m,N2]  (fp32 C)."""
    a = L.GemmTnArgs()
    a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), C_out.data_ptr()
    a.Mk, a.N1, a.N2 = Mk, N1, N2
    a.lda = N1 if lda is None else lda
    a.ldb = N2 if ldb is None else ldb
    a.ldc = N2 if ldc is None else ldc
    a.batch, a.strideA, a.strideB, a.accumulate = batch, strideA, strideB, int(accumulate)
    if C_seg is not None:
        for i, t in enumerate(C_seg):
            a.C_seg[i] = t.data_ptr()
        a.seg_rows = seg_rows
    need = L.lib().dicow_gemm_tn_ws_bytes(C.byref(a))
    if need:
        ws = workspace(need, C_out.device)
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
    L.call_struct("dicow_gemm_tn", a)


def _tn_fill(a, A, B, C_out, Mk, N1, N2, lda, ldb, ldc, accumulate, C_seg, seg_rows):
    a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), C_out.data_ptr()
    a.Mk, a.N1, a.N2 = Mk, N1, N2
    a.lda = N1 if lda is None else lda
    a.ldb = N2 if ldb is None else ldb
    a.ldc = N2 if ldc is None else ldc
    a.batch, a.strideA, a.strideB, a.accumulate = 1, 0, 0, int(accumulate)
    if C_seg is not None:
        for i, t in enumerate(C_seg):
            a.C_seg[i] = t.data_ptr()
        a.seg_rows = seg_rows


class TnGroup:
    """Weight gradients C_i (+)= A_i^T B_i collected over a layer's backward and run as ONE pooled launch
    (dicow_gemm_tn_group): ``add`` records a problem (and keeps its operands alive), ``run`` launches what was recorded."""

    def __init__(self):
        self.items = []

    def add(self, A, B, C_out, Mk, N1, N2, *, lda=None, ldb=None, ldc=None, accumulate=True, C_seg=None, seg_rows=0):
        self.items.append((A, B, C_out, Mk, N1, N2, lda, ldb, ldc, accumulate, C_seg, seg_rows))
        if len(self.items) == L.TN_GROUP_MAX:
            self.run()

    def tiles(self):
        """256 x 256 output tiles recorded so far (the pooled launch wants at least one per CU)."""
        return sum(((it[4] + 255) // 256) * ((it[5] + 255) // 256) for it in self.items)

    def run(self):
        if not self.items:
            return
        g = L.GemmTnGroupArgs()
        g.n = len(self.items)
        for i, it in enumerate(self.items):
            _tn_fill(g.p[i], *it)
        need = L.lib().dicow_gemm_tn_group_ws_bytes(C.byref(g))
        if need:
            ws = workspace(need, self.items[0][2].device)
            g.ws, g.ws_bytes = ws.data_ptr(), ws.numel()
        L.call_struct("dicow_gemm_tn_group", g)
        self.items = []


# ------------------------------------------------------------------------------------------------ speaker-communication block glue
def scb_split(hf, q_in, kv_in, cat, Bp, T, D):
    """hf fp32 [Bp*2*T, D] (interleaved) -> q_in / kv_in bf16 [Bp*T, D]; cat[:, D:2D] = q_in  (cat bf16 [Bp*T, 2D])."""
    L.call("dicow_scb_split", hf.data_ptr(), q_in.data_ptr(), kv_in.data_ptr(), cat.data_ptr(), cat.stride(0), Bp, T, D, L.stream())


def scb_merge_fwd(hf, upd, gate, out, Bp, T, D):
    """out = hf; out[mixture rows] += tanh(gate) * upd."""
    L.call("dicow_scb_merge_fwd", hf.data_ptr(), upd.data_ptr(), gate.data_ptr(), out.data_ptr(), Bp, T, D, L.stream())


def scb_gate_bwd(g, upd, gate, d_upd, d_gate, Bp, T, D):
    """d_upd = bf16(g[mixture] * tanh(gate)); d_gate += (1 - tanh^2) * sum(g[mixture] * upd)  (d_gate None: skipped)."""
    ws = workspace(L.lib().dicow_scb_gate_bwd_ws_bytes(), g.device)
    L.call("dicow_scb_gate_bwd", g.data_ptr(), upd.data_ptr(), gate.data_ptr(), d_upd.data_ptr(), _p(d_gate), Bp, T, D,
           ws.data_ptr(), ws.numel(), L.stream())


def scb_merge_bwd(g, d_qin, d_cat, d_kvin, gin, Bp, T, D):
    """gin = g; gin[mixture] += d_qin + d_cat[:, D:]; gin[enrollment] += d_kvin."""
    L.call("dicow_scb_merge_bwd", g.data_ptr(), d_qin.data_ptr(), d_cat.data_ptr(), d_cat.stride(0), d_kvin.data_ptr(), gin.data_ptr(),
           Bp, T, D, L.stream())


# ------------------------------------------------------------------------------------------------ attention
def _bs_rs(t, name):
    # t: [B, L, H, 64] view (any batch/row stride)
    assert t.dim() == 4 and t.shape[3] == 64 and t.stride(3) == 1 and t.stride(2) == 64, f"{name}: need [B,L,H,64] view"
    return t.stride(0), t.stride(1)


LOG2E = 1.4426950408889634          # q-scale factor of the q_log2 attention mode: head_dim^-0.5 * LOG2E in the projection epilogue


def attn_fwd(q, k, v, o, lse=None, causal=False, q_log2=False):
    """q [B,Lq,H,64], k/v [B,Lk,H,64] bf16 views; o like q; lse [B,H,Lq] fp32.  q_log2: q carries a factor log2(e)."""
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (o, "o")):
        _req(t, BF16, "attn_fwd." + n)
    a = L.AttnFwdArgs()
    a.q, a.k, a.v, a.o, a.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), _p(lse)
    a.q_bs, a.q_rs = _bs_rs(q, "q")
    a.k_bs, a.k_rs = _bs_rs(k, "k")
    a.v_bs, a.v_rs = _bs_rs(v, "v")
    a.o_bs, a.o_rs = _bs_rs(o, "o")
    a.B, a.Lq, a.H = q.shape[0], q.shape[1], q.shape[2]
    a.Lk, a.causal, a.q_log2 = k.shape[1], int(causal), int(q_log2)
    L.call_struct("dicow_attn_fwd", a)


_FWS = {}
ATTN_BWD_FUSED = os.environ.get("DICOW_ATTN_BWD_FUSED", "1") != "0"      # A/B switch of the fused (5-pass) attention backward


def attn_bwd_fused_ws(B, H, Lq, Lk, device):
    """Persistent per-(device, stream) workspace of the fused attention backward (the dQ tiles that travel between the key-block
    workgroups through L2, their flags, a status header).  Stream-ordered reuse, like `workspace`."""
    n = L.lib().dicow_attn_bwd_fused_ws_bytes(B, H, Lq, Lk)
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    t = _FWS.get(key)
    if t is None or t.numel() < n:
        t = torch.zeros(n + 4096, dtype=torch.uint8, device=device)
        _FWS[key] = t
    off = (-t.data_ptr()) % 4096
    return t[off:off + n]


def attn_bwd_fused_status(device=None) -> int:
    """Error bits of the fused attention backward's last launches on this device's workspaces (synchronises).  0 = fine."""
    torch.cuda.synchronize(device)
    st = 0
    for (_, idx, _), t in _FWS.items():
        if device is None or idx == torch.device(device).index:
            off = (-t.data_ptr()) % 4096
            st |= L.lib().dicow_attn_bwd_fused_status(t.data_ptr() + off)
    return st


def attn_bwd(q, k, v, o, d_o, lse, delta, dq, dk, dv, causal=False, dq_scale=1.0, dq_colsum=None, dv_colsum=None, q_log2=False,
             fused=None):
    """Backward of attn_fwd.  All [B,L,H,64] bf16 views; lse [B,H,Lq] fp32; delta = workspace [2,B,H,Lq] fp32.
    fused: hand the library the workspace of its one-kernel form (used for dense problems that fill the chip; None = the
    DICOW_ATTN_BWD_FUSED default; "force" = for every dense problem, whatever its size -- tests)."""
    assert delta.numel() >= 2 * lse.numel(), "attn_bwd: delta workspace must hold 2*B*H*Lq floats"
    a = L.AttnBwdArgs()
    a.q, a.k, a.v, a.o, a.d_o = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), d_o.data_ptr()
    a.lse, a.delta = lse.data_ptr(), delta.data_ptr()
    a.dq, a.dk, a.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    a.q_bs, a.q_rs = _bs_rs(q, "q")
    a.k_bs, a.k_rs = _bs_rs(k, "k")
    a.v_bs, a.v_rs = _bs_rs(v, "v")
    a.o_bs, a.o_rs = _bs_rs(o, "o")
    a.do_bs, a.do_rs = _bs_rs(d_o, "d_o")
    a.dq_bs, a.dq_rs = _bs_rs(dq, "dq")
    a.dk_bs, a.dk_rs = _bs_rs(dk, "dk")
    a.dv_bs, a.dv_rs = _bs_rs(dv, "dv")
    a.B, a.Lq, a.H = q.shape[0], q.shape[1], q.shape[2]
    a.Lk, a.causal, a.dq_scale, a.q_log2 = k.shape[1], int(causal), dq_scale, int(q_log2)
    if dq_colsum is not None or dv_colsum is not None:       # fused q / v bias gradients ([H*64] fp32, accumulated)
        ws = workspace(L.lib().dicow_attn_bwd_colsum_ws_bytes(a.B, a.H, a.Lq, a.Lk), q.device)
        a.dq_colsum, a.dv_colsum, a.cs_ws, a.cs_ws_bytes = _p(dq_colsum), _p(dv_colsum), ws.data_ptr(), ws.numel()
    want = ATTN_BWD_FUSED if fused is None else fused
    if want and not causal and a.B * a.H <= 1000 and (want == "force" or (a.B * a.H * ((a.Lk + 127) // 128) >= 1024 and a.Lq >= 256)):
        fws = attn_bwd_fused_ws(a.B, a.H, a.Lq, a.Lk, q.device)
        a.fused_ws, a.fused_ws_bytes, a.fused_mode = fws.data_ptr(), fws.numel(), int(want == "force")
    L.call_struct("dicow_attn_bwd", a)


# ------------------------------------------------------------------------------------------------ loss / embedding / misc
def ce_args(logits, ld, rows, V, labels, upp_labels, soft, ts, lse, row_loss, choice, loss_sum, count, d_logits=None):
    _req(labels, torch.int64, "ce.labels")
    if upp_labels is not None:
        _req(upp_labels, torch.int64, "ce.upp_labels")
    a = L.CeArgs()
    a.logits, a.ld, a.rows, a.V = logits.data_ptr(), ld, rows, V
    a.labels, a.upp_labels, a.soft = labels.data_ptr(), _p(upp_labels), int(soft)
    if ts is not None:
        a.ts_index, a.ts_ids, a.ts_w, a.n_ts = ts["index"].data_ptr(), ts["ids"].data_ptr(), ts["w"].data_ptr(), ts["ids"].numel()
    a.lse, a.row_loss, a.choice = lse.data_ptr(), row_loss.data_ptr(), choice.data_ptr()
    a.loss_sum, a.count, a.d_logits = loss_sum.data_ptr(), count.data_ptr(), _p(d_logits)
    return a


def ce_loss_fwd(a):
    L.call_struct("dicow_ce_loss_fwd", a)


def ce_loss_bwd(a, grad_scale):
    L.check(L.lib().dicow_ce_loss_bwd(C.byref(a), grad_scale.data_ptr(), L.stream()), "dicow_ce_loss_bwd")


def embed_fwd(ids, tok, pos, out):
    B, Lq = ids.shape
    L.call("dicow_embed_fwd", ids.data_ptr(), tok.data_ptr(), pos.data_ptr(), out.data_ptr(), B, Lq, tok.shape[1], L.stream())


def embed_bwd(ids, g, d_tok, d_pos, D):
    B, Lq = ids.shape
    L.call("dicow_embed_bwd", ids.data_ptr(), g.data_ptr(), _p(d_tok), _p(d_pos), B, Lq, D, L.stream())


def gelu_bwd_bf16(g, pre, out):
    L.call("dicow_gelu_bwd_bf16", g.data_ptr(), pre.data_ptr(), out.data_ptr(), g.numel(), L.stream())


def conv2_col2im_gelu_bwd(dA2, pre1, d_pre1, B, T2, Cc):
    L.call("dicow_conv2_col2im_gelu_bwd", dA2.data_ptr(), _p(pre1), d_pre1.data_ptr(), B, T2, Cc, L.stream())


def ctc_args(logits, ld, B, Tn, Cc, labels, lse, alpha, beta, nll, tlen, loss_sum):
    _req(labels, torch.int64, "ctc.labels")
    a = L.CtcArgs()
    a.logits, a.ld, a.B, a.Tn, a.C = logits.data_ptr(), ld, B, Tn, Cc
    a.labels, a.Lc, a.blank = labels.data_ptr(), labels.shape[1], Cc - 1
    a.lse, a.alpha, a.beta, a.Smax = lse.data_ptr(), alpha.data_ptr(), beta.data_ptr(), 2 * labels.shape[1] + 1
    a.nll, a.tlen, a.loss_sum = nll.data_ptr(), tlen.data_ptr(), loss_sum.data_ptr()
    return a


def ctc_loss_fwd(a):
    L.call_struct("dicow_ctc_loss_fwd", a)


def ctc_loss_bwd(a, grad_scale):
    L.check(L.lib().dicow_ctc_loss_bwd(C.byref(a), grad_scale.data_ptr(), L.stream()), "dicow_ctc_loss_bwd")


def fabric_emulate(buf, gbps, workgroups=16, passes=2):
    """The GPU-side load of an 8-rank all-reduce of `buf` at `gbps` GB/s, on one GPU (dicow_fabric_emulate): in place, values unchanged."""
    L.call("dicow_fabric_emulate", buf.data_ptr(), buf.numel() * buf.element_size(), float(gbps), int(workgroups), int(passes), L.stream())


def sumsq(x, out):
    L.call("dicow_sumsq_f32", x.data_ptr(), x.numel(), out.data_ptr(), L.stream())


def adamw(p, g, m, v, lr, beta1, beta2, eps, wd, step, gnorm_sq=None, max_norm=0.0):
    L.call("dicow_adamw_f32", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2, eps, wd,
           step, _p(gnorm_sq), max_norm, L.stream())


def adamw_dev(p, g, m, v, hyper, beta1, beta2, eps, wd, gnorm_sq=None, max_norm=0.0):
    """adamw with {lr, 1 - beta1^t, 1 - beta2^t} read from the 3-float device tensor `hyper` (hipGraph replay)."""
    L.call("dicow_adamw_f32_dev", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), hyper.data_ptr(), beta1, beta2,
           eps, wd, _p(gnorm_sq), max_norm, L.stream())
