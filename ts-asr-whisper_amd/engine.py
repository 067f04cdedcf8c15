"""Forward / backward orchestration of the DiCoW encoder and Whisper decoder over the HIP C ABI.

No torch compute kernels on the hot path: torch is used for device memory (``torch.empty``), streams, and a
handful of tiny integer / scalar ops.  Every arithmetic step is a ``libdicow_hip.so`` entry point (``ops.py``).

Precision policy = the reference's ``bf16: true`` AMP recipe (configs/base.yaml:49): bf16 GEMM /
attention operands and outputs, fp32 accumulation, fp32 LayerNorm / FDDT / residual stream / loss, fp32 master
parameters and gradients.

Reference structure followed (paths under /root/reference): encoder.py:140-246 (stem, initial FDDT, positions,
per-layer FDDT + optional speaker-communication block + WhisperEncoderLayer, final LayerNorm),
layers.py:145-193 (SCB), HF modeling_whisper.py WhisperEncoderLayer / WhisperDecoderLayer / WhisperAttention,
modeling_dicow.py:248-338 (shift, tied LM head, losses).
"""
import contextlib
import os
from types import SimpleNamespace as NS

import torch

from . import _lib as L
from . import ops

BF16, F32 = torch.bfloat16, torch.float32
CLS = ("silence_linear", "target_linear", "non_target_linear", "overlap_linear")     # stno channel order S,T,N,O


def _e(shape, dtype, dev):
    return torch.empty(shape, dtype=dtype, device=dev)


def _ceil(a, b):
    return (a + b - 1) // b * b


# Weight gradients beside the gradient chain (round 6).  The pooled weight-gradient launch of a layer (ops.TnGroup.run) is off the critical
# path: nothing in the backward pass reads a dW.  With WGRAD_SIDE_STREAM it is enqueued on a second HIP stream behind an event that says
# "this layer's operands are written", and the next layer's dgrad chain starts at once: the HBM-bound row kernels, the attention kernel's
# half-empty last round and the GEMM tails of the chain run beside a matrix-bound kernel instead of after it.  Same kernels, same operands,
# one writer per dW: bit-identical to the one-stream order.  The side stream is a PRIORITY stream (a queue of its own: a normal-priority pool
# stream may share the compute stream's hardware queue, profiles/r06_streams.txt); the backward pass joins it before it returns.
WGRAD_SIDE_STREAM = os.environ.get("DICOW_WGRAD_STREAM", "0") == "1"
WGRAD_STREAM_PRIORITY = int(os.environ.get("DICOW_WGRAD_STREAM_PRIORITY", "-1"))
_WGRAD_STREAMS = {}


def wgrad_stream(dev):
    cur = torch.cuda.current_stream(dev)
    key = (dev.index, cur.cuda_stream)
    st = _WGRAD_STREAMS.get(key)
    if st is None:
        st = _WGRAD_STREAMS[key] = torch.cuda.Stream(device=dev, priority=WGRAD_STREAM_PRIORITY)
    return st


# The encoder FORWARD of a large even batch as two half batches on two HIP streams, into the halves of the SAME full-batch activation
# buffers (rows are batch-major: a half batch is a contiguous row range of every buffer).  Every forward kernel is row-parallel, so
# the results are bit-identical to the one-stream forward; the HBM-bound row kernels, the attention tails and the partial last rounds
# of one half fill under the other half's matrix kernels (tools/dual_stream_probe.py: -3.4 %, profiles/r06_streams.txt item 3).  The
# BACKWARD then runs as ONE full batch on one stream, exactly as before -- no gradient is summed in two pieces, nothing to order
# (what trainer.SplitSync has to do for the whole-step split).  Half 0 stays on the caller's stream, half 1 runs on a high-priority
# stream of its own hardware queue (a normal-priority pool stream may share the caller's queue: profiles/r06_streams.txt item 1).
SPLIT_FWD = os.environ.get("DICOW_SPLIT_FWD", "1") != "0"
SPLIT_FWD_MIN_ROWS = int(os.environ.get("DICOW_SPLIT_FWD_MIN_ROWS", "16000"))      # whisper-base B = 8 (12000 rows, one hipGraph) stays on one stream
SPLIT_DEC = os.environ.get("DICOW_SPLIT_DEC", "1") != "0"                     # the FROZEN decoder's layers too (forward and backward: DecoderEngine)
SPLIT_BWD = os.environ.get("DICOW_SPLIT_BWD", "1") != "0"                     # the encoder BACKWARD's row-parallel chain as two halves too (EncoderEngine._backward_split)
SPLIT_BWD_WGRAD = os.environ.get("DICOW_SPLIT_BWD_WGRAD", "alt")               # which stream launches a layer's pooled weight gradients: "alt" = odd layers on the side stream (the streams carry equal work: 124.5-124.6 ms per step) | "main" (124.7-124.9; one stream 126.3) | "third" = every pooled launch on a third queue, settled one layer later still (measured +1.5 ms: 129.5 against 127.9-128.1, profiles/r06_split_bwd.txt)
SPLIT_BWD_MEM_FRACTION = float(os.environ.get("DICOW_SPLIT_BWD_MEM_FRACTION", "0.6"))   # the split backward only if its buffers fit into this share of the free memory
SPLIT_IN_CAPTURE = os.environ.get("DICOW_SPLIT_IN_CAPTURE", "0") == "1"      # experiment: fork the two half-batch streams inside a hipGraph capture too (parallel branches of the graph)
SPLIT_FWD_PARTS = int(os.environ.get("DICOW_SPLIT_FWD_PARTS", "2"))          # (4 measured against 2: profiles/r06_split_fwd.txt)
_FWD_STREAMS = {}


def _split_allowed():
    return SPLIT_IN_CAPTURE or not torch.cuda.is_current_stream_capturing()


def fwd_side_stream(dev, k=0):
    cur = torch.cuda.current_stream(dev)
    key = (dev.index, cur.cuda_stream, k)
    st = _FWD_STREAMS.get(key)
    if st is None:
        st = _FWD_STREAMS[key] = torch.cuda.Stream(device=dev, priority=-1)
    return st


def _fwd_parts(B, T, dev):
    """[(stream, row slice, batch slice)] of the forward's parts: part 0 on the caller's stream, the others on side streams (forked here)."""
    n = SPLIT_FWD_PARTS if (SPLIT_FWD_PARTS > 2 and B % SPLIT_FWD_PARTS == 0) else 2
    main_st, Bp = torch.cuda.current_stream(dev), B // n
    parts = []
    for k in range(n):
        st = main_st if k == 0 else fwd_side_stream(dev, k - 1)
        if k:
            st.wait_stream(main_st)
        parts.append((st, slice(k * Bp * T, (k + 1) * Bp * T), slice(k * Bp, (k + 1) * Bp)))
    return main_st, parts, Bp


class GradSink:
    """fp32 gradient buffers for a set of parameters.  Parameters that carry a persistent ``_direct_grad`` view (set
    by trainer.FlatStore) are accumulated into directly; the others get ONE zeroed flat buffer whose views are
    returned to autograd.  ``get(p)`` returns None for parameters that do not require grad."""

    def __init__(self, params, dev, extra_rows=None):
        self.index, self.direct = {}, {}
        off = 0
        extra_rows = extra_rows or {}
        for p in params:
            if p is None or not p.requires_grad or id(p) in self.index or id(p) in self.direct:
                continue
            if getattr(p, "_direct_grad", None) is not None:
                self.direct[id(p)] = p._direct_grad
                continue
            n = p.numel() + extra_rows.get(id(p), 0)
            self.index[id(p)] = (off, p.shape, p.numel())
            off += _ceil(n, 64)
        self.flat = torch.zeros(max(off, 1), dtype=F32, device=dev)

    def get(self, p):
        if p is None:
            return None
        if id(p) in self.direct:
            return self.direct[id(p)]
        if id(p) not in self.index:
            return None
        off, shape, n = self.index[id(p)]
        return self.flat[off:off + n].view(shape)

    def result(self, p):
        """What autograd should receive for p: None when the gradient was accumulated in place."""
        return None if (p is None or id(p) in self.direct) else self.get(p)

    def is_direct(self, p):
        return p is not None and id(p) in self.direct

    def raw(self, p, n):
        if id(p) in self.direct:
            d = self.direct[id(p)]
            assert d.numel() == n, "GradSink.raw: a flat-store gradient has no padding rows"
            return d.view(-1)
        off, _, _ = self.index[id(p)]
        return self.flat[off:off + n]


# ------------------------------------------------------------------------------------------------ weight preparation
class LinW:
    """bf16 compute copies of one Linear: w [N,K] (forward), wt [K,N] (dgrad), fp32 bias."""
    __slots__ = ("w", "wt", "b", "N", "K")


_BIAS_COPIES = None           # inside EncoderEngine.prepare(): (dst slice, src) pairs of fused bias vectors, flushed as ONE multi-tensor copy


def prep_linear(weights, biases, dev, need_t=True, n_pad=None, scale_info=None, bias_buf=None):
    """Fuse several Linear weights [N_i, K] (same K) into one bf16 [sum N_i, K] (+ transposed copy).
    bias_buf: a persistent fp32 [sum N_i] buffer for the fused bias (parts without a bias stay zero in it): the parts are copied in
    by the caller's pooled copy (_BIAS_COPIES) instead of a zeros + cat pair of launches per layer and step."""
    K = weights[0].shape[1]
    Ns = [w.shape[0] for w in weights]
    N = sum(Ns)
    Np = N if n_pad is None else n_pad
    lw = LinW()
    lw.N, lw.K = Np, K
    lw.w = torch.zeros(Np, K, dtype=BF16, device=dev) if Np != N else _e((Np, K), BF16, dev)
    lw.wt = (torch.zeros(K, Np, dtype=BF16, device=dev) if Np != N else _e((K, Np), BF16, dev)) if need_t else None
    off = 0
    for w, n in zip(weights, Ns):
        ops.cast_transpose_bf16(w.detach(), out=lw.w[off:off + n], out_t=None if lw.wt is None else lw.wt[:, off:off + n],
                                ld=K, ld_t=Np)
        off += n
    if biases is None or all(b is None for b in biases):
        lw.b = None
    else:
        if bias_buf is not None and _BIAS_COPIES is not None and len(Ns) > 1:
            off = 0
            for b, n in zip(biases, Ns):
                if b is not None:
                    _BIAS_COPIES.append((bias_buf[off:off + n], b.detach()))
                off += n
            lw.b = bias_buf
        else:
            parts = [(b.detach() if b is not None else torch.zeros(n, dtype=F32, device=dev)) for b, n in zip(biases, Ns)]
            lw.b = parts[0] if len(parts) == 1 else torch.cat(parts)
    return lw


def prep_attention(att, dev, fuse_qkv=True, bias_buf=None):
    a = NS()
    if fuse_qkv:
        a.qkv = prep_linear([att.q_proj.weight, att.k_proj.weight, att.v_proj.weight],
                            [att.q_proj.bias, None, att.v_proj.bias], dev, bias_buf=bias_buf)
    else:
        a.q = prep_linear([att.q_proj.weight], [att.q_proj.bias], dev)
        a.kv = prep_linear([att.k_proj.weight, att.v_proj.weight], [None, att.v_proj.bias], dev)
    a.o = prep_linear([att.out_proj.weight], [att.out_proj.bias], dev)
    return a


def fddt_ptrs(fddt, cfg):
    """(mode, w[4], b[4]) raw parameter tensors of an FDDT module in S,T,N,O order; disabled classes -> None."""
    if fddt is None:
        return ops.MODE_NONE, (None,) * 4, (None,) * 4
    w, b = [], []
    for c in CLS:
        m = getattr(fddt, c, None)
        if fddt.bias_only:
            w.append(None)
            b.append(None if m is None else m)
        else:
            w.append(None if m is None else m.weight)
            b.append(None if m is None else m.bias)
    return (ops.MODE_BIAS if fddt.bias_only else ops.MODE_DIAG), tuple(w), tuple(b)


# full (D x D) FDDT (FDDT.py:13-16, 53-62): one GEMM [rows, D] x [D, 4D] + masked combine; shared by the standalone FDDT
# module and the fused encoder path
def fddt_is_full(fddt):
    return fddt is not None and not fddt.is_diagonal


def prep_full_fddt(fddt, dev):
    D = fddt.d_model
    w = NS(use=0)
    w.Wc = torch.zeros(4 * D, D, dtype=BF16, device=dev)
    w.Wct = torch.zeros(D, 4 * D, dtype=BF16, device=dev)
    w.bias = torch.zeros(4 * D, dtype=F32, device=dev)
    for c, name in enumerate(CLS):
        m = getattr(fddt, name, None)
        if m is not None:
            w.use |= 1 << c
            ops.cast_transpose_bf16(m.weight.detach(), out=w.Wc[c * D:(c + 1) * D], out_t=w.Wct[:, c * D:(c + 1) * D], ld=D, ld_t=4 * D)
            w.bias[c * D:(c + 1) * D] = m.bias.detach()
    return w


def full_fddt_fwd(w, h, stno, bstride, rows, T, D):
    """h fp32 or bf16 [rows, D] -> (h' fp32 [rows, D], bf16 copy of h kept for the weight gradients)."""
    hb = h if h.dtype == BF16 else ops.cast_bf16(h)
    y4 = _e((rows, 4 * D), BF16, h.device)
    ops.gemm_nt(hb, w.Wc, y4, rows, 4 * D, D, bias=w.bias)
    out = _e((rows, D), F32, h.device)
    L.call("dicow_fddt_full_combine_fwd", y4.data_ptr(), h.data_ptr(), int(h.dtype == BF16), stno.data_ptr(), bstride, w.use,
           out.data_ptr(), rows, T, D, L.stream())
    return out, hb


def full_fddt_bwd(fddt, w, hb, g, stno, bstride, G, rows, T, D):
    """g fp32 [rows, D] = dL/dh' -> dL/dh fp32 [rows, D]; weight / bias gradients accumulate into G."""
    dev = g.device
    d_y4 = torch.zeros(rows, 4 * D, dtype=BF16, device=dev)
    gh = _e((rows, D), F32, dev)
    L.call("dicow_fddt_full_combine_bwd", g.data_ptr(), stno.data_ptr(), bstride, w.use, d_y4.data_ptr(), gh.data_ptr(), rows, T, D,
           L.stream())
    ops.gemm_nt(d_y4, w.Wct, gh, rows, D, 4 * D, flags=L.EPI_ACCUM)
    for c, name in enumerate(CLS):
        m = getattr(fddt, name, None)
        if m is None:
            continue
        sl = d_y4[:, c * D:(c + 1) * D]
        if G.get(m.bias) is not None:
            ops.colsum_bf16(sl, G.get(m.bias))
        if G.get(m.weight) is not None:
            ops.gemm_tn(sl, hb, G.get(m.weight), rows, D, D, lda=4 * D, ldb=D, ldc=D)
    return gh


# Attention scores in base-2 units (round 3): log2(e) is folded into the q-scale of the projection epilogue (q = bf16((Wx + b) *
# head_dim^-0.5 * log2 e) -- ONE rounding, like the reference's bf16(Wx + b) * head_dim^-0.5, but not the same one), so the forward
# kernel's probabilities are p = 2^s with no arithmetic in front of the exponential and the backward kernels lose their per-score
# multiply.  Same softmax, same lse (natural log), same gradients; dq_scale keeps its meaning (include/dicow_hip.h).
# (DICOW_QK_LOG2=0 in the environment selects the plain form for A/B runs -- tools used to edit this file in place.)
QK_LOG2 = os.environ.get("DICOW_QK_LOG2", "1") != "0"
FUSE_NEXT_FDDT = True         # inference forward: the next layer's diagonal FDDT in the fc2 epilogue (see EncoderEngine.forward)
# LayerNorm folded into the GEMMs on either side of it (round 4; include/dicow_hip.h, DICOW_EPI_LNSTAT / LNFOLD): out-proj / fc2
# also store bf16(h) and per-row partial statistics, qkv / fc1 read bf16(h) against gamma-scaled weights and normalise in their
# epilogues -- no LayerNorm launch between them.  BUILT, PARITY-TESTED (tests/test_gpu_lnfold.py) AND MEASURED SLOWER: the two
# LayerNorm launches it deletes are worth 63 us per layer, the epilogue work it adds 114 us (profiles/r04_lnfold.txt) -- OFF by
# default, DICOW_LN_FOLD=1 switches the inference forward to it (A/B runs); shapes that do not reach the persistent 192 x 320
# kernel keep the LayerNorm kernels either way.
LN_FOLD = os.environ.get("DICOW_LN_FOLD", "0") == "1"
Q_SCALE = 0.125 * (ops.LOG2E if QK_LOG2 else 1.0)


# ------------------------------------------------------------------------------------------------ building blocks
def linear_fwd(x, lw, M, out_dtype=BF16, residual=None, gelu_aux=None, flags=0, scale=1.0, scale_ncols=0, out=None, fddt=None):
    dev = x.device
    if out is None:
        out = _e((M, lw.N), out_dtype, dev)
    ops.gemm_nt(x, lw.w, out, M, lw.N, lw.K, bias=lw.b, residual=residual, aux=gelu_aux,
                flags=flags | ((L.EPI_GELU | L.EPI_GELU_DAUX) if gelu_aux is not None else 0), scale=scale, scale_ncols=scale_ncols, fddt=fddt)
    return out


def linear_dgrad(dy, lw, M, aux=None, out=None, out_dtype=BF16, accumulate=False, dy_cols=None, colsum_out=None):
    """dx[M,K] = dy[M,N] @ W[N,K]  (NT GEMM against the transposed copy); `aux` = gelu' saved by linear_fwd(gelu_aux=...)."""
    N = lw.N if dy_cols is None else dy_cols
    if out is None:
        out = _e((M, lw.K), out_dtype, dy.device)
    flags = (L.EPI_MUL_AUX if aux is not None else 0) | (L.EPI_ACCUM if accumulate else 0)
    ops.gemm_nt(dy, lw.wt, out, M, lw.K, N, lda=dy.stride(0), ldb=lw.wt.stride(0), aux=aux, flags=flags, colsum_out=colsum_out)
    return out


def _first_writer(*params):
    """True when every listed parameter's gradient is flagged "write, do not accumulate" for this micro-batch
    (trainer.FlatStore.zero_grad(first_writer=True)); the flags are consumed: the next micro-batch accumulates."""
    if not params or not all(getattr(p, "_grad_overwrite", False) for p in params):
        return False
    for p in params:
        p._grad_overwrite = False
    return True


def linear_wgrad(dy, x, gw, M, n_rows=None, group=None, param=None):
    """dW[N,K] += dy[M,N]^T @ x[M,K]; dy/x may be column slices (strided).  group: an ops.TnGroup that collects the layer's
    weight gradients for one pooled launch (the operands must stay untouched until group.run()).  param: the parameter gw
    belongs to -- when its gradient is flagged first-writer the GEMM overwrites (dW = ...) instead of accumulating."""
    if gw is None:
        return
    N = dy.shape[1] if n_rows is None else n_rows
    acc = not (param is not None and _first_writer(param))
    if group is not None:
        group.add(dy, x, gw, M, N, x.shape[1], lda=dy.stride(0), ldb=x.stride(0), ldc=gw.stride(0), accumulate=acc)
        return
    ops.gemm_tn(dy, x, gw, M, N, x.shape[1], lda=dy.stride(0), ldb=x.stride(0), ldc=gw.stride(0), accumulate=acc)


def qkv_wgrad(d_qkv, x, gq, gk, gv, M, D, group=None, params=None):
    """Weight gradients of the fused q/k/v projection from ONE TN GEMM (C rows segmented over the three tensors)."""
    if gq is not None and gk is not None and gv is not None and D % 128 == 0:
        acc = not (params is not None and _first_writer(*params))
        kw = dict(lda=d_qkv.stride(0), ldb=x.stride(0), ldc=D, C_seg=(gk, gv), seg_rows=D, accumulate=acc)
        if group is not None:
            group.add(d_qkv, x, gq, M, 3 * D, D, **kw)
        else:
            ops.gemm_tn(d_qkv, x, gq, M, 3 * D, D, **kw)
        return
    pq, pk, pv = params if params is not None else (None, None, None)
    linear_wgrad(d_qkv[:, :D], x, gq, M, group=group, param=pq)
    linear_wgrad(d_qkv[:, D:2 * D], x, gk, M, group=group, param=pk)
    linear_wgrad(d_qkv[:, 2 * D:], x, gv, M, group=group, param=pv)


def bias_grad(dy, gb):
    if gb is not None:
        ops.colsum_bf16(dy, gb)


def heads(t, B, Lx, H):
    """[B*L, >=H*64] column slice -> [B, L, H, 64] strided view."""
    return t.as_strided((B, Lx, H, 64), (Lx * t.stride(0), t.stride(0), 64, 1), t.storage_offset())


# ------------------------------------------------------------------------------------------------ encoder
def plain_layer_fwd(lyr, w, h, B, T, H, F_):
    """One pre-LN Whisper encoder layer without FDDT / SCB on fp32 rows ``h`` [B*T, D] (HF WhisperEncoderLayer; the CTC
    branch's ``additional_layer``, reference encoder.py:16-17,88-94).  Returns (fp32 rows, saved activations)."""
    rows, D = h.shape
    dev = h.device
    s = NS(h_in=h)
    s.xln, s.mean, s.rstd = _e((rows, D), BF16, dev), _e((rows,), F32, dev), _e((rows,), F32, dev)
    ln = lyr.self_attn_layer_norm
    ops.fddt_ln_fwd(h, rows, D, mode=ops.MODE_NONE, ln_w=ln.weight.detach(), ln_b=ln.bias.detach(), y_bf16=s.xln, mean=s.mean,
                    rstd=s.rstd)
    s.qkv = linear_fwd(s.xln, w.att.qkv, rows, flags=L.EPI_SCALE_N, scale=Q_SCALE, scale_ncols=D)
    s.o, s.lse = _e((rows, D), BF16, dev), _e((B, H, T), F32, dev)
    ops.attn_fwd(heads(s.qkv[:, :D], B, T, H), heads(s.qkv[:, D:2 * D], B, T, H), heads(s.qkv[:, 2 * D:], B, T, H),
                 heads(s.o, B, T, H), s.lse, q_log2=QK_LOG2)
    s.h2 = linear_fwd(s.o, w.att.o, rows, out_dtype=F32, residual=h)
    s.xln2, s.mean2, s.rstd2 = _e((rows, D), BF16, dev), _e((rows,), F32, dev), _e((rows,), F32, dev)
    ln2 = lyr.final_layer_norm
    ops.fddt_ln_fwd(s.h2, rows, D, mode=ops.MODE_NONE, ln_w=ln2.weight.detach(), ln_b=ln2.bias.detach(), y_bf16=s.xln2,
                    mean=s.mean2, rstd=s.rstd2)
    s.u = _e((rows, F_), BF16, dev)
    s.a = linear_fwd(s.xln2, w.fc1, rows, gelu_aux=s.u)
    return linear_fwd(s.a, w.fc2, rows, out_dtype=F32, residual=s.h2), s


def plain_layer_bwd(lyr, w, s, gb, G, B, T, H):
    """Backward of plain_layer_fwd.  gb: bf16 rows, gradient wrt the layer output.  Returns the fp32 gradient wrt its input."""
    rows, D = gb.shape
    dev = gb.device
    g = gb.float()
    bias_grad(gb, G.get(lyr.fc2.bias))
    linear_wgrad(gb, s.a, G.get(lyr.fc2.weight), rows)
    d_u = linear_dgrad(gb, w.fc2, rows, aux=s.u, colsum_out=G.get(lyr.fc1.bias))
    linear_wgrad(d_u, s.xln2, G.get(lyr.fc1.weight), rows)
    d_xln2 = linear_dgrad(d_u, w.fc1, rows)
    g2, g2b = _e((rows, D), F32, dev), _e((rows, D), BF16, dev)
    ln2, att = lyr.final_layer_norm, lyr.self_attn
    ops.fddt_ln_bwd(s.h2, rows, D, mode=ops.MODE_NONE, ln_w=ln2.weight.detach(), mean=s.mean2, rstd=s.rstd2, d_y=d_xln2, g_res=g,
                    g_out=g2, g_out_bf16=g2b, dln_w=G.get(ln2.weight), dln_b=G.get(ln2.bias), colsum_out=G.get(att.out_proj.bias))
    linear_wgrad(g2b, s.o, G.get(att.out_proj.weight), rows)
    d_o = linear_dgrad(g2b, w.att.o, rows)
    d_qkv, delta = _e((rows, 3 * D), BF16, dev), _e((2, B, H, T), F32, dev)
    qkv = s.qkv
    ops.attn_bwd(heads(qkv[:, :D], B, T, H), heads(qkv[:, D:2 * D], B, T, H), heads(qkv[:, 2 * D:], B, T, H),
                 heads(s.o, B, T, H), heads(d_o, B, T, H), s.lse, delta, heads(d_qkv[:, :D], B, T, H),
                 heads(d_qkv[:, D:2 * D], B, T, H), heads(d_qkv[:, 2 * D:], B, T, H), dq_scale=0.125, q_log2=QK_LOG2,
                 dq_colsum=G.get(att.q_proj.bias), dv_colsum=G.get(att.v_proj.bias))
    qkv_wgrad(d_qkv, s.xln, G.get(att.q_proj.weight), G.get(att.k_proj.weight), G.get(att.v_proj.weight), rows, D)
    d_xln = linear_dgrad(d_qkv, w.att.qkv, rows)
    g0 = _e((rows, D), F32, dev)
    ln = lyr.self_attn_layer_norm
    ops.fddt_ln_bwd(s.h_in, rows, D, mode=ops.MODE_NONE, ln_w=ln.weight.detach(), mean=s.mean, rstd=s.rstd, d_y=d_xln, g_res=g2,
                    g_out=g0, dln_w=G.get(ln.weight), dln_b=G.get(ln.bias))
    return g0


def conv_stem_fwd(enc, W, input_features, need_grad=True):
    """gelu(conv1(x)) -> gelu(conv2(.)) of HF WhisperEncoder (reference encoder.py:167-170: Conv1d k=3 p=1, Conv1d k=3 s=2 p=1)
    as two NT GEMMs over time-major views: a k=3 convolution is a GEMM whose A rows are three consecutive (zero-padded) frames.
    input_features [B, M, 2T] -> (x2 bf16 [B*T, D], saved activations for conv_stem_bwd).  W: EncoderEngine.prepare()."""
    dev = input_features.device
    B, M, Tin = input_features.shape
    T, D = Tin // 2, enc.conv1.weight.shape[0]
    xt = torch.zeros(B * (Tin + 2) * M + W.k1, dtype=BF16, device=dev)        # + slack for the K padding reads
    ops.mel_to_timemajor(input_features.to(F32), out=xt)
    g1p = _e((B, Tin + 2, D), BF16, dev)
    g1p[:, 0].zero_()
    g1p[:, Tin + 1].zero_()
    # (the pre-activations are only kept for the backward pass: an inference forward does not write them -- 184 MB at B = 16)
    pre1 = _e((B, Tin, D), BF16, dev) if need_grad else None
    ops.gemm_nt(xt, W.conv1, g1p[:, 1:], Tin, D, W.k1, lda=M, bias=enc.conv1.bias.detach(), aux=pre1, flags=L.EPI_GELU,
                batch=B, strideA=(Tin + 2) * M, strideC=(Tin + 2) * D, strideAux=Tin * D)
    x2 = _e((B * T, D), BF16, dev)
    pre2 = _e((B * T, D), BF16, dev) if need_grad else None
    ops.gemm_nt(g1p, W.conv2, x2, T, D, 3 * D, lda=2 * D, bias=enc.conv2.bias.detach(), aux=pre2, flags=L.EPI_GELU,
                batch=B, strideA=(Tin + 2) * D, strideC=T * D, strideAux=T * D)
    return x2, NS(xt=xt, g1p=g1p, pre1=pre1, pre2=pre2, B=B, T=T, M=M)


def conv_stem_bwd(enc, W, st, d_x2, G):
    """Backward of conv_stem_fwd.  d_x2: bf16 [B*T, D] gradient wrt its output; weight / bias gradients are accumulated into
    G.get(param) (skipped where that is None).  The mel input receives no gradient (it is data)."""
    dev = d_x2.device
    B, T, M = st.B, st.T, st.M
    Tin, rows, D = 2 * T, B * T, d_x2.shape[1]
    d_pre2 = _e((rows, D), BF16, dev)
    ops.gelu_bwd_bf16(d_x2, st.pre2, d_pre2)
    bias_grad(d_pre2, G.get(enc.conv2.bias))
    gw2 = G.get(enc.conv2.weight)
    if gw2 is not None:
        tmp = torch.zeros(D, 3 * D, dtype=F32, device=dev)
        ops.gemm_tn(d_pre2, st.g1p, tmp, T, D, 3 * D, lda=D, ldb=2 * D, batch=B, strideA=T * D, strideB=(Tin + 2) * D)
        ops.conv_weight_unpack_grad(tmp, gw2)
    dA2 = _e((rows, 3 * D), BF16, dev)
    ops.gemm_nt(d_pre2, W.conv2_t, dA2, rows, 3 * D, D)
    d_pre1 = _e((B, Tin, D), BF16, dev)
    ops.conv2_col2im_gelu_bwd(dA2, st.pre1, d_pre1, B, T, D)
    bias_grad(d_pre1.view(B * Tin, D), G.get(enc.conv1.bias))
    gw1 = G.get(enc.conv1.weight)
    if gw1 is not None:
        tmp = torch.zeros(D, W.k1, dtype=F32, device=dev)
        ops.gemm_tn(d_pre1, st.xt, tmp, Tin, D, W.k1, lda=D, ldb=M, batch=B, strideA=Tin * D, strideB=(Tin + 2) * M)
        ops.conv_weight_unpack_grad(tmp, gw1)


class EncoderEngine:
    def __init__(self, enc):
        self.enc = enc
        self.cfg = enc.config

    # -- bf16 weight copies (once per optimizer step)
    def prepare(self):
        enc, cfg = self.enc, self.cfg
        dev = enc.conv1.weight.device
        D, M = cfg.d_model, cfg.num_mel_bins
        W = NS()
        W.k1 = _ceil(3 * M, 64)
        W.conv1, W.conv1_t = ops.conv_weight_pack(enc.conv1.weight.detach(), W.k1, want_t=True)
        W.conv2, W.conv2_t = ops.conv_weight_pack(enc.conv2.weight.detach(), 3 * D, want_t=True)
        W.layers = []
        # fused q | k | v bias vectors: persistent buffers (k has no bias: its third stays zero), refreshed by ONE multi-tensor copy
        global _BIAS_COPIES
        if getattr(self, "_qkv_bias", None) is None or self._qkv_bias[0].device != dev:
            self._qkv_bias = [torch.zeros(3 * D, dtype=F32, device=dev) for _ in enc.layers]
        _BIAS_COPIES = []
        for li, lyr in enumerate(enc.layers):
            w = NS()
            with ops.cast_group():               # the layer's six matrices: one pooled cast + transpose launch
                w.att = prep_attention(lyr.self_attn, dev, bias_buf=self._qkv_bias[li])
                w.fc1 = prep_linear([lyr.fc1.weight], [lyr.fc1.bias], dev)
                w.fc2 = prep_linear([lyr.fc2.weight], [lyr.fc2.bias], dev)
            W.layers.append(w)
        if _BIAS_COPIES:
            torch._foreach_copy_([d for d, _ in _BIAS_COPIES], [s_ for _, s_ in _BIAS_COPIES])
        _BIAS_COPIES = None
        W.scb = []
        if cfg.use_enrollments and cfg.scb_layers:
            for blk in enc.ca_enrolls:
                s = NS()
                with ops.cast_group():
                    s.att = prep_attention(blk.cae.cross_attn, dev, fuse_qkv=False)
                    s.f0 = prep_linear([blk.cae.ffn[0].weight], [blk.cae.ffn[0].bias], dev)
                    s.f3 = prep_linear([blk.cae.ffn[3].weight], [blk.cae.ffn[3].bias], dev)
                W.scb.append(s)
        W.full_init = prep_full_fddt(enc.initial_fddt, dev) if (cfg.use_fddt and cfg.use_pre_pos_fddt and fddt_is_full(enc.initial_fddt)) else None
        W.full = [prep_full_fddt(f, dev) if fddt_is_full(f) else None for f in (enc.fddts if cfg.use_fddt else [])]
        W.fold = None                                     # LayerNorm-folded copies: built on first use (prepare_fold)
        self.W = W
        return W

    def prepare_fold(self):
        """Folded weights of every encoder layer's q/k/v and fc1 (they follow self_attn_layer_norm / final_layer_norm):
        bf16(gamma . W), its row sums and beta W^T + b -- one small launch per matrix, cached until the weights change."""
        W, enc = self.W, self.enc
        if W.fold is not None:
            return W.fold
        dev = enc.conv1.weight.device
        D, F_ = self.cfg.d_model, self.cfg.encoder_ffn_dim
        fold = []
        for lyr in enc.layers:
            f = NS()
            att, ln1, ln2 = lyr.self_attn, lyr.self_attn_layer_norm, lyr.final_layer_norm
            f.qkv_w, f.qkv_c, f.qkv_b = _e((3 * D, D), BF16, dev), _e((3 * D,), F32, dev), _e((3 * D,), F32, dev)
            for k, (lin, has_b) in enumerate(((att.q_proj, True), (att.k_proj, False), (att.v_proj, True))):
                sl = slice(k * D, (k + 1) * D)
                ops.lnfold_prep(lin.weight.detach(), ln1.weight.detach(), ln1.bias.detach(), lin.bias.detach() if has_b else None,
                                f.qkv_w[sl], f.qkv_c[sl], f.qkv_b[sl])
            f.fc1_w, f.fc1_c, f.fc1_b = _e((F_, D), BF16, dev), _e((F_,), F32, dev), _e((F_,), F32, dev)
            ops.lnfold_prep(lyr.fc1.weight.detach(), ln2.weight.detach(), ln2.bias.detach(), lyr.fc1.bias.detach(), f.fc1_w, f.fc1_c, f.fc1_b)
            f.eps1, f.eps2 = float(ln1.eps), float(ln2.eps)
            fold.append(f)
        W.fold = fold
        return fold

    def forward(self, input_features, stno_mask, enrollments=None, need_grad=True):
        """Encoder forward.  Inference (need_grad False) of a large even batch: the two half batches run the WHOLE forward side by side on
        two streams (SPLIT_FWD above; each half allocates its temporaries under its own stream, the halves of the output are written in
        place) -- rows never mix in a forward kernel, so the output is bit-identical.  The training forward splits inside _forward_impl."""
        cfg = self.cfg
        B, T = input_features.shape[0], cfg.max_source_positions
        if (SPLIT_FWD and not need_grad and enrollments is None and not cfg.use_enrollments and input_features.is_cuda and B % 2 == 0
                and B * T >= SPLIT_FWD_MIN_ROWS and input_features.dim() == 3 and input_features.shape[2] == 2 * T
                and _split_allowed()):
            dev, D = input_features.device, cfg.d_model
            enc_out, enc_bf = _e((B * T, D), F32, dev), _e((B * T, D), BF16, dev)
            stno_mask = stno_mask.to(device=dev)
            main_st, parts, _ = _fwd_parts(B, T, dev)
            S = None
            for st_, r, bsl in parts:
                with torch.cuda.stream(st_):
                    _, Sh = self._forward_impl(input_features[bsl], stno_mask[bsl], None, False, out=(enc_out[r], enc_bf[r]))
                    del Sh.layers                        # (a side half's temporaries return to ITS stream's pool here, behind its own kernels)
                S = S or Sh
            for st_, _, _ in parts[1:]:
                main_st.wait_stream(st_)
            S.B0, S.B_out, S.enc_bf, S.h_last = B, B, enc_bf, None
            return enc_out.view(B, T, D), S
        return self._forward_impl(input_features, stno_mask, enrollments, need_grad)

    def _forward_impl(self, input_features, stno_mask, enrollments=None, need_grad=True, out=None):
        enc, cfg, W = self.enc, self.cfg, self.W
        dev = input_features.device
        if enrollments is not None:                      # encoder.py:152-154: interleave mixture / enrollment rows
            input_features = torch.stack((input_features, enrollments["input_features"].to(dev)), dim=1).flatten(0, 1)
            stno_mask = torch.stack((stno_mask, enrollments["stno_mask"].to(dev)), dim=1).flatten(0, 1)
        B, M, Tin = input_features.shape
        T, D, H, F_ = cfg.max_source_positions, cfg.d_model, cfg.encoder_attention_heads, cfg.encoder_ffn_dim
        if Tin != 2 * T:
            raise ValueError(f"Whisper expects the mel input features to be of length {2 * T}, but found {Tin}. "
                             f"Make sure to pad the input mel features to {2 * T}.")
        stno = stno_mask.to(device=dev, dtype=F32).contiguous()
        S = NS(B0=B, T=T, stno=stno, layers=[], scb=[])
        # ---- conv stem as two GEMMs over time-major views (encoder.py:167-170)
        x2, stem = conv_stem_fwd(enc, W, input_features, need_grad=need_grad)
        if need_grad:
            S.stem, S.x2 = stem, x2
        # ---- initial FDDT + positions (encoder.py:173-180)
        rows = B * T
        pos = enc.embed_positions.weight
        init_fddt = enc.initial_fddt if (cfg.use_fddt and cfg.use_pre_pos_fddt) else None
        mode, fw, fb = fddt_ptrs(init_fddt, cfg)
        h = _e((rows, D), F32, dev)
        if W.full_init is not None:                  # dense initial FDDT, then the positions
            hf0, S.x2b = full_fddt_fwd(W.full_init, x2, stno, 4 * T, rows, T, D)
            ops.fddt_ln_fwd(hf0, rows, D, mode=ops.MODE_NONE, T=T, pos=pos.detach(), h_out=h)
        else:
            ops.fddt_ln_fwd(x2, rows, D, mode=mode, stno=stno, T=T, w=fw, b=fb, pos=pos.detach(), h_out=h)
        bstride = 4 * T
        Bc = B
        # Inference forward: nothing reads the un-conditioned residual stream, and the diagonal FDDT is element-wise given a row's four
        # masks and a column's eight parameters -- layer i's fc2 epilogue writes FDDT_(i+1)(h) directly (DICOW_EPI_FDDT, bit-identical
        # to the row kernel) and layer i+1 starts with a plain LayerNorm.  Needs the row masks row-major: [rows (+ tile padding), 4].
        fuse_next = (FUSE_NEXT_FDDT and not need_grad and cfg.use_fddt and not cfg.use_enrollments and all(f is None for f in W.full)
                     and ops.gemm_nt(h, W.layers[0].fc2.w, h, rows, D, F_, residual=h, bias=W.layers[0].fc2.b, query_persistent=True))
        rowmask = None
        if fuse_next:
            rowmask = torch.zeros(_ceil(rows, 192) + 64, 4, dtype=F32, device=dev)
            rowmask[:rows].view(B, T, 4).copy_(stno.permute(0, 2, 1))
        # LayerNorm fold (inference forward): needs every GEMM of the layer on the persistent kernel and 320-wide column tiles
        fold = None
        if (LN_FOLD and not need_grad and not cfg.use_enrollments and all(f is None for f in W.full) and D % 320 == 0 and D <= 1280
                and ops.gemm_nt(h, W.layers[0].fc2.w, h, rows, D, F_, residual=h, bias=W.layers[0].fc2.b, query_lnstat=True)
                and ops.gemm_nt(h, W.layers[0].att.o.w, h, rows, D, D, residual=h, bias=W.layers[0].att.o.b, query_lnstat=True)
                and ops.gemm_nt(h, W.layers[0].att.qkv.w, h, rows, 3 * D, D, query_persistent=True)):
            fold = self.prepare_fold()
            stat1 = torch.zeros(rows, L.LN_SLOTS, 2, dtype=F32, device=dev)      # row partials of the LN1 / LN2 inputs
            stat2 = torch.zeros(rows, L.LN_SLOTS, 2, dtype=F32, device=dev)
        hb = None                                        # bf16 copy of h + its row partials in stat1: written by the previous fc2
        fddt_done = False                                # h already carries this layer's FDDT (written by the previous fc2)
        # training forward of a large even batch: two half batches on two streams into shared full-batch buffers (SPLIT_FWD above)
        # (SE-DiCoW: the speaker-communication layers pair mixture and enrollment rows and the enrollment rows are dropped behind the last
        # of them -- the fork comes behind that layer, over the 24 plain layers that follow)
        split_ok = (SPLIT_FWD and need_grad and fold is None and not fuse_next and all(f is None for f in W.full)
                    and _split_allowed())
        split_from = cfg.scb_layers if (cfg.use_enrollments and cfg.scb_layers) else 0
        halves = None
        for i, lyr in enumerate(enc.layers):
            w = W.layers[i]
            Ls = NS(h_in=h, B=Bc, bstride=bstride)
            rows = Bc * T
            if split_ok and halves is None and i == split_from and Bc % 2 == 0 and rows >= SPLIT_FWD_MIN_ROWS:
                S.split_from, S.split_sstep = i, bstride // (4 * T)   # (the split backward covers the same layers: their saved buffers are full-batch, no SCB)
                main_st, halves, Bh = _fwd_parts(Bc, T, dev)      # fork: h (stem + initial FDDT, or the last speaker-communication layer) is complete
                rh, sstep = Bh * T, bstride // (4 * T)             # (sstep: STNO rows per encoder row -- 2 once the enrollment rows are gone)
            if halves is not None:
                fd = enc.fddts[i] if (cfg.use_fddt and i < len(enc.fddts)) else None
                mode, fw, fb = fddt_ptrs(fd, cfg)
                ln, ln2 = lyr.self_attn_layer_norm, lyr.final_layer_norm
                xln, mean, rstd = _e((rows, D), BF16, dev), _e((rows,), F32, dev), _e((rows,), F32, dev)
                hp = _e((rows, D), F32, dev) if mode != ops.MODE_NONE else h
                qkv, o, lse = _e((rows, 3 * D), BF16, dev), _e((rows, D), BF16, dev), _e((Bc, H, T), F32, dev)
                h2, xln2 = _e((rows, D), F32, dev), _e((rows, D), BF16, dev)
                mean2, rstd2 = _e((rows,), F32, dev), _e((rows,), F32, dev)
                u, a, hn = _e((rows, F_), BF16, dev), _e((rows, F_), BF16, dev), _e((rows, D), F32, dev)
                for st_, r, bsl in halves:
                    with torch.cuda.stream(st_):
                        ops.fddt_ln_fwd(h[r], rh, D, mode=mode, stno=stno[bsl.start * sstep:], stno_bstride=bstride, T=T, w=fw, b=fb,
                                        h_out=hp[r] if mode != ops.MODE_NONE else None, ln_w=ln.weight.detach(),
                                        ln_b=ln.bias.detach(), y_bf16=xln[r], mean=mean[r], rstd=rstd[r])
                        linear_fwd(xln[r], w.att.qkv, rh, flags=L.EPI_SCALE_N, scale=Q_SCALE, scale_ncols=D, out=qkv[r])
                        ops.attn_fwd(heads(qkv[r][:, :D], Bh, T, H), heads(qkv[r][:, D:2 * D], Bh, T, H), heads(qkv[r][:, 2 * D:], Bh, T, H),
                                     heads(o[r], Bh, T, H), lse[bsl], q_log2=QK_LOG2)
                        linear_fwd(o[r], w.att.o, rh, out_dtype=F32, residual=hp[r], out=h2[r])
                        ops.fddt_ln_fwd(h2[r], rh, D, mode=ops.MODE_NONE, ln_w=ln2.weight.detach(), ln_b=ln2.bias.detach(), y_bf16=xln2[r],
                                        mean=mean2[r], rstd=rstd2[r])
                        linear_fwd(xln2[r], w.fc1, rh, gelu_aux=u[r], out=a[r])
                        linear_fwd(a[r], w.fc2, rh, out_dtype=F32, residual=h2[r], out=hn[r])
                h = hn
                Ls.rows, Ls.B_after, Ls.hp, Ls.xln, Ls.mean, Ls.rstd = rows, Bc, hp, xln, mean, rstd
                Ls.qkv, Ls.o, Ls.lse, Ls.h2, Ls.xln2, Ls.mean2, Ls.rstd2, Ls.u, Ls.a = qkv, o, lse, h2, xln2, mean2, rstd2, u, a
                S.layers.append(Ls)
                continue
            fd = enc.fddts[i] if (cfg.use_fddt and i < len(enc.fddts)) else None
            mode, fw, fb = fddt_ptrs(fd, cfg)
            if fddt_done:
                mode, fw, fb = ops.MODE_NONE, (None,) * 4, (None,) * 4
            nfd = enc.fddts[i + 1] if (fuse_next and i + 1 < len(enc.layers) and i + 1 < len(enc.fddts)) else None
            nmode, nfw, nfb = fddt_ptrs(nfd, cfg)
            fuse_here = nfd is not None and nmode == ops.MODE_DIAG and all(t is not None for t in nfw) and all(t is not None for t in nfb)
            use_scb = cfg.use_enrollments and cfg.scb_layers is not None and i < cfg.scb_layers
            wfull = W.full[i] if (fd is not None and i < len(W.full)) else None
            if wfull is not None:                    # dense FDDT: its own GEMM + combine, the rest of the layer sees "no FDDT"
                h, Ls.full_hb = full_fddt_fwd(wfull, h, stno, bstride, rows, T, D)
                Ls.h_in = h
                mode, fw, fb = ops.MODE_NONE, (None,) * 4, (None,) * 4
            xln = _e((rows, D), BF16, dev)
            mean, rstd = _e((rows,), F32, dev), _e((rows,), F32, dev)
            ln = lyr.self_attn_layer_norm
            fold1 = fold is not None and hb is not None and mode == ops.MODE_NONE      # LN1 folded: h, hb, stat1 came out of the previous fc2
            if fold1:
                hp = h
            elif not use_scb:
                hp = _e((rows, D), F32, dev) if mode != ops.MODE_NONE else h
                ops.fddt_ln_fwd(h, rows, D, mode=mode, stno=stno, stno_bstride=bstride, T=T, w=fw, b=fb,
                                h_out=hp if mode != ops.MODE_NONE else None, ln_w=ln.weight.detach(),
                                ln_b=ln.bias.detach(), y_bf16=xln, mean=mean, rstd=rstd)
            else:                                    # FDDT, then the speaker-communication block, then LayerNorm
                hf = _e((rows, D), F32, dev) if mode != ops.MODE_NONE else h
                if mode != ops.MODE_NONE:
                    ops.fddt_ln_fwd(h, rows, D, mode=mode, stno=stno, stno_bstride=bstride, T=T, w=fw, b=fb, h_out=hf)
                hs, scb_s = self._scb_fwd(i, hf, Bc, T)
                Ls.scb = scb_s
                Ls.hf = hf
                if i == cfg.scb_layers - 1:          # drop the enrollment rows (encoder.py:210-213): even rows only
                    Bc = Bc // 2
                    rows = Bc * T
                    hs = hs.view(Bc, 2, T, D)[:, 0].contiguous().view(rows, D)
                    bstride = 8 * T
                    xln = _e((rows, D), BF16, dev)
                    mean, rstd = _e((rows,), F32, dev), _e((rows,), F32, dev)
                    Ls.dropped = True
                hp = hs
                ops.fddt_ln_fwd(hs, rows, D, mode=ops.MODE_NONE, ln_w=ln.weight.detach(), ln_b=ln.bias.detach(),
                                y_bf16=xln, mean=mean, rstd=rstd)
            Ls.rows, Ls.B_after, Ls.hp, Ls.xln, Ls.mean, Ls.rstd = rows, Bc, hp, xln, mean, rstd
            # ---- self-attention (HF WhisperAttention; q pre-scaled in the projection epilogue, in base-2 units: Q_SCALE above)
            if fold1:
                fl = fold[i]
                qkv = _e((rows, 3 * D), BF16, dev)
                ops.gemm_nt(hb, fl.qkv_w, qkv, rows, 3 * D, D, bias=fl.qkv_b, flags=L.EPI_SCALE_N, scale=Q_SCALE, scale_ncols=D,
                            ln_fold=(stat1, fl.qkv_c, D, fl.eps1))
            else:
                qkv = linear_fwd(xln, w.att.qkv, rows, flags=L.EPI_SCALE_N, scale=Q_SCALE, scale_ncols=D)
            o = _e((rows, D), BF16, dev)
            lse = _e((Bc, H, T), F32, dev)
            ops.attn_fwd(heads(qkv[:, :D], Bc, T, H), heads(qkv[:, D:2 * D], Bc, T, H), heads(qkv[:, 2 * D:], Bc, T, H),
                         heads(o, Bc, T, H), lse, q_log2=QK_LOG2)
            if fold is not None:                                  # LN2 folded: out-proj also writes bf16(h2) and its row partials
                h2, h2b = _e((rows, D), F32, dev), _e((rows, D), BF16, dev)
                ops.gemm_nt(o, w.att.o.w, h2, rows, D, D, bias=w.att.o.b, residual=hp, ln_stat=(h2b, stat2))
                fl = fold[i]
                a = _e((rows, F_), BF16, dev)
                ops.gemm_nt(h2b, fl.fc1_w, a, rows, F_, D, bias=fl.fc1_b, flags=L.EPI_GELU, ln_fold=(stat2, fl.fc1_c, D, fl.eps2))
                # the next layer's LN1 can be folded too when fc2 writes its final input: its FDDT fused here, or no FDDT at all
                nxt_plain = i + 1 < len(enc.layers) and not (cfg.use_fddt and i + 1 < len(enc.fddts))
                hb = _e((rows, D), BF16, dev) if (i + 1 < len(enc.layers) and (fuse_here or nxt_plain)) else None
                h = _e((rows, D), F32, dev)
                ops.gemm_nt(a, w.fc2.w, h, rows, D, F_, bias=w.fc2.b, residual=h2,
                            fddt=(tuple(t.detach() for t in nfw), tuple(t.detach() for t in nfb), rowmask) if fuse_here else None,
                            ln_stat=(hb, stat1) if hb is not None else None)
                fddt_done = fuse_here
                continue
            h2 = linear_fwd(o, w.att.o, rows, out_dtype=F32, residual=hp)
            # ---- feed-forward
            ln2 = lyr.final_layer_norm
            xln2 = _e((rows, D), BF16, dev)
            mean2, rstd2 = _e((rows,), F32, dev), _e((rows,), F32, dev)
            ops.fddt_ln_fwd(h2, rows, D, mode=ops.MODE_NONE, ln_w=ln2.weight.detach(), ln_b=ln2.bias.detach(), y_bf16=xln2,
                            mean=mean2, rstd=rstd2)
            if need_grad:
                u = _e((rows, F_), BF16, dev)                     # gelu'(pre-activation), consumed by the backward pass
                a = linear_fwd(xln2, w.fc1, rows, gelu_aux=u)
            else:
                u, a = None, linear_fwd(xln2, w.fc1, rows, flags=L.EPI_GELU)
            h = linear_fwd(a, w.fc2, rows, out_dtype=F32, residual=h2,
                           fddt=(tuple(t.detach() for t in nfw), tuple(t.detach() for t in nfb), rowmask) if fuse_here else None)
            fddt_done = fuse_here
            if need_grad:                                         # inference: nothing is kept, buffers recycle per layer
                Ls.qkv, Ls.o, Ls.lse, Ls.h2, Ls.xln2, Ls.mean2, Ls.rstd2, Ls.u, Ls.a = qkv, o, lse, h2, xln2, mean2, rstd2, u, a
                S.layers.append(Ls)
        rows = Bc * T
        S.h_last, S.B_out, S.bstride_out = h, Bc, bstride
        enc_out, enc_bf = out if out is not None else (_e((rows, D), F32, dev), _e((rows, D), BF16, dev))
        S.meanf, S.rstdf = _e((rows,), F32, dev), _e((rows,), F32, dev)
        if halves is not None:                           # the final LayerNorm per half too, then the join
            for st_, r, bsl in halves:
                with torch.cuda.stream(st_):
                    ops.fddt_ln_fwd(h[r], rh, D, mode=ops.MODE_NONE, ln_w=enc.layer_norm.weight.detach(), ln_b=enc.layer_norm.bias.detach(),
                                    y_bf16=enc_bf[r], y_f32=enc_out[r], mean=S.meanf[r], rstd=S.rstdf[r])
            for st_, _, _ in halves[1:]:
                main_st.wait_stream(st_)
        else:
            ops.fddt_ln_fwd(h, rows, D, mode=ops.MODE_NONE, ln_w=enc.layer_norm.weight.detach(), ln_b=enc.layer_norm.bias.detach(),
                            y_bf16=enc_bf, y_f32=enc_out, mean=S.meanf, rstd=S.rstdf)
        S.enc_bf = enc_bf
        return enc_out.view(Bc, T, D), S

    # -- speaker communication block (layers.py:145-193) on interleaved rows: even = mixture, odd = enrollment
    def _scb_fwd(self, i, hf, Bc, T):
        enc, cfg = self.enc, self.cfg
        w = self.W.scb[i]
        blk = enc.ca_enrolls[i].cae
        D, H, F_ = cfg.d_model, cfg.encoder_attention_heads, cfg.encoder_ffn_dim
        dev = hf.device
        rows = Bc * T
        Bp = Bc // 2
        rp = Bp * T
        # de-interleave + cast in one pass (ops.scb_split): mixture rows -> q_in, enrollment rows -> kv_in, and the right half of
        # `cat` = q_in ([attn_output | q], layers.py:161)
        q_in, kv_in = _e((rp, D), BF16, dev), _e((rp, D), BF16, dev)
        cat = _e((rp, 2 * D), BF16, dev)
        ops.scb_split(hf, q_in, kv_in, cat, Bp, T, D)
        q = linear_fwd(q_in, w.att.q, rp, flags=L.EPI_SCALE_N, scale=Q_SCALE, scale_ncols=D)
        kv = linear_fwd(kv_in, w.att.kv, rp)
        o = _e((rp, D), BF16, dev)
        lse = _e((Bp, H, T), F32, dev)
        ops.attn_fwd(heads(q, Bp, T, H), heads(kv[:, :D], Bp, T, H), heads(kv[:, D:], Bp, T, H), heads(o, Bp, T, H), lse, q_log2=QK_LOG2)
        ops.gemm_nt(o, w.att.o.w, cat, rp, D, D, ldc=2 * D, bias=w.att.o.b)
        u = _e((rp, F_), BF16, dev)
        a = linear_fwd(cat, w.f0, rp, gelu_aux=u)
        upd = linear_fwd(a, w.f3, rp)                                       # bf16 [rp, D]
        out = _e((rows, D), F32, dev)
        ops.scb_merge_fwd(hf, upd, blk.cross_gate.gate.detach(), out, Bp, T, D)      # out = hf, mixture rows += tanh(gate) * upd
        s = NS(q_in=q_in, kv_in=kv_in, q=q, kv=kv, o=o, lse=lse, cat=cat, u=u, a=a, upd=upd, Bp=Bp)
        return out, s

    def _scb_bwd(self, i, s, g, G, T):
        """g: fp32 grad wrt SCB output [rows, D] (interleaved).  Returns grad wrt SCB input (fp32, same shape)."""
        enc, cfg = self.enc, self.cfg
        w = self.W.scb[i]
        blk = enc.ca_enrolls[i].cae
        D, H, F_ = cfg.d_model, cfg.encoder_attention_heads, cfg.encoder_ffn_dim
        Bp, rp = s.Bp, s.Bp * T
        dev = g.device
        gate_p = blk.cross_gate.gate
        d_upd = _e((rp, D), BF16, dev)
        ops.scb_gate_bwd(g, s.upd, gate_p.detach(), d_upd, G.get(gate_p), Bp, T, D)   # d_upd = bf16(g_mix * tanh(gate)); gate gradient
        tng = ops.TnGroup()                      # the block's six weight gradients: one pooled launch (400 tiles at large-v3-turbo)
        bias_grad(d_upd, G.get(blk.ffn[3].bias))
        linear_wgrad(d_upd, s.a, G.get(blk.ffn[3].weight), rp, group=tng)
        d_u = linear_dgrad(d_upd, w.f3, rp, aux=s.u)
        bias_grad(d_u, G.get(blk.ffn[0].bias))
        linear_wgrad(d_u, s.cat, G.get(blk.ffn[0].weight), rp, group=tng)
        d_cat = linear_dgrad(d_u, w.f0, rp)                                  # [rp, 2D]
        d_attn = d_cat[:, :D]
        att = blk.cross_attn
        bias_grad(d_attn, G.get(att.out_proj.bias))
        linear_wgrad(d_attn, s.o, G.get(att.out_proj.weight), rp, group=tng)
        d_o = _e((rp, D), BF16, dev)
        ops.gemm_nt(d_attn, w.att.o.wt, d_o, rp, D, D, lda=2 * D)
        dq = _e((rp, D), BF16, dev)
        dkv = _e((rp, 2 * D), BF16, dev)
        delta = _e((2, Bp, H, T), F32, dev)
        ops.attn_bwd(heads(s.q, Bp, T, H), heads(s.kv[:, :D], Bp, T, H), heads(s.kv[:, D:], Bp, T, H), heads(s.o, Bp, T, H),
                     heads(d_o, Bp, T, H), s.lse, delta, heads(dq, Bp, T, H), heads(dkv[:, :D], Bp, T, H),
                     heads(dkv[:, D:], Bp, T, H), dq_scale=0.125, q_log2=QK_LOG2)
        bias_grad(dq, G.get(att.q_proj.bias))
        bias_grad(dkv[:, D:], G.get(att.v_proj.bias))
        linear_wgrad(dq, s.q_in, G.get(att.q_proj.weight), rp, group=tng)
        linear_wgrad(dkv[:, :D], s.kv_in, G.get(att.k_proj.weight), rp, group=tng)
        linear_wgrad(dkv[:, D:], s.kv_in, G.get(att.v_proj.weight), rp, group=tng)
        tng.run()
        d_qin = linear_dgrad(dq, w.att.q, rp, out_dtype=F32)
        d_kvin = linear_dgrad(dkv, w.att.kv, rp, out_dtype=F32)
        gin = _e((2 * rp, D), F32, dev)
        ops.scb_merge_bwd(g, d_qin, d_cat, d_kvin, gin, Bp, T, D)            # gin = g (+ d_qin + d_cat[:, D:] | + d_kvin)
        return gin

    # -- backward
    def backward(self, S, d_enc, G):
        """d_enc: fp32 [B_out*T, D] gradient wrt encoder_last_hidden_state.  Accumulates parameter grads into G."""
        enc, cfg, W = self.enc, self.cfg, self.W
        dev = d_enc.device
        T, D, H, F_ = S.T, cfg.d_model, cfg.encoder_attention_heads, cfg.encoder_ffn_dim
        rows = S.B_out * T
        nl = len(enc.layers)
        g = _e((rows, D), F32, dev)
        gb = _e((rows, D), BF16, dev)
        last = enc.layers[nl - 1]
        # trainer.SplitSync (the step as two half batches on two streams): the FOLLOWER half waits, segment by segment, until the leader
        # half has enqueued that segment's parameter-gradient writes -- the two halves' sums then land in a fixed order
        sync = getattr(enc, "_split_sync", None)
        enter = sync.enter if sync is not None else (lambda name: None)
        enter("final_ln")
        ops.fddt_ln_bwd(S.h_last, rows, D, mode=ops.MODE_NONE, ln_w=enc.layer_norm.weight.detach(), mean=S.meanf, rstd=S.rstdf,
                        d_y=d_enc.contiguous(), g_out=g, g_out_bf16=gb, dln_w=G.get(enc.layer_norm.weight),
                        dln_b=G.get(enc.layer_norm.bias), colsum_out=G.get(last.fc2.bias))
        seg_hook = getattr(enc, "_segment_hook", None) or (lambda name: None)

        def hook(name):
            seg_hook(name)
            if sync is not None:
                sync.done(name)
        hook("final_ln")
        start = nl - 1
        if (SPLIT_BWD and SPLIT_FWD and getattr(S, "split_from", None) is not None and (sync is None or sync.role is None) and not WGRAD_SIDE_STREAM
                and S.B_out % 2 == 0 and _split_allowed() and self._split_bwd_fits(S, d_enc.device)):
            g, gb = self._backward_split(S, g, gb, G, hook)      # layers nl-1 .. split_from on two streams
            start = S.split_from - 1                             # (SE-DiCoW: the speaker-communication layers below follow on one stream)
            if start < 0:
                enter("stem")
                self._stem_backward(S, g, G)
                hook("stem")
                return
        # Weight gradients are recorded per layer and run as ONE pooled launch (ops.TnGroup / dicow_gemm_tn_group) as soon as
        # the pool holds a tile for every CU: large-v3-turbo has 300 tiles per layer (one launch per layer), whisper-base 48
        # (all six layers in one launch at the end).  A layer's DP bucket is only handed over once its gradients have run.
        tng, pend, ncu, per_layer = ops.TnGroup(), [], ops.num_cus(dev), 0
        side = wgrad_stream(dev) if (WGRAD_SIDE_STREAM and d_enc.is_cuda and not torch.cuda.is_current_stream_capturing()) else None

        def run_wgrads(names):
            if side is None or not tng.items:
                tng.run()
                for name in names:
                    hook(name)
                return
            ready = torch.cuda.Event()
            ready.record()                                    # every operand of the recorded problems has been enqueued
            held = [t for it in tng.items for t in (it[0], it[1], it[2])] + ([G.flat] if G.flat.numel() > 1 else [])
            with torch.cuda.stream(side):
                side.wait_event(ready)
                for t in held:                                # allocated on the compute stream, read here: no reuse before this stream is through
                    if t is not None and t.is_cuda:
                        t.record_stream(side)
                tng.run()
                for name in names:                            # (the DP bucket / the split-step event are recorded behind the launch, on this stream)
                    hook(name)
        for i in range(start, -1, -1):
            n_before = len(tng.items)
            enter(f"layer{i}")
            lyr, w, Ls = enc.layers[i], W.layers[i], S.layers[i]
            rows, Bc = Ls.rows, Ls.B_after
            # ---- FFN backward (the weight-gradient operands gb, d_u, g2b, d_qkv and the saved activations live on until tng.run())
            linear_wgrad(gb, Ls.a, G.get(lyr.fc2.weight), rows, group=tng, param=lyr.fc2.weight)
            d_u = linear_dgrad(gb, w.fc2, rows, aux=Ls.u, colsum_out=G.get(lyr.fc1.bias))     # fc1 bias grad = colsum(d_u), fused
            linear_wgrad(d_u, Ls.xln2, G.get(lyr.fc1.weight), rows, group=tng, param=lyr.fc1.weight)
            d_xln2 = linear_dgrad(d_u, w.fc1, rows)
            g2 = _e((rows, D), F32, dev)
            g2b = _e((rows, D), BF16, dev)
            ln2 = lyr.final_layer_norm
            ops.fddt_ln_bwd(Ls.h2, rows, D, mode=ops.MODE_NONE, ln_w=ln2.weight.detach(), mean=Ls.mean2, rstd=Ls.rstd2,
                            d_y=d_xln2, g_res=g, g_out=g2, g_out_bf16=g2b, dln_w=G.get(ln2.weight), dln_b=G.get(ln2.bias),
                            colsum_out=G.get(lyr.self_attn.out_proj.bias))
            # ---- attention backward
            att = lyr.self_attn
            linear_wgrad(g2b, Ls.o, G.get(att.out_proj.weight), rows, group=tng, param=att.out_proj.weight)
            d_o = linear_dgrad(g2b, w.att.o, rows)
            d_qkv = _e((rows, 3 * D), BF16, dev)
            delta = _e((2, Bc, H, T), F32, dev)
            qkv = Ls.qkv
            ops.attn_bwd(heads(qkv[:, :D], Bc, T, H), heads(qkv[:, D:2 * D], Bc, T, H), heads(qkv[:, 2 * D:], Bc, T, H),
                         heads(Ls.o, Bc, T, H), heads(d_o, Bc, T, H), Ls.lse, delta, heads(d_qkv[:, :D], Bc, T, H),
                         heads(d_qkv[:, D:2 * D], Bc, T, H), heads(d_qkv[:, 2 * D:], Bc, T, H), dq_scale=0.125, q_log2=QK_LOG2,
                         dq_colsum=G.get(att.q_proj.bias), dv_colsum=G.get(att.v_proj.bias))   # q / v bias grads, fused
            qkv_wgrad(d_qkv, Ls.xln, G.get(att.q_proj.weight), G.get(att.k_proj.weight), G.get(att.v_proj.weight), rows, D, group=tng,
                      params=(att.q_proj.weight, att.k_proj.weight, att.v_proj.weight))
            d_xln = linear_dgrad(d_qkv, w.att.qkv, rows)
            # ---- LayerNorm1 (+ SCB) + FDDT backward; the column sum of the result is the previous fc2's bias grad
            fd = enc.fddts[i] if (cfg.use_fddt and i < len(enc.fddts)) else None
            mode, fw, fb = fddt_ptrs(fd, cfg)
            dw = tuple(G.get(x) for x in fw)
            db = tuple(G.get(x) for x in fb)
            ln = lyr.self_attn_layer_norm
            prev_b2 = G.get(enc.layers[i - 1].fc2.bias) if i > 0 else None
            rows_in = Ls.B * T
            g0 = _e((rows_in, D), F32, dev)
            g0b = _e((rows_in, D), BF16, dev) if i > 0 else None
            wfull = W.full[i] if (fd is not None and i < len(W.full)) else None
            if wfull is not None:                    # dense FDDT: LayerNorm (+ SCB) backward first, then its own backward
                gf = _e((rows, D), F32, dev)
                ops.fddt_ln_bwd(Ls.hp, rows, D, mode=ops.MODE_NONE, ln_w=ln.weight.detach(), mean=Ls.mean, rstd=Ls.rstd,
                                d_y=d_xln, g_res=g2, g_out=gf, dln_w=G.get(ln.weight), dln_b=G.get(ln.bias))
                if hasattr(Ls, "scb"):
                    if getattr(Ls, "dropped", False):
                        full = torch.zeros(rows_in, D, dtype=F32, device=dev)
                        full.view(Ls.B // 2, 2, T, D)[:, 0].copy_(gf.view(Ls.B // 2, T, D))
                        gf = full
                    gf = self._scb_bwd(i, Ls.scb, gf, G, T)
                g0 = full_fddt_bwd(fd, wfull, Ls.full_hb, gf, S.stno, Ls.bstride, G, rows_in, T, D)
                if i > 0:
                    g0b = ops.cast_bf16(g0)
                    if prev_b2 is not None:
                        ops.colsum_bf16(g0b, prev_b2)
            elif not hasattr(Ls, "scb"):
                ops.fddt_ln_bwd(Ls.h_in, rows, D, mode=mode, stno=S.stno, stno_bstride=Ls.bstride, T=T, w=fw, b=fb,
                                ln_w=ln.weight.detach(), mean=Ls.mean, rstd=Ls.rstd, d_y=d_xln, g_res=g2, g_out=g0,
                                g_out_bf16=g0b, dln_w=G.get(ln.weight), dln_b=G.get(ln.bias), dw=dw, db=db,
                                colsum_out=prev_b2)
            else:
                gs = _e((rows, D), F32, dev)
                ops.fddt_ln_bwd(Ls.hp, rows, D, mode=ops.MODE_NONE, ln_w=ln.weight.detach(), mean=Ls.mean, rstd=Ls.rstd,
                                d_y=d_xln, g_res=g2, g_out=gs, dln_w=G.get(ln.weight), dln_b=G.get(ln.bias))
                if getattr(Ls, "dropped", False):    # re-interleave: enrollment rows receive zero gradient
                    full = torch.zeros(rows_in, D, dtype=F32, device=dev)
                    full.view(Ls.B // 2, 2, T, D)[:, 0].copy_(gs.view(Ls.B // 2, T, D))
                    gs = full
                gin = self._scb_bwd(i, Ls.scb, gs, G, T)
                ops.fddt_ln_bwd(Ls.h_in, rows_in, D, mode=mode, stno=S.stno, stno_bstride=Ls.bstride, T=T, w=fw, b=fb,
                                g_res=gin, g_out=g0, g_out_bf16=g0b, dw=dw, db=db, colsum_out=prev_b2)
            g, gb = g0, g0b
            pend.append(f"layer{i}")
            per_layer = max(per_layer, len(tng.items) - n_before)      # problems a layer records (4 with fused q/k/v, else 6)
            if i == 0 or tng.tiles() >= ncu or len(tng.items) + per_layer > L.TN_GROUP_MAX:
                run_wgrads(pend)
                pend = []
        enter("stem")
        self._stem_backward(S, g, G)
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)  # every dW is written before autograd / the optimizer / the "stem" bucket go on
        hook("stem")

    # -- the layers' backward as two half batches on two streams (the forward forked the same way: every saved buffer is full-batch)
    def _split_bwd_fits(self, S, dev):
        """The split backward keeps every layer's gradient buffers until the pass ends (no block is handed from one stream to the other mid-pass):
        ~40 bytes per row and model column per layer (31 GB at B = 16 for large-v3-turbo).  A batch too large for that falls back to the one-stream
        backward, which frees a layer's buffers as it goes."""
        cfg = self.cfg
        nl = len(self.enc.layers) - S.split_from
        need = nl * S.B_out * S.T * (2 * cfg.encoder_ffn_dim + 24 * cfg.d_model + 64)
        if not dev.type == "cuda":
            return True
        total = getattr(self, "_dev_total_mem", None)
        if total is None:
            total = self._dev_total_mem = torch.cuda.get_device_properties(dev).total_memory
        free = total - torch.cuda.memory_allocated(dev)          # (the allocator's own books: no driver call, nothing that could wait for the device)
        return need < SPLIT_BWD_MEM_FRACTION * free

    def _shadow(self, G, dev):
        """Zeroed fp32 stand-ins for the VECTOR gradients of the layers (LayerNorm affine, biases, diagonal FDDT): the second half's
        row reductions land here and are added to the real gradients layer by layer, behind the first half's -- a fixed order."""
        enc, cfg = self.enc, self.cfg
        sh = getattr(self, "_sh", None)
        if sh is None or sh.flat.device != dev:
            per, off = [], 0
            for i, lyr in enumerate(enc.layers):
                ps = [p for p in lyr.parameters() if p.dim() == 1]
                fd = enc.fddts[i] if (cfg.use_fddt and i < len(enc.fddts)) else None
                if fd is not None:
                    ps += [p for p in fd.parameters() if p.dim() == 1]
                per.append(ps)
                off += sum(_ceil(p.numel(), 64) for p in ps)
            sh = NS(flat=torch.zeros(max(off, 1), dtype=F32, device=dev), view={}, per=per)
            off = 0
            for ps in per:
                for p in ps:
                    sh.view[id(p)] = sh.flat[off:off + p.numel()].view(p.shape)
                    off += _ceil(p.numel(), 64)
            self._sh = sh
        sh.flat.zero_()

        def get(p):
            if p is None or G.get(p) is None:
                return None
            return sh.view[id(p)]                         # (KeyError = a reduction target this scheme does not know: fail loudly)
        return sh, get

    def _backward_split(self, S, g, gb, G, hook):
        """Layers nl-1 .. 0 with the row-parallel chain (dgrads, attention backward, row kernels) of the two half batches on two streams.
        What is NOT row-parallel keeps one owner and one order:
          * weight matrices: the layer's pooled TN launch over the FULL batch (both halves write the halves of the same operand buffers),
            launched one layer late -- behind an event of the other half -- so that no stream ever waits for work not yet enqueued;
            same kernel, same operands as the one-stream backward: bit-equal;
          * vectors (biases, LayerNorm affine, diagonal FDDT: the fused column / row reductions of the chain's kernels): half 0 adds into
            the gradient itself, half 1 into zeroed stand-ins (_shadow) that are added behind it, layer by layer, before the layer's
            DP bucket leaves: a fixed order, bit-reproducible; equal to the one-stream sums to rounding (a sum in two pieces).
        Every buffer of the pass lives until its end (no block freed here is handed to the other stream mid-pass).  Covers layers nl-1 .. S.split_from
        (0 for a plain encoder); returns (g, gb) of the lowest one."""
        enc, cfg, W = self.enc, self.cfg, self.W
        dev = g.device
        T, D, H, F_ = S.T, cfg.d_model, cfg.encoder_attention_heads, cfg.encoder_ffn_dim
        nl = len(enc.layers)
        Bc = S.B_out
        rows = Bc * T
        main_st, parts, Bh = _fwd_parts(Bc, T, dev)             # fork: g / gb (final LayerNorm backward) are complete
        side_st = parts[1][0]
        rh = Bh * T
        sh, shget = self._shadow(G, dev)
        side_st.wait_stream(main_st)                             # (the zero fill of the stand-ins was enqueued behind the fork above)
        keep, pending = [], None                                 # pending = (layer, its TnGroup, its stream's partner event)

        third_st = wgrad_stream(dev) if SPLIT_BWD_WGRAD == "third" else None
        deferred = []                                            # ("third": layers whose weight gradients run on the third stream, not yet settled)

        def settle(item):                                        # ("third") one layer later still: the stand-in sums and the DP bucket behind the third stream's launch
            i, ev_side, ev3 = item
            main_st.wait_event(ev_side)
            main_st.wait_event(ev3)
            dst = [G.get(p) for p in sh.per[i]]
            src = [sh.view[id(p)] for p, d in zip(sh.per[i], dst) if d is not None]
            dst = [d for d in dst if d is not None]
            if dst:
                torch._foreach_add_(dst, src)
            hook(f"layer{i}")

        def finish(item):                                        # a layer whose chain is enqueued on both streams: weight gradients, stand-in sums, DP bucket
            i, tg, ev_main, ev_side = item
            if third_st is not None:                             # (experiment: every pooled launch on a third queue, neither chain ever carries one)
                with torch.cuda.stream(third_st):
                    third_st.wait_event(ev_main)
                    third_st.wait_event(ev_side)
                    tg.run()
                    ev3 = third_st.record_event()
                deferred.append((i, ev_side, ev3))
                if len(deferred) > 1:
                    settle(deferred.pop(0))
                return
            if SPLIT_BWD_WGRAD == "alt" and (i & 1):             # (experiment: odd layers' pooled launch on the side stream, to balance the streams)
                with torch.cuda.stream(side_st):
                    side_st.wait_event(ev_main)
                    tg.run()
                    ev_side = side_st.record_event()
                main_st.wait_event(ev_side)
            else:
                main_st.wait_event(ev_side)
                tg.run()
            dst = [G.get(p) for p in sh.per[i]]
            src = [sh.view[id(p)] for p, d in zip(sh.per[i], dst) if d is not None]
            dst = [d for d in dst if d is not None]
            if dst:
                torch._foreach_add_(dst, src)
            hook(f"layer{i}")

        lo, sstep = S.split_from, S.split_sstep
        for i in range(nl - 1, lo - 1, -1):
            lyr, w, Ls = enc.layers[i], W.layers[i], S.layers[i]
            att, ln, ln2 = lyr.self_attn, lyr.self_attn_layer_norm, lyr.final_layer_norm
            fd = enc.fddts[i] if (cfg.use_fddt and i < len(enc.fddts)) else None
            mode, fw, fb = fddt_ptrs(fd, cfg)
            prev_b2 = enc.layers[i - 1].fc2.bias if i > 0 else None
            d_u, d_xln2 = _e((rows, F_), BF16, dev), _e((rows, D), BF16, dev)
            g2, g2b, d_o = _e((rows, D), F32, dev), _e((rows, D), BF16, dev), _e((rows, D), BF16, dev)
            d_qkv, d_xln = _e((rows, 3 * D), BF16, dev), _e((rows, D), BF16, dev)
            g0, g0b = _e((rows, D), F32, dev), (_e((rows, D), BF16, dev) if i > 0 else None)
            keep += [d_u, d_xln2, g2, g2b, d_o, d_qkv, d_xln, g0, g0b, g, gb]
            qkv = Ls.qkv
            evs = []
            for k, (st_, r, bsl) in enumerate(parts):
                gg = G.get if k == 0 else shget
                with torch.cuda.stream(st_):
                    linear_dgrad(gb[r], w.fc2, rh, aux=Ls.u[r], colsum_out=gg(lyr.fc1.bias), out=d_u[r])
                    linear_dgrad(d_u[r], w.fc1, rh, out=d_xln2[r])
                    ops.fddt_ln_bwd(Ls.h2[r], rh, D, mode=ops.MODE_NONE, ln_w=ln2.weight.detach(), mean=Ls.mean2[r], rstd=Ls.rstd2[r],
                                    d_y=d_xln2[r], g_res=g[r], g_out=g2[r], g_out_bf16=g2b[r], dln_w=gg(ln2.weight), dln_b=gg(ln2.bias),
                                    colsum_out=gg(att.out_proj.bias))
                    linear_dgrad(g2b[r], w.att.o, rh, out=d_o[r])
                    delta = _e((2, Bh, H, T), F32, dev)
                    ops.attn_bwd(heads(qkv[r][:, :D], Bh, T, H), heads(qkv[r][:, D:2 * D], Bh, T, H), heads(qkv[r][:, 2 * D:], Bh, T, H),
                                 heads(Ls.o[r], Bh, T, H), heads(d_o[r], Bh, T, H), Ls.lse[bsl], delta, heads(d_qkv[r][:, :D], Bh, T, H),
                                 heads(d_qkv[r][:, D:2 * D], Bh, T, H), heads(d_qkv[r][:, 2 * D:], Bh, T, H), dq_scale=0.125, q_log2=QK_LOG2,
                                 dq_colsum=gg(att.q_proj.bias), dv_colsum=gg(att.v_proj.bias))
                    linear_dgrad(d_qkv[r], w.att.qkv, rh, out=d_xln[r])
                    ops.fddt_ln_bwd(Ls.h_in[r], rh, D, mode=mode, stno=S.stno[bsl.start * sstep:], stno_bstride=Ls.bstride, T=T, w=fw, b=fb,
                                    ln_w=ln.weight.detach(), mean=Ls.mean[r], rstd=Ls.rstd[r], d_y=d_xln[r], g_res=g2[r], g_out=g0[r],
                                    g_out_bf16=g0b[r] if g0b is not None else None, dln_w=gg(ln.weight), dln_b=gg(ln.bias),
                                    dw=tuple(gg(x) for x in fw), db=tuple(gg(x) for x in fb), colsum_out=gg(prev_b2))
                    keep.append(delta)
                    evs.append(st_.record_event())
            # the layer's weight gradients: recorded now (first-writer flags consumed in layer order), launched one layer late
            tg = ops.TnGroup()
            linear_wgrad(gb, Ls.a, G.get(lyr.fc2.weight), rows, group=tg, param=lyr.fc2.weight)
            linear_wgrad(d_u, Ls.xln2, G.get(lyr.fc1.weight), rows, group=tg, param=lyr.fc1.weight)
            linear_wgrad(g2b, Ls.o, G.get(att.out_proj.weight), rows, group=tg, param=att.out_proj.weight)
            qkv_wgrad(d_qkv, Ls.xln, G.get(att.q_proj.weight), G.get(att.k_proj.weight), G.get(att.v_proj.weight), rows, D, group=tg,
                      params=(att.q_proj.weight, att.k_proj.weight, att.v_proj.weight))
            if pending is not None:
                finish(pending)
            pending = (i, tg, evs[0], evs[1])
            g, gb = g0, g0b
        finish(pending)
        while deferred:
            settle(deferred.pop(0))
        main_st.wait_stream(side_st)
        if third_st is not None:
            main_st.wait_stream(third_st)
        if lo > 0:                                               # layer lo's row kernel reduced into layer lo-1's fc2 bias: half 1's share joins it here
            pb = enc.layers[lo - 1].fc2.bias
            if G.get(pb) is not None:
                G.get(pb).add_(sh.view[id(pb)])
        del keep
        return g, gb

    def _stem_backward(self, S, g, G):
        enc, cfg, W = self.enc, self.cfg, self.W
        dev = g.device
        T, D = S.T, cfg.d_model
        # ---- positions + initial FDDT + conv stem (encoder.py:167-180)
        B, Tin, M = S.B0, 2 * T, cfg.num_mel_bins
        rows = B * T
        gpos = G.get(enc.embed_positions.weight)
        if gpos is not None:
            ops.sum_over_batch(g.view(B, T * D), gpos)
        conv_train = enc.conv1.weight.requires_grad or enc.conv2.weight.requires_grad or enc.conv1.bias.requires_grad
        init_fddt = enc.initial_fddt if (cfg.use_fddt and cfg.use_pre_pos_fddt) else None
        mode, fw, fb = fddt_ptrs(init_fddt, cfg)
        dw = tuple(G.get(x) for x in fw)
        db = tuple(G.get(x) for x in fb)
        if not conv_train and all(x is None for x in dw + db):
            return
        if W.full_init is not None:
            d_x2 = ops.cast_bf16(full_fddt_bwd(init_fddt, W.full_init, S.x2b, g, S.stno, 4 * T, G, rows, T, D))
        else:
            d_x2 = _e((rows, D), BF16, dev)
            ops.fddt_ln_bwd(S.x2, rows, D, mode=mode, stno=S.stno, T=T, w=fw, b=fb, g_res=g, g_out_bf16=d_x2, dw=dw, db=db)
        if not conv_train:
            return
        conv_stem_bwd(enc, W, S.stem, d_x2, G)


# ------------------------------------------------------------------------------------------------ decoder + LM head + loss
class DecoderEngine:
    def __init__(self, model):
        self.model = model                      # DiCoWForConditionalGeneration
        self.cfg = model.config

    def prepare(self):
        dec, cfg = self.model.model.decoder, self.cfg
        dev = dec.embed_tokens.weight.device
        W = NS(layers=[])
        for lyr in dec.layers:
            w = NS()
            w.sa = prep_attention(lyr.self_attn, dev)
            w.ca = prep_attention(lyr.encoder_attn, dev, fuse_qkv=False)
            w.fc1 = prep_linear([lyr.fc1.weight], [lyr.fc1.bias], dev)
            w.fc2 = prep_linear([lyr.fc2.weight], [lyr.fc2.bias], dev)
            W.layers.append(w)
        W.vpad = _ceil(cfg.vocab_size, 128)
        W.head = prep_linear([self.model.proj_out.weight], None, dev, n_pad=W.vpad)
        self.W = W
        return W

    def _layers_fwd(self, enc_bf, B, T, Lq, h):
        """The decoder layers on B sequences: h fp32 [B*Lq, D] (embeddings) -> (last hidden state, per-layer saved activations)."""
        model, cfg, W = self.model, self.cfg, self.W
        dec = model.model.decoder
        dev = enc_bf.device
        D, H, F_ = cfg.d_model, cfg.decoder_attention_heads, cfg.decoder_ffn_dim
        rows = B * Lq
        out = []
        for i, lyr in enumerate(dec.layers):
            w = W.layers[i]
            Ls = NS(h_in=h)
            # causal self-attention
            ln = lyr.self_attn_layer_norm
            Ls.x1, Ls.m1, Ls.r1 = _e((rows, D), BF16, dev), _e((rows,), F32, dev), _e((rows,), F32, dev)
            ops.fddt_ln_fwd(h, rows, D, ln_w=ln.weight.detach(), ln_b=ln.bias.detach(), y_bf16=Ls.x1, mean=Ls.m1, rstd=Ls.r1)
            Ls.qkv = linear_fwd(Ls.x1, w.sa.qkv, rows, flags=L.EPI_SCALE_N, scale=Q_SCALE, scale_ncols=D)
            Ls.o1, Ls.lse1 = _e((rows, D), BF16, dev), _e((B, H, Lq), F32, dev)
            ops.attn_fwd(heads(Ls.qkv[:, :D], B, Lq, H), heads(Ls.qkv[:, D:2 * D], B, Lq, H), heads(Ls.qkv[:, 2 * D:], B, Lq, H),
                         heads(Ls.o1, B, Lq, H), Ls.lse1, causal=True, q_log2=QK_LOG2)
            Ls.h2 = linear_fwd(Ls.o1, w.sa.o, rows, out_dtype=F32, residual=h)
            # cross-attention over the encoder output
            ln = lyr.encoder_attn_layer_norm
            Ls.x2, Ls.m2, Ls.r2 = _e((rows, D), BF16, dev), _e((rows,), F32, dev), _e((rows,), F32, dev)
            ops.fddt_ln_fwd(Ls.h2, rows, D, ln_w=ln.weight.detach(), ln_b=ln.bias.detach(), y_bf16=Ls.x2, mean=Ls.m2, rstd=Ls.r2)
            Ls.q = linear_fwd(Ls.x2, w.ca.q, rows, flags=L.EPI_SCALE_N, scale=Q_SCALE, scale_ncols=D)
            Ls.kv = linear_fwd(enc_bf, w.ca.kv, B * T)
            Ls.o2, Ls.lse2 = _e((rows, D), BF16, dev), _e((B, H, Lq), F32, dev)
            ops.attn_fwd(heads(Ls.q, B, Lq, H), heads(Ls.kv[:, :D], B, T, H), heads(Ls.kv[:, D:], B, T, H), heads(Ls.o2, B, Lq, H),
                         Ls.lse2, q_log2=QK_LOG2)
            Ls.h3 = linear_fwd(Ls.o2, w.ca.o, rows, out_dtype=F32, residual=Ls.h2)
            # feed-forward
            ln = lyr.final_layer_norm
            Ls.x3, Ls.m3, Ls.r3 = _e((rows, D), BF16, dev), _e((rows,), F32, dev), _e((rows,), F32, dev)
            ops.fddt_ln_fwd(Ls.h3, rows, D, ln_w=ln.weight.detach(), ln_b=ln.bias.detach(), y_bf16=Ls.x3, mean=Ls.m3, rstd=Ls.r3)
            Ls.u = _e((rows, F_), BF16, dev)
            Ls.a = linear_fwd(Ls.x3, w.fc1, rows, gelu_aux=Ls.u)
            h = linear_fwd(Ls.a, w.fc2, rows, out_dtype=F32, residual=Ls.h3)
            out.append(Ls)
        return h, out

    def forward(self, enc_bf, B, T, decoder_input_ids, labels, upp_labels, ts=None):
        model, cfg, W = self.model, self.cfg, self.W
        dec = model.model.decoder
        dev = enc_bf.device
        D, H, F_ = cfg.d_model, cfg.decoder_attention_heads, cfg.decoder_ffn_dim
        Lq = decoder_input_ids.shape[1]
        if Lq > cfg.max_target_positions:
            raise ValueError(f"decoder length {Lq} exceeds max_target_positions {cfg.max_target_positions}")
        rows = B * Lq
        S = NS(B=B, T=T, Lq=Lq, ids=decoder_input_ids.contiguous(), enc_bf=enc_bf)
        h = _e((rows, D), F32, dev)
        ops.embed_fwd(S.ids, dec.embed_tokens.weight.detach(), dec.embed_positions.weight.detach(), h)
        S.xf, S.mf, S.rf = _e((rows, D), BF16, dev), _e((rows,), F32, dev), _e((rows,), F32, dev)
        # A FROZEN decoder (the reference's default: frozen keyword "decoder") is row-parallel in both directions -- no parameter gradient
        # is reduced over rows -- so a large even batch runs its decoder layers as two half batches on two streams (SPLIT_FWD above):
        # the decoder's kernels at B x 128 rows fill a third of the chip each, two of them side by side fill two thirds.  The LM head
        # and the loss stay on the full batch behind the join; everything is bit-identical.
        split = (SPLIT_FWD and SPLIT_DEC and B % 2 == 0 and B * T >= SPLIT_FWD_MIN_ROWS and enc_bf.is_cuda
                 and _split_allowed() and not model.proj_out.weight.requires_grad
                 and not any(p.requires_grad for p in dec.parameters()))
        if split:
            main_st, parts, Bh = _fwd_parts(B, 1, dev)       # (row slices in units of sequences)
        else:
            main_st, parts, Bh = None, ((None, slice(0, B), slice(0, B)),), B
        S.parts = []
        for st_, _, bsl in parts:
            rq, re = slice(bsl.start * Lq, bsl.stop * Lq), slice(bsl.start * T, bsl.stop * T)
            with (torch.cuda.stream(st_) if st_ is not None else contextlib.nullcontext()):
                hl, layers = self._layers_fwd(enc_bf[re], Bh, T, Lq, h[rq])
                ops.fddt_ln_fwd(hl, Bh * Lq, D, ln_w=dec.layer_norm.weight.detach(), ln_b=dec.layer_norm.bias.detach(), y_bf16=S.xf[rq],
                                mean=S.mf[rq], rstd=S.rf[rq])
            S.parts.append(NS(st=st_, rq=rq, re=re, B=Bh, h_last=hl, layers=layers))
        for st_, _, _ in parts[1:]:
            main_st.wait_stream(st_)
        logits = _e((rows, W.vpad), BF16, dev)
        ops.gemm_nt(S.xf, W.head.w, logits, rows, W.vpad, D)
        S.logits = logits
        loss = None
        if labels is not None:
            S.labels = labels.contiguous().view(-1)
            S.upp = None if upp_labels is None else upp_labels.contiguous().view(-1)
            S.lse, S.row_loss = _e((rows,), F32, dev), _e((rows,), F32, dev)
            S.choice = _e((rows,), torch.int32, dev)
            S.acc = torch.zeros(2, dtype=F32, device=dev)               # [loss_sum, count]
            S.soft = ts is not None
            S.ce = ops.ce_args(logits, W.vpad, rows, cfg.vocab_size, S.labels, S.upp, S.soft, ts, S.lse, S.row_loss, S.choice,
                               S.acc[0:1], S.acc[1:2])
            ops.ce_loss_fwd(S.ce)
            if S.soft:                                                   # modeling_dicow.py:144
                S.denom = S.acc[1].clamp(min=1.0)
            else:                                                        # modeling_dicow.py:318-323 (.mean() over all positions)
                S.denom = torch.full((), float(rows), dtype=F32, device=dev)
            loss = S.acc[0] / S.denom
        return loss, logits.view(B, Lq, W.vpad)[:, :, :cfg.vocab_size], S

    def _layers_bwd(self, layers, enc_bf, B, T, Lq, g, gb, G, d_enc):
        """Backward of _layers_fwd on B sequences: g / gb = gradient wrt the last hidden state (fp32 / bf16); accumulates the
        cross-attention's share into d_enc [B*T, D] (when given) and returns the gradient wrt the embeddings."""
        model, cfg, W = self.model, self.cfg, self.W
        dec = model.model.decoder
        dev = g.device
        D, H, F_ = cfg.d_model, cfg.decoder_attention_heads, cfg.decoder_ffn_dim
        rows = B * Lq
        nl = len(dec.layers)
        for i in range(nl - 1, -1, -1):
            lyr, w, Ls = dec.layers[i], W.layers[i], layers[i]
            # FFN
            linear_wgrad(gb, Ls.a, G.get(lyr.fc2.weight), rows)
            d_u = linear_dgrad(gb, w.fc2, rows, aux=Ls.u)
            bias_grad(d_u, G.get(lyr.fc1.bias))
            linear_wgrad(d_u, Ls.x3, G.get(lyr.fc1.weight), rows)
            d_x3 = linear_dgrad(d_u, w.fc1, rows)
            g3, g3b = _e((rows, D), F32, dev), _e((rows, D), BF16, dev)
            ln = lyr.final_layer_norm
            ops.fddt_ln_bwd(Ls.h3, rows, D, ln_w=ln.weight.detach(), mean=Ls.m3, rstd=Ls.r3, d_y=d_x3, g_res=g, g_out=g3,
                            g_out_bf16=g3b, dln_w=G.get(ln.weight), dln_b=G.get(ln.bias),
                            colsum_out=G.get(lyr.encoder_attn.out_proj.bias))
            # cross-attention
            att = lyr.encoder_attn
            linear_wgrad(g3b, Ls.o2, G.get(att.out_proj.weight), rows)
            d_o2 = linear_dgrad(g3b, w.ca.o, rows)
            dq = _e((rows, D), BF16, dev)
            dkv = _e((B * T, 2 * D), BF16, dev)
            delta = _e((2, B, H, Lq), F32, dev)
            ops.attn_bwd(heads(Ls.q, B, Lq, H), heads(Ls.kv[:, :D], B, T, H), heads(Ls.kv[:, D:], B, T, H), heads(Ls.o2, B, Lq, H),
                         heads(d_o2, B, Lq, H), Ls.lse2, delta, heads(dq, B, Lq, H), heads(dkv[:, :D], B, T, H),
                         heads(dkv[:, D:], B, T, H), dq_scale=0.125, q_log2=QK_LOG2)
            bias_grad(dq, G.get(att.q_proj.bias))
            bias_grad(dkv[:, D:], G.get(att.v_proj.bias))
            linear_wgrad(dq, Ls.x2, G.get(att.q_proj.weight), rows)
            linear_wgrad(dkv[:, :D], enc_bf, G.get(att.k_proj.weight), B * T)
            linear_wgrad(dkv[:, D:], enc_bf, G.get(att.v_proj.weight), B * T)
            if d_enc is not None:
                linear_dgrad(dkv, w.ca.kv, B * T, out=d_enc, accumulate=True)
            d_x2 = linear_dgrad(dq, w.ca.q, rows)
            g2, g2b = _e((rows, D), F32, dev), _e((rows, D), BF16, dev)
            ln = lyr.encoder_attn_layer_norm
            ops.fddt_ln_bwd(Ls.h2, rows, D, ln_w=ln.weight.detach(), mean=Ls.m2, rstd=Ls.r2, d_y=d_x2, g_res=g3, g_out=g2,
                            g_out_bf16=g2b, dln_w=G.get(ln.weight), dln_b=G.get(ln.bias),
                            colsum_out=G.get(lyr.self_attn.out_proj.bias))
            # causal self-attention
            att = lyr.self_attn
            linear_wgrad(g2b, Ls.o1, G.get(att.out_proj.weight), rows)
            d_o1 = linear_dgrad(g2b, w.sa.o, rows)
            d_qkv = _e((rows, 3 * D), BF16, dev)
            qkv = Ls.qkv
            ops.attn_bwd(heads(qkv[:, :D], B, Lq, H), heads(qkv[:, D:2 * D], B, Lq, H), heads(qkv[:, 2 * D:], B, Lq, H),
                         heads(Ls.o1, B, Lq, H), heads(d_o1, B, Lq, H), Ls.lse1, delta, heads(d_qkv[:, :D], B, Lq, H),
                         heads(d_qkv[:, D:2 * D], B, Lq, H), heads(d_qkv[:, 2 * D:], B, Lq, H), causal=True, dq_scale=0.125, q_log2=QK_LOG2)
            bias_grad(d_qkv[:, :D], G.get(att.q_proj.bias))
            bias_grad(d_qkv[:, 2 * D:], G.get(att.v_proj.bias))
            linear_wgrad(d_qkv[:, :D], Ls.x1, G.get(att.q_proj.weight), rows)
            linear_wgrad(d_qkv[:, D:2 * D], Ls.x1, G.get(att.k_proj.weight), rows)
            linear_wgrad(d_qkv[:, 2 * D:], Ls.x1, G.get(att.v_proj.weight), rows)
            d_x1 = linear_dgrad(d_qkv, w.sa.qkv, rows)
            g1, g1b = _e((rows, D), F32, dev), _e((rows, D), BF16, dev)
            ln = lyr.self_attn_layer_norm
            ops.fddt_ln_bwd(Ls.h_in, rows, D, ln_w=ln.weight.detach(), mean=Ls.m1, rstd=Ls.r1, d_y=d_x1, g_res=g2, g_out=g1,
                            g_out_bf16=g1b, dln_w=G.get(ln.weight), dln_b=G.get(ln.bias),
                            colsum_out=G.get(dec.layers[i - 1].fc2.bias) if i > 0 else None)
            g, gb = g1, g1b
        return g

    def backward(self, S, grad_loss, G, need_d_enc=True, d_logits_ext=None):
        """Returns d_enc fp32 [B*T, D] (or None)."""
        model, cfg, W = self.model, self.cfg, self.W
        dec = model.model.decoder
        dev = S.logits.device
        D, H, F_ = cfg.d_model, cfg.decoder_attention_heads, cfg.decoder_ffn_dim
        B, T, Lq = S.B, S.T, S.Lq
        rows = B * Lq
        d_logits = _e((rows, W.vpad), BF16, dev)
        scale = (grad_loss.to(F32) / S.denom).reshape(1)
        S.ce.d_logits = d_logits.data_ptr()
        ops.ce_loss_bwd(S.ce, scale)
        # tied LM head (modeling_dicow.py:302): d_x = d_logits @ E ; dE += d_logits^T @ x
        ge = G.get(model.proj_out.weight)
        if ge is not None:
            if G.is_direct(model.proj_out.weight) and W.vpad != cfg.vocab_size:
                # flat-store gradient (trainer.FlatStore) of a vocabulary that is not a multiple of 128: the padded rows go
                # through a temporary so that the tied weight's gradient is COMPLETE in the flat store before the "decoder"
                # segment is handed to the data-parallel all-reduce (it used to reach p.grad only after autograd returned)
                tmp = torch.zeros(W.vpad, D, dtype=F32, device=dev)
                ops.gemm_tn(d_logits, S.xf, tmp, rows, W.vpad, D)
                ge.add_(tmp[:cfg.vocab_size])
            else:
                ops.gemm_tn(d_logits, S.xf, G.raw(model.proj_out.weight, W.vpad * D).view(W.vpad, D), rows, W.vpad, D)
        d_xf = linear_dgrad(d_logits, W.head, rows)
        g = _e((rows, D), F32, dev)
        nl = len(dec.layers)
        d_enc = torch.zeros(B * T, D, dtype=F32, device=dev) if need_d_enc else None
        split = len(S.parts) > 1                         # (only a frozen decoder was split: no gradient below is reduced over rows)
        if split:
            main_st = torch.cuda.current_stream(dev)
            for P in S.parts[1:]:
                P.st.wait_stream(main_st)                # fork: d_xf and the zeroed d_enc are complete
        for P in S.parts:
            rq, re, rows_p = P.rq, P.re, P.B * Lq
            with (torch.cuda.stream(P.st) if split else contextlib.nullcontext()):
                gb = _e((rows_p, D), BF16, dev)
                ops.fddt_ln_bwd(P.h_last, rows_p, D, ln_w=dec.layer_norm.weight.detach(), mean=S.mf[rq], rstd=S.rf[rq], d_y=d_xf[rq], g_out=g[rq],
                                g_out_bf16=gb, dln_w=G.get(dec.layer_norm.weight), dln_b=G.get(dec.layer_norm.bias),
                                colsum_out=G.get(dec.layers[nl - 1].fc2.bias))
                g_emb = self._layers_bwd(P.layers, S.enc_bf[re], P.B, T, Lq, g[rq], gb, G, d_enc[re] if need_d_enc else None)
        if split:
            for P in S.parts[1:]:
                main_st.wait_stream(P.st)
        gt, gp = G.get(dec.embed_tokens.weight), G.get(dec.embed_positions.weight)
        if gt is not None or gp is not None:             # (a trainable decoder is never split: one part, g_emb covers every row)
            assert not split
            ops.embed_bwd(S.ids, g_emb, gt, gp, D)
        return d_enc


# ------------------------------------------------------------------------------------------------ CTC auxiliary branch
class CtcEngine:
    """get_enc_logits + CTC loss of the reference (modeling_dicow.py:242-246,326-336; encoder.py:87-135):
    optional extra self-attention (its output REPLACES the hidden states), two activation-free stride-2 convolutions
    (as GEMMs over overlapping time-major views), lm_head over vocab+1 classes, CTC loss with blank = vocab."""

    def __init__(self, enc):
        self.enc = enc
        self.cfg = enc.config

    def prepare(self):
        enc, cfg = self.enc, self.cfg
        dev = enc.lm_head.weight.device
        D = cfg.d_model
        W = NS()
        W.att = prep_attention(enc.additional_self_attention_layer, dev) if hasattr(enc, "additional_self_attention_layer") else None
        W.lyr = None
        if hasattr(enc, "additional_layer"):                 # takes precedence over the bare attention (encoder.py:88-95)
            lyr = enc.additional_layer
            W.lyr = NS(att=prep_attention(lyr.self_attn, dev), fc1=prep_linear([lyr.fc1.weight], [lyr.fc1.bias], dev),
                       fc2=prep_linear([lyr.fc2.weight], [lyr.fc2.bias], dev))
            W.att = None
        if hasattr(enc, "subsample_conv1"):
            W.c1, W.c1_t = ops.conv_weight_pack(enc.subsample_conv1.weight.detach(), 3 * D, want_t=True)
            W.c2, W.c2_t = ops.conv_weight_pack(enc.subsample_conv2.weight.detach(), 3 * D, want_t=True)
        W.cpad = _ceil(cfg.vocab_size + 1, 128)
        W.head = prep_linear([enc.lm_head.weight], None, dev, n_pad=W.cpad)
        self.W = W
        return W

    def forward(self, enc_bf, B, T, labels, enc_f32=None):
        cfg, W = self.cfg, self.W
        dev = enc_bf.device
        S = self.encode_logits(enc_bf, B, T, enc_f32)
        logits, Tn = S.logits, S.Tn
        Cc = cfg.vocab_size + 1
        lab = labels.contiguous()
        Lc = lab.shape[1]
        S.lse, S.nll, S.tlen = _e((B * Tn,), F32, dev), _e((B,), F32, dev), _e((B,), F32, dev)
        Smax = 2 * Lc + 1
        S.ab = _e((2, B, Tn, Smax), F32, dev)
        S.acc = torch.zeros(1, dtype=F32, device=dev)
        S.labels = lab
        S.ctc = ops.ctc_args(logits, W.cpad, B, Tn, Cc, lab, S.lse, S.ab[0], S.ab[1], S.nll, S.tlen, S.acc)
        ops.ctc_loss_fwd(S.ctc)
        if cfg.ctc_loss_reduction != "mean":
            raise L.DicowError("only ctc_loss_reduction='mean' (reference default) is implemented")
        return S.acc[0] / B, S

    def encode_logits(self, enc_bf, B, T, enc_f32=None):
        """The CTC branch up to its logits (encoder.py:87-106, get_enc_logits): S.logits bf16 [B * Tn, cpad].
        enc_f32: the fp32 rows enc_bf was rounded from (the residual stream of ``additional_layer``)."""
        enc, cfg, W = self.enc, self.cfg, self.W
        dev = enc_bf.device
        D, H = cfg.d_model, cfg.encoder_attention_heads
        rows = B * T
        S = NS(B=B, T=T, enc_bf=enc_bf)
        sub = hasattr(enc, "subsample_conv1")
        if sub and T % 4 != 0:
            raise L.DicowError("pre_ctc_sub_sample needs max_source_positions % 4 == 0")
        h = enc_bf
        hpad = None
        if W.lyr is not None:
            hf, S.lyr = plain_layer_fwd(enc.additional_layer, W.lyr, enc_f32 if enc_f32 is not None else enc_bf.float(), B, T, H,
                                        cfg.encoder_ffn_dim)
            h = ops.cast_bf16(hf)
            if sub:
                hpad = torch.zeros(B, T + 2, D, dtype=BF16, device=dev)
                hpad[:, 1:T + 1].copy_(h.view(B, T, D))
        elif W.att is not None:
            S.qkv = linear_fwd(enc_bf, W.att.qkv, rows, flags=L.EPI_SCALE_N, scale=Q_SCALE, scale_ncols=D)
            S.o, S.lse_a = _e((rows, D), BF16, dev), _e((B, H, T), F32, dev)
            ops.attn_fwd(heads(S.qkv[:, :D], B, T, H), heads(S.qkv[:, D:2 * D], B, T, H), heads(S.qkv[:, 2 * D:], B, T, H),
                         heads(S.o, B, T, H), S.lse_a, q_log2=QK_LOG2)
            if sub:                                   # write the projection straight into the zero-padded conv input
                hpad = _e((B, T + 2, D), BF16, dev)
                hpad[:, 0].zero_()
                hpad[:, T + 1].zero_()
                ops.gemm_nt(S.o, W.att.o.w, hpad[:, 1:], T, D, D, bias=W.att.o.b, batch=B, strideA=T * D, strideC=(T + 2) * D)
            else:
                h = linear_fwd(S.o, W.att.o, rows)
        elif sub:
            hpad = torch.zeros(B, T + 2, D, dtype=BF16, device=dev)
            hpad[:, 1:T + 1].copy_(enc_bf.view(B, T, D))
        Tn = T
        if sub:
            T1, T2 = T // 2, T // 4
            c1pad = _e((B, T1 + 2, D), BF16, dev)
            c1pad[:, 0].zero_()
            c1pad[:, T1 + 1].zero_()
            ops.gemm_nt(hpad, W.c1, c1pad[:, 1:], T1, D, 3 * D, lda=2 * D, batch=B, strideA=(T + 2) * D, strideC=(T1 + 2) * D)
            h = _e((B * T2, D), BF16, dev)
            ops.gemm_nt(c1pad, W.c2, h, T2, D, 3 * D, lda=2 * D, batch=B, strideA=(T1 + 2) * D, strideC=T2 * D)
            S.hpad, S.c1pad, S.T1 = hpad, c1pad, T1
            Tn = T2
        S.h, S.Tn = h, Tn
        logits = _e((B * Tn, W.cpad), BF16, dev)
        ops.gemm_nt(h, W.head.w, logits, B * Tn, W.cpad, D)
        S.logits = logits
        return S

    def backward(self, S, grad_loss, G):
        """Returns d_enc fp32 [B*T, D]."""
        enc, cfg, W = self.enc, self.cfg, self.W
        dev = S.logits.device
        D, H = cfg.d_model, cfg.encoder_attention_heads
        B, T, Tn = S.B, S.T, S.Tn
        rows = B * T
        d_logits = _e((B * Tn, W.cpad), BF16, dev)
        S.ctc.d_logits = d_logits.data_ptr()
        ops.ctc_loss_bwd(S.ctc, grad_loss.to(F32).reshape(1))
        return self.backward_from_logits(S, d_logits, G)

    def backward_from_logits(self, S, d_logits, G):
        """d_logits bf16 [B*Tn, cpad] (padding columns zero) -> d_enc fp32 [B*T, D]; the head, the subsampling convs and the
        extra attention layer accumulate their gradients into G."""
        enc, cfg, W = self.enc, self.cfg, self.W
        dev = S.logits.device
        D, H = cfg.d_model, cfg.encoder_attention_heads
        B, T, Tn = S.B, S.T, S.Tn
        rows = B * T
        glm = G.get(enc.lm_head.weight)
        if glm is not None:
            tmp = torch.zeros(W.cpad, D, dtype=F32, device=dev)          # vocab+1 is not a multiple of 8: padded rows
            ops.gemm_tn(d_logits, S.h, tmp, B * Tn, W.cpad, D)
            glm.add_(tmp[:cfg.vocab_size + 1])
        d_h = linear_dgrad(d_logits, W.head, B * Tn)                     # bf16 [B*Tn, D]
        sub = hasattr(enc, "subsample_conv1")
        if sub:
            T1 = S.T1
            # subsample_conv2 backward
            g2 = G.get(enc.subsample_conv2.weight)
            if g2 is not None:
                tmp = torch.zeros(D, 3 * D, dtype=F32, device=dev)
                ops.gemm_tn(d_h, S.c1pad, tmp, Tn, D, 3 * D, lda=D, ldb=2 * D, batch=B, strideA=Tn * D, strideB=(T1 + 2) * D)
                ops.conv_weight_unpack_grad(tmp, g2)
            dA = _e((B * Tn, 3 * D), BF16, dev)
            ops.gemm_nt(d_h, W.c2_t, dA, B * Tn, 3 * D, D)
            d_c1 = _e((B, T1, D), BF16, dev)
            ops.conv2_col2im_gelu_bwd(dA, None, d_c1, B, Tn, D)
            # subsample_conv1 backward
            g1 = G.get(enc.subsample_conv1.weight)
            if g1 is not None:
                tmp = torch.zeros(D, 3 * D, dtype=F32, device=dev)
                ops.gemm_tn(d_c1, S.hpad, tmp, T1, D, 3 * D, lda=D, ldb=2 * D, batch=B, strideA=T1 * D, strideB=(T + 2) * D)
                ops.conv_weight_unpack_grad(tmp, g1)
            dA1 = _e((B * T1, 3 * D), BF16, dev)
            ops.gemm_nt(d_c1.view(B * T1, D), W.c1_t, dA1, B * T1, 3 * D, D)
            d_hid = _e((B, T, D), BF16, dev)
            ops.conv2_col2im_gelu_bwd(dA1, None, d_hid, B, T1, D)
            d_h = d_hid.view(rows, D)
        if W.lyr is not None:
            return plain_layer_bwd(enc.additional_layer, W.lyr, S.lyr, d_h.contiguous(), G, B, T, H)
        if W.att is None:
            return d_h.float()
        att = enc.additional_self_attention_layer
        bias_grad(d_h, G.get(att.out_proj.bias))
        linear_wgrad(d_h, S.o, G.get(att.out_proj.weight), rows)
        d_o = linear_dgrad(d_h, W.att.o, rows)
        d_qkv = _e((rows, 3 * D), BF16, dev)
        delta = _e((2, B, H, T), F32, dev)
        qkv = S.qkv
        ops.attn_bwd(heads(qkv[:, :D], B, T, H), heads(qkv[:, D:2 * D], B, T, H), heads(qkv[:, 2 * D:], B, T, H),
                     heads(S.o, B, T, H), heads(d_o, B, T, H), S.lse_a, delta, heads(d_qkv[:, :D], B, T, H),
                     heads(d_qkv[:, D:2 * D], B, T, H), heads(d_qkv[:, 2 * D:], B, T, H), dq_scale=0.125, q_log2=QK_LOG2)
        bias_grad(d_qkv[:, :D], G.get(att.q_proj.bias))
        bias_grad(d_qkv[:, 2 * D:], G.get(att.v_proj.bias))
        qkv_wgrad(d_qkv, S.enc_bf, G.get(att.q_proj.weight), G.get(att.k_proj.weight), G.get(att.v_proj.weight), rows, D)
        return linear_dgrad(d_qkv, W.att.qkv, rows, out_dtype=F32)
