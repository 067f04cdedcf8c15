"""Drop-in module surface of the reference's DiCoW model, backed by the HIP engine.

Mirrors (reference, /root/reference/src/models/dicow): ``FDDT`` (FDDT.py:6-63), ``DiCoWEncoder`` (encoder.py:10-246),
``DiCoW`` / ``DiCoWForConditionalGeneration`` (modeling_dicow.py:146-357) -- same constructor arguments, forward
keyword arguments, returned fields (``.loss``, ``.logits``, ``.encoder_last_hidden_state``) and state-dict keys
(SURVEY.md section 8b), so ``model(**batch)`` / ``loss.backward()`` callers (HF Trainer, the reference's
``src/train.py``) work unchanged.  The ``nn.Module`` tree only HOLDS parameters (fp32 masters); all arithmetic runs
in ``engine.py`` -> ``libdicow_hip.so``.  Modules refuse CPU tensors: there is no fallback path.
"""
import math
import re
from collections import OrderedDict
from types import SimpleNamespace as NS_
from typing import Optional

import torch
from torch import nn

from . import _lib as L
from . import ops
from .config import DiCoWConfig, _HF

if _HF:                                     # the reference's class is a transformers.PreTrainedModel: the HF Trainer the reference trains with
    from transformers import PreTrainedModel as _ModelBase, GenerationConfig as _GenerationConfig      # (src/train.py:227-262) asks for
else:                                       # pragma: no cover                                                       # exactly that type
    _ModelBase, _GenerationConfig = nn.Module, None
from .engine import EncoderEngine, DecoderEngine, CtcEngine, GradSink, fddt_ptrs, CLS, prep_full_fddt, full_fddt_fwd, full_fddt_bwd

F32, BF16 = torch.float32, torch.bfloat16


class ModelOutput(OrderedDict):
    """Attribute + index access like HF's ModelOutput (None fields are skipped when indexing by position)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __getitem__(self, k):
        if isinstance(k, int):
            return [v for v in self.values() if v is not None][k]
        return super().__getitem__(k)

    def to_tuple(self):
        return tuple(v for v in self.values() if v is not None)


# ------------------------------------------------------------------------------------------------ parameter holders
class DiagonalLinear(nn.Module):
    """out = x * weight (+ bias): parameters of one diagonal FDDT class (reference layers.py:49-77)."""

    def __init__(self, d_model, bias=True, init_eye_val=0.0, fddt_init=None):
        super().__init__()
        self.init_eye_val, self.fddt_init = init_eye_val, fddt_init
        self.weight = nn.Parameter(torch.empty(d_model))
        self.bias = nn.Parameter(torch.zeros(d_model)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        with torch.no_grad():
            bound = math.sqrt(3.0 / self.weight.numel())
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.zero_()
            if self.fddt_init == "non-disturbing":
                self.weight.fill_(1.0)
            elif self.fddt_init == "suppressive":
                self.weight.fill_(self.init_eye_val)


class DenseLinear(nn.Linear):
    """Full D x D FDDT class / SCB feed-forward Linear with the reference's initialisation modes (layers.py:7-47)."""

    def __init__(self, in_f, out_f, bias=True, init_eye_val=0.0, fddt_init=None, init_fun=None):
        self.init_eye_val, self.fddt_init, self.init_fun = init_eye_val, fddt_init, init_fun
        super().__init__(in_f, out_f, bias=bias)

    def reset_parameters(self):
        with torch.no_grad():
            if getattr(self, "init_fun", None) is not None:
                self.init_fun(self)
                return
            nn.init.xavier_uniform_(self.weight)
            if self.bias is not None:
                nn.init.zeros_(self.bias)
            mode = getattr(self, "fddt_init", None)
            if mode in ("non-disturbing", "suppressive"):
                val = 1.0 if mode == "non-disturbing" else self.init_eye_val
                eye = torch.zeros_like(self.weight)
                n = min(self.weight.shape)
                eye[:n, :n] = val * torch.eye(n)
                self.weight.copy_(eye)


def _init_first_half_identity(m):          # layers.py:95-106
    nn.init.xavier_uniform_(m.weight, gain=1e-1)
    h = m.weight.shape[1] // 2
    m.weight.data[:h, :h] += torch.eye(h)
    m.bias.data.zero_()


def _init_first_embeds_identity(m):        # layers.py:109-117
    nn.init.xavier_uniform_(m.weight, gain=1e-1)
    m.weight.data[:, :m.weight.shape[0]] += torch.eye(m.weight.shape[0])
    m.bias.data.zero_()


class Gate(nn.Module):
    def __init__(self, items, init_val=0.0):
        super().__init__()
        self.init_val = init_val
        self.gate = nn.Parameter(torch.full((items,), init_val))

    def reset_parameters(self):
        with torch.no_grad():
            self.gate.fill_(self.init_val)


class Attention(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.k_proj = nn.Linear(d, d, bias=False)
        self.v_proj = nn.Linear(d, d)
        self.q_proj = nn.Linear(d, d)
        self.out_proj = nn.Linear(d, d)


class EncoderLayer(nn.Module):
    def __init__(self, d, f):
        super().__init__()
        self.self_attn = Attention(d)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.fc1 = nn.Linear(d, f)
        self.fc2 = nn.Linear(f, d)
        self.final_layer_norm = nn.LayerNorm(d)


class DecoderLayer(nn.Module):
    def __init__(self, d, f):
        super().__init__()
        self.self_attn = Attention(d)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.encoder_attn = Attention(d)
        self.encoder_attn_layer_norm = nn.LayerNorm(d)
        self.fc1 = nn.Linear(d, f)
        self.fc2 = nn.Linear(f, d)
        self.final_layer_norm = nn.LayerNorm(d)


class CrossAttentionEnrollBlock(nn.Module):
    def __init__(self, config):
        super().__init__()
        d, f = config.d_model, config.encoder_ffn_dim
        self.cross_attn = Attention(d)
        self.cross_gate = Gate(1, init_val=0.0)
        self.ffn = nn.Sequential(DenseLinear(2 * d, f, init_fun=_init_first_half_identity), nn.GELU(), nn.Dropout(0.0),
                                 DenseLinear(f, d, init_fun=_init_first_embeds_identity), nn.Dropout(0.0))


class SpeakerCommunicationBlock(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.streams = 2
        self.cae = CrossAttentionEnrollBlock(config)


def _require_cuda(t, what):
    if not t.is_cuda:
        raise L.DicowError(f"{what}: tensors must be on the GPU -- the MI355X path has no CPU fallback "
                           "(the CPU oracle under oracle/ is test infrastructure only)")


def _param_sig(params):
    return tuple((p.data_ptr(), p._version) for p in params)


# ------------------------------------------------------------------------------------------------ FDDT module
class _FDDTFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, hidden, stno, *params):
        _require_cuda(hidden, "FDDT")
        B, T, D = hidden.shape
        h = hidden.contiguous()
        if h.dtype not in (F32, BF16):
            h = h.to(F32)
        st = stno.to(device=h.device, dtype=F32).contiguous()
        mode, w, b = fddt_ptrs(mod, None)
        out = torch.empty(B, T, D, dtype=F32, device=h.device)
        ops.fddt_ln_fwd(h, B * T, D, mode=mode, stno=st, T=T, w=tuple(None if x is None else x.detach() for x in w),
                        b=tuple(None if x is None else x.detach() for x in b), h_out=out)
        ctx.mod, ctx.h, ctx.st, ctx.params = mod, h, st, params
        return out

    @staticmethod
    def backward(ctx, g):
        mod, h, st = ctx.mod, ctx.h, ctx.st
        B, T, D = h.shape
        G = GradSink(ctx.params, h.device)
        mode, w, b = fddt_ptrs(mod, None)
        gh = torch.empty(B, T, D, dtype=F32, device=h.device)
        ops.fddt_ln_bwd(h, B * T, D, mode=mode, stno=st, T=T, w=tuple(None if x is None else x.detach() for x in w),
                        b=tuple(None if x is None else x.detach() for x in b), g_res=g.contiguous().to(F32), g_out=gh,
                        dw=tuple(G.get(x) for x in w), db=tuple(G.get(x) for x in b))
        return (None, gh.to(ctx.h.dtype) if ctx.h.dtype != F32 else gh, None) + tuple(G.result(p) for p in ctx.params)


class FDDT(nn.Module):
    """Frame-level diarization-dependent transformation; signature of reference FDDT.py:7-8."""

    def __init__(self, d_model, non_target_rate=0.01, fddt_init=None, is_diagonal=False, bias_only=False, use_silence=True,
                 use_target=True, use_overlap=True, use_non_target=True):
        super().__init__()

        def make(val):
            if bias_only:
                return nn.Parameter(torch.zeros(d_model))
            if is_diagonal:
                return DiagonalLinear(d_model, bias=True, fddt_init=fddt_init, init_eye_val=val)
            return DenseLinear(d_model, d_model, bias=True, fddt_init=fddt_init, init_eye_val=val)

        if use_target:
            self.target_linear = make(1.0)
        if use_non_target:
            self.non_target_linear = make(non_target_rate)
        if use_overlap:
            self.overlap_linear = make(1.0)
        if use_silence:
            self.silence_linear = make(non_target_rate)
        self.use_silence, self.use_target, self.use_overlap, self.use_non_target = use_silence, use_target, use_overlap, use_non_target
        self.bias_only, self.is_diagonal, self.d_model = bias_only, is_diagonal or bias_only, d_model

    def forward(self, hidden_states, stno_mask):
        if not self.is_diagonal:
            return self._forward_full(hidden_states, stno_mask)
        params = [p for p in self.parameters()]
        return _FDDTFn.apply(self, hidden_states, stno_mask, *params)

    # full D x D variant (FDDT.py:13-16): one GEMM [rows,D] x [D,4D] + masked combine
    def _forward_full(self, hidden_states, stno_mask):
        params = [p for p in self.parameters()]
        return _FDDTFullFn.apply(self, hidden_states, stno_mask, *params)


class _FDDTFullFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, hidden, stno, *params):
        _require_cuda(hidden, "FDDT")
        B, T, D = hidden.shape
        if D % 64 != 0:
            raise L.DicowError("full FDDT needs d_model % 64 == 0")
        dev = hidden.device
        h = hidden.contiguous().to(F32).view(B * T, D)
        st = stno.to(device=dev, dtype=F32).contiguous()
        w = prep_full_fddt(mod, dev)
        out, hb = full_fddt_fwd(w, h, st, 4 * T, B * T, T, D)
        ctx.mod, ctx.w, ctx.hb, ctx.st, ctx.params, ctx.shape = mod, w, hb, st, params, (B, T, D)
        return out.view(B, T, D)

    @staticmethod
    def backward(ctx, g):
        B, T, D = ctx.shape
        G = GradSink(ctx.params, g.device)
        gh = full_fddt_bwd(ctx.mod, ctx.w, ctx.hb, g.contiguous().to(F32).view(B * T, D), ctx.st, 4 * T, G, B * T, T, D)
        return (None, gh.view(B, T, D), None) + tuple(G.result(p) for p in ctx.params)


def _refuse_unsupported(where, **kw):
    """Arguments of the HF / reference signatures that this path does not implement must not be dropped silently."""
    bad = [k for k, v in kw.items() if v is not None]
    if bad:
        raise NotImplementedError(f"{where}: {', '.join(bad)} not supported by the HIP path (per-head masks, attention / "
                                  "hidden-state outputs, decoder masks / positions have no kernels here)")


# ------------------------------------------------------------------------------------------------ encoder
class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, input_features, stno, enrollments, need_grad, *params):
        eng = enc._engine()
        out, S = eng.forward(input_features, stno, enrollments, need_grad=need_grad)
        ctx.enc, ctx.S, ctx.params = enc, (S if need_grad else None), params
        return out

    @staticmethod
    def backward(ctx, d_enc):
        enc, S = ctx.enc, ctx.S
        G = GradSink(ctx.params, d_enc.device)
        D = enc.config.d_model
        enc._engine(prepare=False).backward(S, d_enc.contiguous().view(-1, D).to(F32), G)
        return (None, None, None, None, None) + tuple(G.result(p) for p in ctx.params)


class DiCoWEncoder(nn.Module):
    def __init__(self, config: DiCoWConfig):
        super().__init__()
        self.config = config
        d, f = config.d_model, config.encoder_ffn_dim
        self.conv1 = nn.Conv1d(config.num_mel_bins, d, kernel_size=3, padding=1)
        self.conv2 = nn.Conv1d(d, d, kernel_size=3, stride=2, padding=1)
        self.embed_positions = nn.Embedding(config.max_source_positions, d)
        self.embed_positions.requires_grad_(False)
        self.layers = nn.ModuleList([EncoderLayer(d, f) for _ in range(config.encoder_layers)])
        self.layer_norm = nn.LayerNorm(d)
        self.ctc_weight = config.ctc_weight
        if config.ctc_weight > 0.0:                      # CTC auxiliary branch (reference encoder.py:15-44)
            if config.additional_layer:                  # a full extra encoder layer; wins over the bare attention below
                self.additional_layer = EncoderLayer(d, f)
            if config.additional_self_attention_layer:
                self.additional_self_attention_layer = Attention(d)
            if config.pre_ctc_sub_sample:
                self.subsample_conv1 = nn.Conv1d(d, d, kernel_size=3, stride=2, padding=1, bias=False)
                self.subsample_conv2 = nn.Conv1d(d, d, kernel_size=3, stride=2, padding=1, bias=False)
            self.lm_head = nn.Linear(d, config.vocab_size + 1, bias=False)
        self.first_task_token = config.vocab_size - 30 * 50 - 1 - 6
        if config.use_fddt:
            def mk(rate):
                return FDDT(d_model=d, non_target_rate=rate, fddt_init=config.fddt_init, is_diagonal=config.fddt_is_diagonal,
                            bias_only=config.fddt_bias_only, use_silence=config.fddt_use_silence,
                            use_target=config.fddt_use_target, use_overlap=config.fddt_use_overlap,
                            use_non_target=config.fddt_use_non_target)
            self.fddts = nn.ModuleList([mk(1.0) for _ in range(config.num_fddts)])
            if config.use_pre_pos_fddt:
                self.initial_fddt = mk(config.non_target_fddt_value)
        if config.use_enrollments and config.scb_layers is not None:
            self.ca_enrolls = nn.ModuleList([SpeakerCommunicationBlock(config) for _ in range(config.scb_layers)])
        self._eng = None
        self._sig = None
        self._ctc_eng = None
        self._ctc_sig = None

    _CTC_PREFIXES = ("additional_layer.", "additional_self_attention_layer.", "subsample_conv", "lm_head.")

    def get_loss(self, logits, labels):
        """CTC loss of ``forward(..., return_logits=True).logits`` (reference encoder.py:108-135)."""
        if int(labels.max()) >= self.config.vocab_size:
            raise ValueError(f"Label values must be <= vocab_size: {self.config.vocab_size}")
        lab = labels.to(logits.device)
        if self.config.remove_timestamps_from_ctc:          # encoder.py:111-113: keep labels below the first task token
            keep = lab < self.first_task_token
            order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)
            lab = torch.gather(lab, 1, order)
            cnt = keep.sum(dim=1, keepdim=True)
            lab = torch.where(torch.arange(lab.shape[1], device=lab.device)[None, :] < cnt, lab, torch.full_like(lab, -100))
            lab = lab[:, :max(int(cnt.max()), 1)]
        valid = lab >= 0                                    # valid targets first (CTC target_lengths = count of >= 0)
        lab = torch.gather(lab, 1, torch.argsort((~valid).to(torch.int8), dim=1, stable=True)).contiguous()
        return _CtcLossFn.apply(self, logits, lab)

    def ctc_parameters(self):
        return [p for n, p in self.named_parameters() if n.startswith(self._CTC_PREFIXES)]

    def _ctc_engine(self, prepare=True):
        if self._ctc_eng is None:
            self._ctc_eng = CtcEngine(self)
        if prepare:
            sig = _param_sig(self.ctc_parameters())
            if sig != self._ctc_sig:
                self._ctc_eng.prepare()
                self._ctc_sig = sig
        return self._ctc_eng

    def _engine(self, prepare=True):
        if self._eng is None:
            self._eng = EncoderEngine(self)
        if prepare:
            ctc_ids = {id(p) for p in self.ctc_parameters()} if self.ctc_weight > 0.0 else set()
            sig = _param_sig([p for p in self.parameters() if id(p) not in ctc_ids])
            if sig != self._sig:
                self._eng.prepare()
                self._sig = sig
        return self._eng

    def get_max_len(self):
        return self.config.max_source_positions * 2

    def forward(self, input_features, attention_mask=None, head_mask=None, output_attentions=None, output_hidden_states=None,
                return_dict=None, stno_mask=None, return_logits=False, enrollments=None):
        _require_cuda(input_features, "DiCoWEncoder")
        if stno_mask is None:
            raise ValueError("stno_mask is required")
        _refuse_unsupported("DiCoWEncoder.forward", head_mask=head_mask, output_attentions=output_attentions or None,
                            output_hidden_states=output_hidden_states or None)
        ctc_ids = {id(p) for p in self.ctc_parameters()} if self.ctc_weight > 0.0 else set()
        params = [p for p in self.parameters() if id(p) not in ctc_ids]
        # under torch.no_grad() (evaluation / decoding) no activation is kept for a backward pass
        need_grad = torch.is_grad_enabled() and (input_features.requires_grad or any(p.requires_grad for p in params))
        out = _EncoderFn.apply(self, input_features, stno_mask, enrollments, need_grad, *params)
        if return_logits:                                   # encoder.py:233-240 (CTC pre-training / get_enc_logits)
            if self.ctc_weight <= 0.0:
                raise L.DicowError("return_logits needs the CTC head (ctc_weight > 0)")
            logits = _CtcLogitsFn.apply(self, out, *self.ctc_parameters())
            return ModelOutput(loss=None, logits=logits, hidden_states=out)
        if return_dict is False:
            return (out,)
        return ModelOutput(last_hidden_state=out, hidden_states=None, attentions=None)


class _CtcFn(torch.autograd.Function):
    """CTC auxiliary loss on the encoder output (reference modeling_dicow.py:326-336)."""

    @staticmethod
    def forward(ctx, enc, enc_out, ctc_labels, *params):
        eng = enc._ctc_engine()
        B, T, D = enc_out.shape
        enc_f = enc_out.contiguous().to(F32).view(B * T, D)
        enc_bf = ops.cast_bf16(enc_f).view(B * T, D)
        loss, S = eng.forward(enc_bf, B, T, ctc_labels, enc_f)
        ctx.enc, ctx.S, ctx.params = enc, S, params
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        enc, S = ctx.enc, ctx.S
        G = GradSink(ctx.params, g_loss.device)
        d_enc = enc._ctc_engine(prepare=False).backward(S, g_loss, G)
        hook = getattr(enc, "_segment_hook", None)
        if hook is not None:
            hook("ctc")
        return (None, d_enc.view(S.B, S.T, -1), None) + tuple(G.result(p) for p in ctx.params)


class _CtcLogitsFn(torch.autograd.Function):
    """Encoder output -> CTC-head logits as an autograd node (the ``return_logits=True`` path of encoder.py:233-240, which
    the CTC pre-training trainer calls: src/utils/trainers.py:76-101)."""

    @staticmethod
    def forward(ctx, enc, enc_out, *params):
        eng = enc._ctc_engine()
        B, T, D = enc_out.shape
        enc_f = enc_out.contiguous().to(F32).view(B * T, D)
        enc_bf = ops.cast_bf16(enc_f).view(B * T, D)
        S = eng.encode_logits(enc_bf, B, T, enc_f)
        ctx.enc, ctx.S, ctx.params = enc, S, params
        return S.logits.view(B, S.Tn, -1)[:, :, :enc.config.vocab_size + 1]

    @staticmethod
    def backward(ctx, g):
        enc, S = ctx.enc, ctx.S
        eng = enc._ctc_engine(prepare=False)
        cpad, V1 = eng.W.cpad, enc.config.vocab_size + 1
        pad = getattr(enc, "_ctc_dlogits", None)
        if pad is not None and pad.shape == (S.B * S.Tn, cpad) and g.data_ptr() == pad.data_ptr() and g.stride() == (S.Tn * cpad, cpad, 1):
            d = pad                                           # get_loss's own padded gradient buffer: no copy
        else:
            d = torch.zeros(S.B * S.Tn, cpad, dtype=BF16, device=g.device)
            d.view(S.B, S.Tn, cpad)[:, :, :V1].copy_(g)
        enc._ctc_dlogits = None
        G = GradSink(ctx.params, g.device)
        d_enc = eng.backward_from_logits(S, d, G)
        hook = getattr(enc, "_segment_hook", None)
        if hook is not None:
            hook("ctc")
        return (None, d_enc.view(S.B, S.T, -1)) + tuple(G.result(p) for p in ctx.params)


class _CtcLossFn(torch.autograd.Function):
    """CTC loss of given logits (encoder.py:108-135: fp32 log-softmax, blank = last class, reduction "mean", zero_infinity)."""

    @staticmethod
    def forward(ctx, enc, logits, labels):
        B, Tn, V1 = logits.shape
        dev = logits.device
        if logits.dtype == BF16 and logits.stride(2) == 1 and logits.stride(0) == Tn * logits.stride(1):
            buf, ld = logits, logits.stride(1)                # the padded rows produced by _CtcLogitsFn
        else:
            ld = (V1 + 127) // 128 * 128
            buf = torch.zeros(B, Tn, ld, dtype=BF16, device=dev)
            buf[:, :, :V1].copy_(logits)
        lab = labels.contiguous()
        S = NS_(B=B, Tn=Tn, ld=ld, buf=buf)
        S.lse, S.nll, S.tlen = (torch.empty(n, dtype=F32, device=dev) for n in (B * Tn, B, B))
        S.ab = torch.empty(2, B, Tn, 2 * lab.shape[1] + 1, dtype=F32, device=dev)
        S.acc = torch.zeros(1, dtype=F32, device=dev)
        S.labels = lab
        S.ctc = ops.ctc_args(buf, ld, B, Tn, V1, lab, S.lse, S.ab[0], S.ab[1], S.nll, S.tlen, S.acc)
        ops.ctc_loss_fwd(S.ctc)
        ctx.enc, ctx.S, ctx.V1 = enc, S, V1
        return S.acc[0] / B

    @staticmethod
    def backward(ctx, g):
        enc, S = ctx.enc, ctx.S
        d = torch.empty(S.B * S.Tn, S.ld, dtype=BF16, device=g.device)
        S.ctc.d_logits = d.data_ptr()
        ops.ctc_loss_bwd(S.ctc, g.to(F32).reshape(1))
        enc._ctc_dlogits = d
        return None, d.view(S.B, S.Tn, S.ld)[:, :, :ctx.V1], None


def prepare_ctc_labels(labels, config, prefix_tokens, first_task_token):
    """Label preparation of the CTC branch: strip decoder prefix tokens shared by the whole batch, eos -> -100
    (reference modeling_dicow.py:328-333), optionally drop timestamp/task tokens and re-pad (encoder.py:111-113)."""
    lab = labels.clone()
    for tok in prefix_tokens:
        if bool((lab[:, 0] == tok).all()):
            lab = lab[:, 1:]
    lab[lab == config.eos_token_id] = -100
    if config.remove_timestamps_from_ctc:
        keep = lab < first_task_token                      # also keeps the -100 padding, like the reference
        order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)
        lab = torch.gather(lab, 1, order)
        cnt = keep.sum(dim=1, keepdim=True)
        lab = torch.where(torch.arange(lab.shape[1], device=lab.device)[None, :] < cnt, lab, torch.full_like(lab, -100))
        lab = lab[:, :max(int(cnt.max()), 1)]
    if int(lab.max()) >= config.vocab_size:
        raise ValueError(f"Label values must be <= vocab_size: {config.vocab_size}")
    # CTC expects the valid targets as a prefix: move any interior -100 behind them
    valid = lab >= 0
    order = torch.argsort((~valid).to(torch.int8), dim=1, stable=True)
    return torch.gather(lab, 1, order).contiguous()


# ------------------------------------------------------------------------------------------------ decoder
class WhisperDecoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        d = config.d_model
        self.embed_tokens = nn.Embedding(config.vocab_size, d, padding_idx=config.pad_token_id)
        self.embed_positions = nn.Embedding(config.max_target_positions, d)
        self.layers = nn.ModuleList([DecoderLayer(d, config.decoder_ffn_dim) for _ in range(config.decoder_layers)])
        self.layer_norm = nn.LayerNorm(d)


class DiCoW(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.encoder = DiCoWEncoder(config)
        self.decoder = WhisperDecoder(config)

    def get_encoder(self):
        return self.encoder


def shift_tokens_right(input_ids, pad_token_id, decoder_start_token_id):
    out = input_ids.new_zeros(input_ids.shape)
    out[:, 1:] = input_ids[:, :-1].clone()
    out[:, 0] = decoder_start_token_id
    out.masked_fill_(out == -100, pad_token_id)
    return out


class _DecoderLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, enc_out, dec_ids, labels, upp_labels, *params):
        eng = model._engine()
        B, T, D = enc_out.shape
        enc_bf = ops.cast_bf16(enc_out.contiguous().to(F32)).view(B * T, D)
        loss, logits, S = eng.forward(enc_bf, B, T, dec_ids, labels, upp_labels, ts=model._ts_tables)
        ctx.model, ctx.S, ctx.params = model, S, params
        ctx.need_enc = ctx.needs_input_grad[1]
        ctx.mark_non_differentiable(logits)
        if loss is None:
            loss = torch.zeros((), device=enc_out.device)
        return loss, logits

    @staticmethod
    def backward(ctx, g_loss, _g_logits):
        model, S = ctx.model, ctx.S
        dev = g_loss.device
        Wv = model._engine(prepare=False).W
        p_emb = model.model.decoder.embed_tokens.weight
        extra = {id(p_emb): (Wv.vpad - model.config.vocab_size) * model.config.d_model}
        G = GradSink(ctx.params, dev, extra_rows=extra)
        sync = getattr(model, "_split_sync", None)       # trainer.SplitSync: the follower half's decoder gradients after the leader's
        if sync is not None:
            sync.enter("decoder")
        d_enc = model._engine(prepare=False).backward(S, g_loss, G, need_d_enc=ctx.need_enc)
        if d_enc is not None:
            d_enc = d_enc.view(S.B, S.T, -1)
        hook = getattr(model, "_segment_hook", None)
        if hook is not None:
            hook("decoder")
        if sync is not None:
            sync.done("decoder")
        return (None, d_enc, None, None, None) + tuple(G.result(p) for p in ctx.params)


_KEYWORD_ONLY_GENERATE_ARGS = ("temperature",)     # read from generate()'s keywords, never from the generation config


class DiCoWForConditionalGeneration(_ModelBase):
    """``transformers.PreTrainedModel`` (as the reference's class, modeling_dicow.py:224-229) so that the caller the reference really
    uses -- ``transformers.Seq2SeqTrainer`` through ``CustomTrainer`` (src/utils/trainers.py:106-139) -- finds what it asks its model
    for: ``config.to_json_string``, ``generation_config``, ``main_input_name``, ``floating_point_ops``, ``save_pretrained(dir,
    state_dict=...)`` with the tied head, ``can_generate``.  Loading / saving / initialisation / generation are this class's own
    (below); HF's base supplies the bookkeeping.  ``tests/test_gpu_hf_trainer.py`` runs the installed Seq2SeqTrainer over it."""
    config_class = DiCoWConfig
    base_model_prefix = "model"
    main_input_name = "input_features"
    supports_gradient_checkpointing = False
    _tied_weights_keys = {"proj_out.weight": "model.decoder.embed_tokens.weight"}
    _keys_to_ignore_on_save = None

    @classmethod
    def can_generate(cls) -> bool:            # (HF only looks for GenerationMixin in the bases; generate() below is this class's own)
        return True

    def __init__(self, config: DiCoWConfig):
        if _HF:
            super().__init__(config)          # .config, .generation_config (GenerationConfig.from_model_config), .loss_type, ...
        else:
            super().__init__()
            self.config = config
            self.generation_config = None
        self.model = DiCoW(config)
        self.proj_out = nn.Linear(config.d_model, config.vocab_size, bias=False)
        self.tokenizer = None
        self.soft_label_creator = None
        self._ts_tables = None
        self._eng = None
        self._sig = None
        self.apply(self._init_weights)
        for m in self.modules():              # (HF's post_init / from_pretrained re-initialise whatever does not say it has been)
            m._is_hf_initialized = True
        self.tie_weights()
        if _HF:
            self.all_tied_weights_keys = dict(self._tied_weights_keys)
        # reference checkpoints saved after set_tokenizer() carry the dense [n_ts, V] smoothing buffer
        # (modeling_dicow.py:33); this implementation rebuilds compact tables from the tokenizer instead
        self._register_load_state_dict_pre_hook(self._drop_soft_label_buffer)

    @staticmethod
    def _drop_soft_label_buffer(state_dict, prefix, *args):
        for k in [k for k in state_dict if k.startswith(prefix + "soft_label_creator.")]:
            state_dict.pop(k)

    # -- initialisation (HF Whisper init_std 0.02 + the reference's FDDT / gate / SCB schemes)
    def _init_weights(self, m):
        if isinstance(m, (DiagonalLinear, DenseLinear, Gate)):
            m.reset_parameters()
        elif isinstance(m, (nn.Linear, nn.Conv1d)):
            nn.init.normal_(m.weight, mean=0.0, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Embedding):
            nn.init.normal_(m.weight, mean=0.0, std=0.02)
            if m.padding_idx is not None:
                with torch.no_grad():
                    m.weight[m.padding_idx].zero_()
        elif isinstance(m, nn.LayerNorm):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)
        if isinstance(m, DiCoWEncoder):
            with torch.no_grad():
                m.embed_positions.weight.copy_(sinusoids(*m.embed_positions.weight.shape))

    def tie_weights(self, missing_keys=None, recompute_mapping=True):          # (HF's signature; the tie itself is unconditional)
        self.proj_out.weight = self.model.decoder.embed_tokens.weight

    def estimate_tokens(self, input_dict):
        x = input_dict.get(self.main_input_name) if isinstance(input_dict, dict) else None
        return int(x.numel()) if x is not None else 0

    def floating_point_ops(self, input_dict, exclude_embeddings=True):
        """What HF's Trainer accumulates into ``total_flos`` (transformers 4.55 ModuleUtilsMixin.floating_point_ops, the version the
        reference pins: 6 x tokens x parameters, tokens = elements of the main input)."""
        n = sum(p.numel() for name, p in self.named_parameters()
                if not (exclude_embeddings and ("embed_tokens" in name or "embed_positions" in name)))
        return 6 * self.estimate_tokens(input_dict) * n

    def get_input_embeddings(self):
        return self.model.decoder.embed_tokens

    def set_input_embeddings(self, value):
        self.model.decoder.embed_tokens = value
        self.tie_weights()

    # -- checkpoints: the reference builds the model with ``from_pretrained(name, **overrides)`` (containers.py:47-50)
    @classmethod
    def from_pretrained(cls, name_or_path, *model_args, **overrides):
        """Local directory with ``config.json`` + ``model.safetensors`` / ``pytorch_model.bin`` (an HF Whisper or a DiCoW
        checkpoint: identical key names), or a preset name such as ``openai/whisper-large-v3-turbo`` (no network in this
        build: the preset gives the architecture, weights stay randomly initialised).  Keys the checkpoint lacks -- the
        FDDT / SCB / CTC modules of a plain Whisper checkpoint -- keep their reference initialisation, like HF's
        ``from_pretrained`` does for newly added modules."""
        import json
        import os
        if os.path.isdir(str(name_or_path)):
            with open(os.path.join(name_or_path, "config.json")) as f:
                cfg = DiCoWConfig.from_hf(json.load(f), **overrides)
            model = cls(cfg)
            gc_path = os.path.join(name_or_path, "generation_config.json")
            if _GenerationConfig is not None and os.path.exists(gc_path):
                model.generation_config = _GenerationConfig.from_pretrained(name_or_path)
            st_path, bin_path = os.path.join(name_or_path, "model.safetensors"), os.path.join(name_or_path, "pytorch_model.bin")
            if os.path.exists(st_path):
                from safetensors.torch import load_file
                sd = load_file(st_path)
            elif os.path.exists(bin_path):
                sd = torch.load(bin_path, map_location="cpu", weights_only=True)
            else:
                raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {name_or_path}")
            if "proj_out.weight" not in sd and "model.decoder.embed_tokens.weight" in sd:
                sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]          # safetensors drops the tied copy
            missing, unexpected = model.load_state_dict(sd, strict=False)
            model.tie_weights()
            model._load_report = {"missing": list(missing), "unexpected": list(unexpected)}
            return model
        model = cls(DiCoWConfig.preset(str(name_or_path), **overrides))
        model._load_report = {"missing": None, "unexpected": None, "note": "preset architecture, random initialisation (offline build)"}
        return model

    def save_pretrained(self, directory, state_dict=None, safe_serialization=True, **kwargs):
        """``config.json`` + ``model.safetensors`` (+ ``generation_config.json``): what ``from_pretrained`` above reads.  Signature of
        ``PreTrainedModel.save_pretrained`` as ``Trainer._save`` calls it (``state_dict=`` an already gathered state dict); the tied
        head is stored once (safetensors refuses tensors that share storage)."""
        import json
        import os
        from safetensors.torch import save_file
        os.makedirs(directory, exist_ok=True)
        with open(os.path.join(directory, "config.json"), "w") as f:
            json.dump(dict(self.config.hot_path_dict(), model_type="DiCoW", architectures=["DiCoWForConditionalGeneration"]), f, indent=1)
        gc = getattr(self, "generation_config", None)
        if gc is not None and hasattr(gc, "save_pretrained"):
            gc.save_pretrained(directory)
        sd = self.state_dict() if state_dict is None else state_dict
        sd = {k: v.detach().cpu().contiguous() for k, v in sd.items() if k != "proj_out.weight"}     # tied
        if safe_serialization:
            save_file(sd, os.path.join(directory, "model.safetensors"), metadata={"format": "pt"})
        else:
            torch.save(sd, os.path.join(directory, "pytorch_model.bin"))

    def generate(self, input_features=None, stno_mask=None, attention_mask=None, decoder_input_ids=None, max_new_tokens=None,
                 max_length=None, generation_config=None, enrollments=None, num_beams=1, return_timestamps=None, use_graphs=False,
                 **kwargs):
        """Greedy or beam-search decoding with the reference's logits-processor chain.  One 30 s window without timestamp
        prediction (or without a tokenizer): one pass, returns prompt + tokens (generation.GreedyDecoder.generate / .beam_search).
        Longer recordings, and one window WITH timestamp prediction: the seek loop of HF's / the reference's generate
        (generation.LongFormDecoder) and the window-relative sequences of _fix_timestamps_from_segmentation.
        ``generation_config``: any object with the HF / reference attribute names (eos_token_id, pad_token_id, suppress_tokens,
        begin_suppress_tokens, return_timestamps, no_timestamps_token_id, max_initial_timestamp_index, ctc_weight, ...).
        The prompt is ``decoder_input_ids`` or [decoder_start_token_id] + the tokenizer's prefix tokens."""
        from .generation import GreedyDecoder
        gc = generation_config if generation_config is not None else self.generation_config
        # (explicit keyword arguments win over the generation config, as in HF's generate)
        # -- except `temperature`, which HF's WhisperGenerationMixin.generate (the method the reference delegates to) takes from
        # the explicit keyword ONLY: a GenerationConfig carries temperature = 1.0 by default under the reference's pinned
        # transformers 4.55 (also when loaded from a Whisper checkpoint's generation_config.json), and greedy / beam decoding
        # never consult it.
        def get(k, d=None):                                   # (an attribute a GenerationConfig holds as None counts as not set)
            if kwargs.get(k) is not None:
                return kwargs[k]
            if k in _KEYWORD_ONLY_GENERATE_ARGS or gc is None:
                return d
            v = getattr(gc, k, None)
            return d if v is None else v
        beams = max(num_beams or 1, get("num_beams", 1) or 1)
        self.stno_mask = stno_mask                            # reference generate() keeps it for detect_language (generation.py:556)
        cfg = self.config
        ts_on = return_timestamps if return_timestamps is not None else get("return_timestamps", False)
        one_window = input_features.shape[-1] == 2 * cfg.max_source_positions
        # Sampling / fallback arguments that are not honoured on a path raise instead of being ignored (cf. _refuse_unsupported):
        # HF samples at a scalar temperature > 0 and runs generate_with_fallback inside its seek loop on every input.
        temps = get("temperature")
        ladder = isinstance(temps, (tuple, list))
        takes_seek_loop = (one_window and ts_on and hasattr(self.tokenizer, "get_vocab")) or input_features.shape[-1] > 2 * cfg.max_source_positions
        if temps is not None and not ladder and float(temps) > 0.0:
            raise NotImplementedError("generate(temperature=t > 0): plain sampling is not implemented; pass a tuple (0.0, 0.2, ...) for "
                                      "the temperature-fallback ladder (GreedyDecoder.generate(temperature=...) samples one window)")
        if ladder and beams > 1:
            raise NotImplementedError("generate(): the temperature-fallback ladder is implemented for greedy decoding (num_beams == 1)")
        if ladder and not takes_seek_loop and any(float(t) > 0.0 for t in temps):
            raise NotImplementedError("generate(): the temperature-fallback ladder runs inside the seek loop (timestamps on and a "
                                      "tokenizer set, or a recording longer than one window); this call takes the single-pass path")
        if get("no_speech_threshold") is not None and get("logprob_threshold") is None and ladder:
            raise ValueError("no_speech_threshold only takes effect together with logprob_threshold (a window is skipped when it is "
                             "unlikely AND silent)")
        # HF's generate -- which the reference's calls (generation.py:558) -- runs its seek loop on EVERY input: a single window
        # that ends in an open timestamp gets a second pass over its tail, and the return value is the segment-derived matrix of
        # _fix_timestamps_from_segmentation.  With timestamps on and a real tokenizer set (the fix-up needs its ids) a
        # one-window input therefore takes the same path as a long recording; all of its frames count as valid by default.
        if one_window and ts_on and hasattr(self.tokenizer, "get_vocab"):
            if attention_mask is None:
                attention_mask = torch.ones(input_features.shape[0], input_features.shape[-1], dtype=torch.long, device=input_features.device)
            return self._generate_long_form(input_features, stno_mask, attention_mask, decoder_input_ids, max_new_tokens, get, beams,
                                            enrollments, gc)
        if input_features.shape[-1] > 2 * cfg.max_source_positions:
            return self._generate_long_form(input_features, stno_mask, attention_mask, decoder_input_ids, max_new_tokens, get, beams,
                                            enrollments, gc)
        if input_features.shape[-1] != 2 * cfg.max_source_positions:
            raise ValueError("input_features shorter than one window: pad the features to 2 * max_source_positions frames")
        B = input_features.shape[0]
        if decoder_input_ids is None and get("lang_to_id") is not None:
            decoder_input_ids = self.retrieve_init_tokens(input_features, stno_mask, gc, enrollments, return_timestamps)
        elif decoder_input_ids is None:
            prefix = list(getattr(self.tokenizer, "prefix_tokens", [])) if self.tokenizer is not None else []
            start = get("decoder_start_token_id", cfg.decoder_start_token_id)
            prompt = prefix if (prefix and prefix[0] == start) else [start] + prefix
            decoder_input_ids = torch.tensor([prompt] * B, dtype=torch.long)
        P = decoder_input_ids.shape[1]
        if max_new_tokens is None:
            limit = max_length if max_length is not None else get("max_length", cfg.max_target_positions)
            max_new_tokens = min(limit, cfg.max_target_positions) - P
        ts = return_timestamps if return_timestamps is not None else get("return_timestamps", False)
        timestamps = None
        if ts:
            timestamps = dict(no_timestamps_token_id=get("no_timestamps_token_id"),
                              max_initial_timestamp_index=get("max_initial_timestamp_index"))
        ctc = None
        if (get("ctc_weight", 0.0) or 0.0) > 0.0:
            tok = self.tokenizer
            ctc = dict(weight=get("ctc_weight"), first_timestamp=tok.get_vocab()["<|0.00|>"],
                       upper_cased=list(getattr(tok, "upper_cased_tokens", {}).items()), prefix_len=len(tok.prefix_tokens))
        if not hasattr(self, "_decoder"):
            self._decoder = GreedyDecoder(self)
        if beams > 1:                                        # reference: generation_num_beams 5 in configs/decode/*_beam_joint.yaml
            seq, _ = self._decoder.beam_search(input_features, stno_mask, decoder_input_ids, P + max_new_tokens, beams,
                                               eos_token_id=get("eos_token_id", cfg.eos_token_id), pad_token_id=get("pad_token_id", cfg.pad_token_id),
                                               length_penalty=get("length_penalty", 1.0), early_stopping=get("early_stopping", False),
                                               suppress_tokens=get("suppress_tokens"), begin_suppress_tokens=get("begin_suppress_tokens"),
                                               enrollments=enrollments, timestamps=timestamps, ctc=ctc)
            return seq
        dec = self._decoder
        if use_graphs:                                       # hipGraph replay of each decoder position (evaluation: fixed weights)
            if not hasattr(self, "_decoder_graphed"):
                self._decoder_graphed = GreedyDecoder(self, use_graphs=True)
            dec = self._decoder_graphed
        return dec.generate(input_features, stno_mask, decoder_input_ids, max_new_tokens,
                                      eos_token_id=get("eos_token_id", cfg.eos_token_id), pad_token_id=get("pad_token_id", cfg.pad_token_id),
                                      suppress_tokens=get("suppress_tokens"), begin_suppress_tokens=get("begin_suppress_tokens"),
                                      enrollments=enrollments, ctc=ctc, timestamps=timestamps)

    def retrieve_init_tokens(self, input_features, stno_mask, generation_config, enrollments=None, return_timestamps=None):
        """The forced prompt per row the way the reference's evaluation gets it (DiCoWGenerationMixin._retrieve_init_tokens,
        generation.py:121-149, falling through to HF's): ``forced_decoder_ids`` is not supported (the reference sets it to None,
        containers.py:67); start token, then the language token -- ``generation_config.language`` (a '<|xx|>' key of
        ``lang_to_id``, a language code, a language name, or one per row) or, when it is None (containers.py:58), the STNO-conditioned
        ``detect_language`` of each row -- then the task token, then ``<|notimestamps|>`` unless timestamps are predicted."""
        gc = generation_config
        B = input_features.shape[0]
        if getattr(gc, "forced_decoder_ids", None) is not None:
            raise NotImplementedError("forced_decoder_ids: use generation_config.language / .task (the reference clears it)")
        start = getattr(gc, "decoder_start_token_id", None)
        start = self.config.decoder_start_token_id if start is None else start
        lang_to_id, language, task = gc.lang_to_id, getattr(gc, "language", None), getattr(gc, "task", None)

        def lang_id(name):
            key = name.lower()
            if key not in lang_to_id:
                if f"<|{key}|>" in lang_to_id:
                    key = f"<|{key}|>"
                else:
                    from transformers.models.whisper.tokenization_whisper import TO_LANGUAGE_CODE
                    if key not in TO_LANGUAGE_CODE or f"<|{TO_LANGUAGE_CODE[key]}|>" not in lang_to_id:
                        raise ValueError(f"Unsupported language: {name}")
                    key = f"<|{TO_LANGUAGE_CODE[key]}|>"
            return lang_to_id[key]

        if isinstance(language, (list, tuple)):
            if len(language) != B or any(l is None for l in language):
                raise ValueError(f"a list of languages must name one language per row ({B})")
            langs = [lang_id(l) for l in language]
        elif language is not None:
            langs = [lang_id(language)] * B
        else:
            langs = self.detect_language(input_features, stno_mask, gc, enrollments).tolist()
        tail = []
        if task is not None:
            if task not in ("transcribe", "translate"):
                raise ValueError(f"The `{task}` task is not supported")
            tail.append(gc.task_to_id[task])
        elif language is not None and getattr(gc, "task_to_id", None) is not None:
            tail.append(gc.task_to_id["transcribe"])
        ts = getattr(gc, "return_timestamps", False) if return_timestamps is None else return_timestamps
        no_ts = getattr(gc, "no_timestamps_token_id", None)
        if not ts and no_ts is not None:
            tail.append(no_ts)
        return torch.tensor([[start, l] + tail for l in langs], dtype=torch.long)

    def _generate_long_form(self, input_features, stno_mask, attention_mask, decoder_input_ids, max_new_tokens, get, beams, enrollments,
                            gc=None):
        """Recordings longer than one window (reference generate(), generation.py:536-564, with HF's seek loop underneath):
        sequential windows at temperature 0 with timestamps, segments per recording, and -- what the reference returns for
        such inputs -- the window-relative token sequences of ``_fix_timestamps_from_segmentation`` (padded LongTensor).  The
        segments themselves are kept in ``self.last_segments``."""
        from .generation import LongFormDecoder, fix_timestamps_from_segmentation
        cfg, tok = self.config, self.tokenizer
        if attention_mask is None:
            raise ValueError("long-form generation needs attention_mask [B, frames] (valid feature frames per recording)")
        if tok is None:
            raise ValueError("long-form generation needs set_tokenizer(): the returned sequences carry its timestamp / prefix ids")
        vocab = tok.get_vocab()
        first_ts = vocab["<|0.00|>"]
        no_ts = get("no_timestamps_token_id", first_ts - 1)
        if decoder_input_ids is None and get("lang_to_id") is not None:               # language from the first window
            decoder_input_ids = self.retrieve_init_tokens(input_features, stno_mask, gc, enrollments, return_timestamps=True)
        elif decoder_input_ids is None:
            prefix = [t for t in getattr(tok, "prefix_tokens", []) if t != no_ts]          # timestamps are predicted
            start = get("decoder_start_token_id", cfg.decoder_start_token_id)
            decoder_input_ids = torch.tensor([prefix if (prefix and prefix[0] == start) else [start] + prefix], dtype=torch.long)
        ctc = None
        if (get("ctc_weight", 0.0) or 0.0) > 0.0:
            ctc = dict(weight=get("ctc_weight"), first_timestamp=first_ts,
                       upper_cased=list(getattr(tok, "upper_cased_tokens", {}).items()), prefix_len=len(tok.prefix_tokens))
        eos = get("eos_token_id", cfg.eos_token_id)
        dec = LongFormDecoder(self, beams)
        fb = {}
        temps = get("temperature")
        if isinstance(temps, (tuple, list)):          # HF: a tuple of temperatures switches the fallback ladder on
            fb = dict(temperatures=tuple(float(t) for t in temps), compression_ratio_threshold=get("compression_ratio_threshold"),
                      logprob_threshold=get("logprob_threshold"), no_speech_threshold=get("no_speech_threshold"),
                      no_speech_token_id=no_ts - 1, generator=get("generator"))   # HF: <|nospeech|> = <|notimestamps|> - 1
        segs = dec.transcribe(input_features, stno_mask, attention_mask.sum(-1).cpu().tolist(), decoder_input_ids, no_ts,
                              eos_token_id=eos, pad_token_id=get("pad_token_id", cfg.pad_token_id), max_new_tokens=max_new_tokens,
                              enrollments=enrollments, suppress_tokens=get("suppress_tokens"),
                              begin_suppress_tokens=get("begin_suppress_tokens"),
                              max_initial_timestamp_index=get("max_initial_timestamp_index", 50),
                              length_penalty=get("length_penalty", 1.0), early_stopping=get("early_stopping", False), ctc=ctc, **fb)
        self.last_segments = segs
        pad_id = getattr(tok, "pad_token_id", None)
        return fix_timestamps_from_segmentation(segs, first_ts, vocab["\u0120"], cfg.pad_token_id if pad_id is None else pad_id,
                                                prefix_ids=list(getattr(tok, "prefix_tokens", [])), suffix_ids=[eos],
                                                device=input_features.device)

    def detect_language(self, input_features=None, stno_mask=None, generation_config=None, enrollments=None,
                        num_segment_frames=None, lang_token_ids=None):
        """Language id per row from one decoder position (reference generation.py:151-221).  ``stno_mask`` defaults to the one
        stored by the last generate() call, as in the reference; language tokens come from ``generation_config.lang_to_id``."""
        from .generation import GreedyDecoder
        gc = generation_config if generation_config is not None else self.generation_config
        if lang_token_ids is None:
            lang_token_ids = list(getattr(gc, "lang_to_id").values())
        if stno_mask is None:
            stno_mask = self.stno_mask
        if not hasattr(self, "_decoder"):
            self._decoder = GreedyDecoder(self)
        start = getattr(gc, "decoder_start_token_id", None) if gc is not None else None
        return self._decoder.detect_language(input_features, stno_mask, lang_token_ids, start, enrollments)

    def post_init(self):
        """The reference's container calls it after ``from_pretrained`` (containers.py:52).  Nothing is re-initialised here (every module
        is marked initialised by __init__); HF's version records its per-model bookkeeping and ties."""
        if _HF:
            super().post_init()
        self.tie_weights()

    def get_encoder(self):
        return self.model.encoder

    def get_enc_logits(self, hidden_states):
        """CTC head on the encoder output (reference modeling_dicow.py get_enc_logits -> encoder.py:87-106), used by the
        CTC-rescored decoding: returns bf16 logits [B, Tn, vocab_size + 1] (a view into a 128-padded row, blank last)."""
        enc = self.model.encoder
        if enc.ctc_weight <= 0.0:
            raise L.DicowError("get_enc_logits: the model was built without a CTC head (ctc_weight == 0)")
        _require_cuda(hidden_states, "get_enc_logits")
        B, T, D = hidden_states.shape
        with torch.no_grad():
            enc_f = hidden_states.detach().contiguous().to(F32).view(B * T, D)
            enc_bf = ops.cast_bf16(enc_f).view(B * T, D)
            S = enc._ctc_engine().encode_logits(enc_bf, B, T, enc_f)
        return S.logits.view(B, S.Tn, -1)[:, :, :self.config.vocab_size + 1]

    def get_decoder(self):
        return self.model.decoder

    def get_output_embeddings(self):
        return self.proj_out

    def set_tokenizer(self, tokenizer):
        """Enables the soft-label (timestamp-smoothed) loss, as reference modeling_dicow.py:237-240."""
        self.tokenizer = tokenizer
        self._ts_tables = build_ts_tables(tokenizer.get_vocab(), self.config.vocab_size,
                                          self.model.decoder.embed_tokens.weight.device)
        self.soft_label_creator = self._ts_tables is not None or True

    def _engine(self, prepare=True):
        if self._eng is None:
            self._eng = DecoderEngine(self)
        if prepare:
            sig = _param_sig(list(self.model.decoder.parameters()))
            if sig != self._sig:
                self._eng.prepare()
                self._sig = sig
        return self._eng

    def forward(self, input_features=None, attention_mask=None, stno_mask=None, decoder_input_ids=None,
                decoder_attention_mask=None, head_mask=None, decoder_head_mask=None, cross_attn_head_mask=None,
                encoder_outputs=None, past_key_values=None, decoder_inputs_embeds=None, decoder_position_ids=None,
                labels=None, upp_labels=None, use_cache=None, output_attentions=None, output_hidden_states=None,
                return_dict=None, cache_position=None, forced_decoder_ids=None, enrollments=None):
        cfg = self.config
        if labels is not None and decoder_input_ids is None and decoder_inputs_embeds is None:
            decoder_input_ids = shift_tokens_right(labels, cfg.pad_token_id, cfg.decoder_start_token_id)
        if decoder_inputs_embeds is not None or past_key_values is not None:
            raise NotImplementedError("decoder_inputs_embeds / KV-cache decoding are outside the training-step path")
        # arguments of the HF signature this path has no kernels for: refuse instead of silently computing something else
        # (attention_mask is accepted and unused exactly as in HF Whisper, whose encoder does not mask)
        _refuse_unsupported("DiCoWForConditionalGeneration.forward", decoder_attention_mask=decoder_attention_mask,
                            head_mask=head_mask, decoder_head_mask=decoder_head_mask, cross_attn_head_mask=cross_attn_head_mask,
                            decoder_position_ids=decoder_position_ids, cache_position=cache_position,
                            forced_decoder_ids=forced_decoder_ids, output_attentions=output_attentions or None,
                            output_hidden_states=output_hidden_states or None,
                            use_cache=(use_cache or None) if labels is None else None)
        if encoder_outputs is None:
            _require_cuda(input_features, "DiCoWForConditionalGeneration")
            enc_out = self.model.encoder(input_features, stno_mask=stno_mask, enrollments=enrollments).last_hidden_state
        else:
            enc_out = encoder_outputs[0]
        if decoder_input_ids is None:
            raise ValueError("either labels or decoder_input_ids must be given")
        dev = enc_out.device
        dec_ids = decoder_input_ids.to(dev)
        lab = None if labels is None else labels.to(dev)
        upp = None if upp_labels is None else upp_labels.to(dev)
        if self._ts_tables is not None and self._ts_tables["ids"].device != dev:
            self._ts_tables = {k: v.to(dev) for k, v in self._ts_tables.items()}
        params = list(self.model.decoder.parameters())
        loss, logits = _DecoderLossFn.apply(self, enc_out, dec_ids, lab, upp, *params)
        if labels is None:
            loss = None
        elif cfg.ctc_weight > 0.0:                         # modeling_dicow.py:326-336
            enc = self.model.encoder
            prefix = getattr(self.tokenizer, "prefix_tokens", []) if self.tokenizer is not None else []
            ctc_lab = prepare_ctc_labels(lab, cfg, prefix, enc.first_task_token)
            ctc_loss = _CtcFn.apply(enc, enc_out, ctc_lab, *enc.ctc_parameters())
            loss = (1 - cfg.ctc_weight) * loss + cfg.ctc_weight * ctc_loss
        if return_dict is False:
            return tuple(x for x in (loss, logits, enc_out) if x is not None)
        return ModelOutput(loss=loss, logits=logits, past_key_values=None, decoder_hidden_states=None,
                           decoder_attentions=None, cross_attentions=None, encoder_last_hidden_state=enc_out,
                           encoder_hidden_states=None, encoder_attentions=None)


def sinusoids(length, channels, max_timescale=10000.0):
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    t = torch.arange(length)[:, None] * inv[None, :]
    return torch.cat([t.sin(), t.cos()], dim=1)


def build_ts_tables(vocab, vocab_size, device, sigma=0.08):
    """Timestamp smoothing tables for the soft-label loss: sorted timestamp ids, row-normalised Gaussian
    weights over them, and the token-id -> timestamp-row lookup (what reference modeling_dicow.py:35-93 keeps as a
    dense [n_ts, V] matrix)."""
    pat = re.compile(r"<\|(\d+\.\d+)\|>")
    pairs = sorted((tid, float(mt.group(1))) for tok, tid in vocab.items() for mt in [pat.match(tok)] if mt)
    if not pairs:
        return None
    ids = torch.tensor([p[0] for p in pairs], dtype=torch.int32)
    times = torch.tensor([p[1] for p in pairs], dtype=torch.float32)
    w = torch.exp(-((times[:, None] - times[None, :]) ** 2) / (2 * sigma ** 2))
    w = w / w.sum(dim=1, keepdim=True)
    index = torch.full((max(vocab_size, int(ids.max()) + 1),), -1, dtype=torch.int32)
    index[ids.long()] = torch.arange(len(pairs), dtype=torch.int32)
    return {"ids": ids.to(device), "w": w.contiguous().to(device), "index": index.to(device)}
