"""Oracle (test infrastructure, plain PyTorch CPU): DiCoW / SE-DiCoW forward (+autograd backward).

A from-scratch functional restatement of the reference's training-step arithmetic; it
uses only elementary torch ops (matmul, exp, erf, sum ...) on a flat ``{name: tensor}``
state dict whose keys equal the reference's ``state_dict()`` keys.
No HuggingFace / reference code is imported here.

Reference call sites restated (paths relative to /root/reference):
  FDDT (diag / full / bias-only, class switches)  src/models/dicow/FDDT.py:41-63, layers.py:73-77
  encoder forward (stem, initial FDDT, pos-emb, layer loop, SCB, final LN)
                                                   src/models/dicow/encoder.py:140-246
  speaker communication block                      src/models/dicow/layers.py:145-193, Gate :86-89
  model forward + tied LM head                     src/models/dicow/modeling_dicow.py:152-221, 248-302
  hard-label fallback loss                         src/models/dicow/modeling_dicow.py:310-323
  soft-label (timestamp-smoothed) loss             src/models/dicow/modeling_dicow.py:35-144
Third-party arithmetic reached by those call sites (transformers==4.55.0 per the reference's
requirements.txt:22, not vendored): WhisperAttention (q pre-scaled by head_dim**-0.5, k_proj
without bias, softmax(QK^T)V, out_proj), pre-LN encoder/decoder layers with exact-erf GELU,
learned decoder positions, LayerNorm eps 1e-5, shift_tokens_right.

``emulate_bf16=True`` rounds to bf16 exactly where the reference's ``bf16: true`` AMP policy
does (configs/base.yaml:49; SURVEY.md section 5 precision probe): GEMM/conv/attention inputs
and outputs bf16, LayerNorm / FDDT / residual stream / loss fp32.  It is what the HIP path is
compared against at tight tolerance; the fp32 mode is what the goldens pin.
"""
from dataclasses import dataclass, field
from typing import Dict, Optional
import math
import re

import torch

T = torch.Tensor


@dataclass
class OracleConfig:
    vocab_size: int = 51865
    num_mel_bins: int = 80
    d_model: int = 384
    encoder_layers: int = 4
    encoder_attention_heads: int = 6
    decoder_layers: int = 4
    decoder_attention_heads: int = 6
    encoder_ffn_dim: int = 1536
    decoder_ffn_dim: int = 1536
    max_source_positions: int = 1500
    max_target_positions: int = 448
    pad_token_id: int = 50257
    decoder_start_token_id: int = 50258
    # DiCoW switches (reference src/models/dicow/config.py:11-59)
    use_fddt: bool = True
    fddt_is_diagonal: bool = True
    fddt_bias_only: bool = False
    fddt_use_silence: bool = True
    fddt_use_target: bool = True
    fddt_use_overlap: bool = True
    fddt_use_non_target: bool = True
    apply_fddt_to_n_layers: int = -1
    use_pre_pos_fddt: bool = False
    use_enrollments: bool = False
    scb_layers: Optional[int] = None
    ctc_weight: float = 0.0
    additional_self_attention_layer: bool = False
    additional_layer: bool = False
    pre_ctc_sub_sample: bool = False
    remove_timestamps_from_ctc: bool = False
    ctc_loss_reduction: str = "mean"
    eos_token_id: int = 50257

    @property
    def n_fddts(self):
        return self.encoder_layers if self.apply_fddt_to_n_layers == -1 else self.apply_fddt_to_n_layers


# --------------------------------------------------------------------------- primitives

def _r(x: T, emu: bool) -> T:
    """Round to bf16 and back when emulating the AMP policy."""
    return x.to(torch.bfloat16).to(torch.float32) if emu else x


def gelu_erf(x: T) -> T:
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def layer_norm(x: T, w: T, b: T, eps: float = 1e-5) -> T:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def linear(x: T, w: T, b: Optional[T], emu: bool) -> T:
    y = _r(x, emu) @ _r(w, emu).t()
    if b is not None:
        y = y + b
    return _r(y, emu)


def conv1d_k3(x: T, w: T, b: T, stride: int, emu: bool) -> T:
    """x [B,C,L], w [O,C,3], padding 1 -> [B,O,L/stride]; written as three shifted matmuls."""
    x = _r(x, emu)
    w = _r(w, emu)
    xp = torch.nn.functional.pad(x, (1, 1))
    l_out = (x.shape[-1] + 2 - 3) // stride + 1
    y = 0
    for tap in range(3):
        xs = xp[:, :, tap: tap + stride * (l_out - 1) + 1: stride]          # [B,C,l_out]
        y = y + torch.einsum("oc,bcl->bol", w[:, :, tap], xs)
    return _r(y + b[None, :, None], emu)


def attention(q_in: T, kv_in: T, p: Dict[str, T], prefix: str, n_heads: int, causal: bool, emu: bool) -> T:
    """WhisperAttention: q=(x Wq^T + bq)*hd^-0.5, k = x Wk^T (no bias), v = x Wv^T + bv."""
    B, Lq, D = q_in.shape
    Lk = kv_in.shape[1]
    hd = D // n_heads
    q = _r(linear(q_in, p[prefix + "q_proj.weight"], p[prefix + "q_proj.bias"], emu) * (hd ** -0.5), emu)
    k = linear(kv_in, p[prefix + "k_proj.weight"], None, emu)
    v = linear(kv_in, p[prefix + "v_proj.weight"], p[prefix + "v_proj.bias"], emu)
    q = q.view(B, Lq, n_heads, hd).transpose(1, 2)
    k = k.view(B, Lk, n_heads, hd).transpose(1, 2)
    v = v.view(B, Lk, n_heads, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if causal:
        mask = torch.ones(Lq, Lk, dtype=torch.bool, device=s.device).tril()
        s = s.masked_fill(~mask, float("-inf"))
    m = s.max(dim=-1, keepdim=True).values
    e = torch.exp(s - m)
    l = e.sum(dim=-1, keepdim=True)
    if emu:
        # flash-style: unnormalised probabilities rounded to bf16 for the PV product, fp32 row sum
        o = (_r(e, True) @ v) / l
    else:
        o = (e / l) @ v
    o = _r(o, emu).transpose(1, 2).reshape(B, Lq, D)
    return linear(o, p[prefix + "out_proj.weight"], p[prefix + "out_proj.bias"], emu)


# --------------------------------------------------------------------------- FDDT

_FDDT_CLASSES = (("silence_linear", "fddt_use_silence"), ("target_linear", "fddt_use_target"),
                 ("non_target_linear", "fddt_use_non_target"), ("overlap_linear", "fddt_use_overlap"))


def fddt(h: T, stno: T, p: Dict[str, T], prefix: str, cfg: OracleConfig, emu: bool = False) -> T:
    """h [B,T,D], stno [B,4,T] (S,T,N,O) -> [B,T,D] fp32.  FDDT.py:41-63."""
    m = stno[..., None]                                     # [B,4,T,1]
    if cfg.fddt_bias_only:
        out = h
        for c, (name, flag) in enumerate(_FDDT_CLASSES):
            if getattr(cfg, flag):
                out = out + m[:, c] * p[prefix + name]
        return out
    out = 0
    for c, (name, flag) in enumerate(_FDDT_CLASSES):
        if getattr(cfg, flag):
            w, b = p[prefix + name + ".weight"], p[prefix + name + ".bias"]
            if cfg.fddt_is_diagonal:
                t = h * w + b                               # layers.py:73-77 (fp32 by promotion)
            else:
                t = linear(h, w, b, emu)
        else:
            t = h
        out = out + t * m[:, c]
    return out


# --------------------------------------------------------------------------- encoder / decoder

def encoder_layer(h: T, p: Dict[str, T], pre: str, n_heads: int, emu: bool) -> T:
    x = layer_norm(h, p[pre + "self_attn_layer_norm.weight"], p[pre + "self_attn_layer_norm.bias"])
    h = h + attention(x, x, p, pre + "self_attn.", n_heads, False, emu)
    x = layer_norm(h, p[pre + "final_layer_norm.weight"], p[pre + "final_layer_norm.bias"])
    x = _r(gelu_erf(linear(x, p[pre + "fc1.weight"], p[pre + "fc1.bias"], emu)), emu)
    return h + linear(x, p[pre + "fc2.weight"], p[pre + "fc2.bias"], emu)


def scb(h: T, p: Dict[str, T], pre: str, cfg: OracleConfig, emu: bool) -> T:
    """SpeakerCommunicationBlock on interleaved [2B,T,D] (even=mixture, odd=enrollment). layers.py:145-193."""
    B2, Tn, D = h.shape
    x = h.view(B2 // 2, 2, Tn, D)
    q, kv = x[:, 0], x[:, 1]
    a = attention(q, kv, p, pre + "cae.cross_attn.", cfg.encoder_attention_heads, False, emu)
    cat = torch.cat([a, q], dim=-1)
    u = _r(gelu_erf(linear(cat, p[pre + "cae.ffn.0.weight"], p[pre + "cae.ffn.0.bias"], emu)), emu)
    u = linear(u, p[pre + "cae.ffn.3.weight"], p[pre + "cae.ffn.3.bias"], emu)
    q_out = q + torch.tanh(p[pre + "cae.cross_gate.gate"]) * u
    return torch.stack([q_out, kv], dim=1).view(B2, Tn, D)


def encoder_forward(p: Dict[str, T], cfg: OracleConfig, input_features: T, stno_mask: T,
                    enrollments: Optional[Dict[str, T]] = None, emu: bool = False, collect=None) -> T:
    """encoder.py:140-246.  Returns last_hidden_state [B,T,D] fp32."""
    e = "model.encoder."
    if enrollments is not None:
        input_features = torch.stack((input_features, enrollments["input_features"]), dim=1).flatten(0, 1)
        stno_mask = torch.stack((stno_mask, enrollments["stno_mask"]), dim=1).flatten(0, 1)
    if input_features.shape[-1] != 2 * cfg.max_source_positions:
        raise ValueError("mel length must be 2*max_source_positions")
    x = _r(gelu_erf(conv1d_k3(input_features, p[e + "conv1.weight"], p[e + "conv1.bias"], 1, emu)), emu)
    x = _r(gelu_erf(conv1d_k3(x, p[e + "conv2.weight"], p[e + "conv2.bias"], 2, emu)), emu)
    h = x.permute(0, 2, 1)
    if collect is not None:
        collect["stem"] = h
    if cfg.use_fddt and cfg.use_pre_pos_fddt:
        h = fddt(h, stno_mask, p, e + "initial_fddt.", cfg, emu)
    h = h + p[e + "embed_positions.weight"]
    for i in range(cfg.encoder_layers):
        if cfg.use_fddt and i < cfg.n_fddts:
            h = fddt(h, stno_mask, p, f"{e}fddts.{i}.", cfg, emu)
        if cfg.use_enrollments and cfg.scb_layers is not None and i < cfg.scb_layers:
            h = scb(h, p, f"{e}ca_enrolls.{i}.", cfg, emu)
            if i == cfg.scb_layers - 1:
                h = h[::2]
                stno_mask = stno_mask[::2]
        h = encoder_layer(h, p, f"{e}layers.{i}.", cfg.encoder_attention_heads, emu)
        if collect is not None:
            collect[f"layer{i}"] = h
    return layer_norm(h, p[e + "layer_norm.weight"], p[e + "layer_norm.bias"])


def shift_tokens_right(labels: T, pad_id: int, start_id: int) -> T:
    out = labels.new_zeros(labels.shape)
    out[:, 1:] = labels[:, :-1]
    out[:, 0] = start_id
    return out.masked_fill(out == -100, pad_id)


def decoder_forward(p: Dict[str, T], cfg: OracleConfig, input_ids: T, enc: T, emu: bool = False) -> T:
    d = "model.decoder."
    L = input_ids.shape[1]
    h = p[d + "embed_tokens.weight"][input_ids] + p[d + "embed_positions.weight"][:L]
    nh = cfg.decoder_attention_heads
    for i in range(cfg.decoder_layers):
        pre = f"{d}layers.{i}."
        x = layer_norm(h, p[pre + "self_attn_layer_norm.weight"], p[pre + "self_attn_layer_norm.bias"])
        h = h + attention(x, x, p, pre + "self_attn.", nh, True, emu)
        x = layer_norm(h, p[pre + "encoder_attn_layer_norm.weight"], p[pre + "encoder_attn_layer_norm.bias"])
        h = h + attention(x, enc, p, pre + "encoder_attn.", nh, False, emu)
        x = layer_norm(h, p[pre + "final_layer_norm.weight"], p[pre + "final_layer_norm.bias"])
        x = _r(gelu_erf(linear(x, p[pre + "fc1.weight"], p[pre + "fc1.bias"], emu)), emu)
        h = h + linear(x, p[pre + "fc2.weight"], p[pre + "fc2.bias"], emu)
    return layer_norm(h, p[d + "layer_norm.weight"], p[d + "layer_norm.bias"])


# --------------------------------------------------------------------------- losses

def _lse_and_pick(logits: T, labels: T):
    m = logits.max(dim=-1, keepdim=True).values
    lse = (m + torch.log(torch.exp(logits - m).sum(dim=-1, keepdim=True))).squeeze(-1)
    return lse


def hard_loss(logits: T, labels: T, upp_labels: Optional[T]) -> T:
    """modeling_dicow.py:310-323: CE(ignore_index=-100, reduction none) for both label sets,
    per-token min, mean over ALL B*L positions (ignored positions contribute 0)."""
    V = logits.shape[-1]
    fl = logits.reshape(-1, V).float()
    lse = _lse_and_pick(fl, None)

    def ce(lab):
        lab = lab.reshape(-1)
        valid = lab != -100
        picked = fl.gather(1, lab.clamp(min=0)[:, None]).squeeze(1)
        return torch.where(valid, lse - picked, torch.zeros_like(lse))

    l1 = ce(labels)
    if upp_labels is None:
        return l1.mean()
    return torch.minimum(l1, ce(upp_labels)).mean()


def build_ts_smoothing(vocab: Dict[str, int], sigma: float = 0.08):
    """modeling_dicow.py:35-72.  Returns (sorted timestamp ids [n], weights [n, n]) where row i is
    the normalised Gaussian over the n timestamp tokens (the dense [n, V] matrix has these
    weights scattered at the timestamp ids and zeros elsewhere); None if no timestamp tokens."""
    pat = re.compile(r"<\|(\d+\.\d+)\|>")
    id_to_time = {}
    for tok, tid in vocab.items():
        mt = pat.match(tok)
        if mt:
            id_to_time[tid] = float(mt.group(1))
    if not id_to_time:
        return None
    ids = sorted(id_to_time)
    times = torch.tensor([id_to_time[i] for i in ids])
    w = torch.exp(-((times[:, None] - times[None, :]) ** 2) / (2 * sigma ** 2))
    w = w / w.sum(dim=1, keepdim=True)
    return torch.tensor(ids), w


def soft_loss(logits: T, labels: T, upp_labels: Optional[T], ts) -> T:
    """modeling_dicow.py:95-144: soft-target CE (timestamp rows Gaussian-smoothed), per-token
    min(lower, upper), masked by labels != -100, sum / max(count, 1)."""
    V = logits.shape[-1]
    fl = logits.reshape(-1, V).float()
    logp = fl - _lse_and_pick(fl, None)[:, None]

    def ce(lab):
        lab = lab.reshape(-1)
        soft = torch.nn.functional.one_hot(lab.clamp(min=0), V).float()
        if ts is not None:
            ids, w = ts
            is_ts = torch.isin(lab, ids)
            if is_ts.any():
                row = torch.searchsorted(ids, lab[is_ts])
                dense = torch.zeros(int(is_ts.sum()), V)
                dense[:, ids] = w[row]
                soft[is_ts] = dense
        return -(soft * logp).sum(dim=-1)

    flat = labels.reshape(-1)
    mask = (flat != -100).float()
    lo = ce(labels) * mask
    up = ce(upp_labels) * mask if upp_labels is not None else lo
    return torch.minimum(lo, up).sum() / mask.sum().clamp(min=1)


# --------------------------------------------------------------------------- CTC auxiliary branch
def conv1d_k3s2_nobias(x: T, w: T, emu: bool) -> T:
    """[B,T,D] time-major in/out, kernel 3 stride 2 padding 1, no bias (encoder.py:26-41)."""
    y = conv1d_k3(x.transpose(1, 2), w, torch.zeros(w.shape[0]), 2, emu)
    return y.transpose(1, 2)


def ctc_logits(p: Dict[str, T], cfg: OracleConfig, enc: T, emu: bool = False) -> T:
    """get_enc_logits (modeling_dicow.py:242-246) -> possibly_update_last_hidden_states (encoder.py:87-106)."""
    e = "model.encoder."
    h = enc
    if cfg.additional_layer:
        # a full pre-LN encoder layer (residuals included) takes precedence over the bare attention: encoder.py:88-94
        h = encoder_layer(h, p, e + "additional_layer.", cfg.encoder_attention_heads, emu)
    elif cfg.additional_self_attention_layer:
        # the attention output REPLACES the hidden states (no residual, no LayerNorm): encoder.py:95-101
        h = attention(h, h, p, e + "additional_self_attention_layer.", cfg.encoder_attention_heads, False, emu)
    if cfg.pre_ctc_sub_sample:
        h = conv1d_k3s2_nobias(h, p[e + "subsample_conv1.weight"], emu)
        h = conv1d_k3s2_nobias(h, p[e + "subsample_conv2.weight"], emu)
    return linear(h, p[e + "lm_head.weight"], None, emu)


def ctc_prepare_labels(labels: T, cfg: OracleConfig, prefix_tokens) -> T:
    """modeling_dicow.py:328-333 + encoder.py:111-113."""
    lab = labels.clone()
    for tok in prefix_tokens:
        if bool((lab[:, 0] == tok).all()):
            lab = lab[:, 1:]
    lab[lab == cfg.eos_token_id] = -100
    if cfg.remove_timestamps_from_ctc:
        first_task_token = cfg.vocab_size - 30 * 50 - 1 - 6
        rows = [r[r < first_task_token] for r in lab]
        width = max(int(r.numel()) for r in rows)
        out = lab.new_full((len(rows), max(width, 1)), -100)
        for i, r in enumerate(rows):
            out[i, :r.numel()] = r
        lab = out
    return lab


def ctc_loss(logits: T, labels: T, reduction: str = "mean") -> T:
    """Plain log-domain CTC forward algorithm (blank = last class, zero_infinity=True), as
    torch.nn.functional.ctc_loss is called at encoder.py:119-134.  logits [B,T,C], labels [B,L] (-100 padded)."""
    B, Tn, C = logits.shape
    blank = C - 1
    lp = logits.float() - _lse_and_pick(logits.float(), None)[..., None]
    tl = (labels >= 0).sum(-1)
    Lmax = max(int(tl.max()), 1)
    ext = labels.new_full((B, 2 * Lmax + 1), blank)
    ext[:, 1::2] = labels[:, :Lmax].clamp(min=0)
    S = 2 * tl + 1
    neg = -1e30                      # finite "minus infinity": keeps autograd free of inf * 0 = NaN
    idx = torch.arange(2 * Lmax + 1)[None, :]
    valid = idx < S[:, None]
    skip = torch.zeros_like(valid)
    skip[:, 2:] = (ext[:, 2:] != blank) & (ext[:, 2:] != ext[:, :-2])
    g = lp.gather(2, ext[:, None, :].expand(B, Tn, -1))                 # [B,T,S]
    alpha = torch.full((B, 2 * Lmax + 1), neg)
    alpha[:, 0] = g[:, 0, 0]
    alpha[:, 1] = torch.where(tl > 0, g[:, 0, 1], torch.full((B,), neg)) if 2 * Lmax + 1 > 1 else alpha[:, 1]
    for t in range(1, Tn):
        a1 = torch.cat([torch.full((B, 1), neg), alpha[:, :-1]], 1)
        a2 = torch.cat([torch.full((B, 2), neg), alpha[:, :-2]], 1)
        a2 = torch.where(skip, a2, torch.full_like(a2, neg))
        msafe = torch.maximum(torch.maximum(alpha, a1), a2).detach()
        alpha = msafe + torch.log(torch.exp(alpha - msafe) + torch.exp(a1 - msafe) + torch.exp(a2 - msafe)) + g[:, t]
        alpha = alpha.clamp(min=neg)
        alpha = torch.where(valid, alpha, torch.full_like(alpha, neg))
    last = alpha.gather(1, (S - 1)[:, None]).squeeze(1)
    prev = torch.where(tl > 0, alpha.gather(1, (S - 2).clamp(min=0)[:, None]).squeeze(1), torch.full((B,), neg))
    msafe = torch.maximum(last, prev).detach()
    nll = -(msafe + torch.log(torch.exp(last - msafe) + torch.exp(prev - msafe)))
    nll = torch.where(nll > 1e29, torch.zeros_like(nll), nll)            # zero_infinity
    if reduction == "mean":
        return (nll / tl.clamp(min=1).float()).mean()
    return nll.sum() if reduction == "sum" else nll


# --------------------------------------------------------------------------- whole model

def model_forward(p: Dict[str, T], cfg: OracleConfig, input_features: T, stno_mask: T, labels: T,
                  upp_labels: Optional[T] = None, enrollments=None, ts=None, emu: bool = False,
                  collect=None, prefix_tokens=()):
    """DiCoWForConditionalGeneration.forward (modeling_dicow.py:248-354) incl. the CTC branch (:326-336).
    Returns dict(loss, logits, encoder_last_hidden_state[, ctc_loss, dec_loss])."""
    enc = encoder_forward(p, cfg, input_features, stno_mask, enrollments, emu, collect)
    dec_in = shift_tokens_right(labels, cfg.pad_token_id, cfg.decoder_start_token_id)
    hdec = decoder_forward(p, cfg, dec_in, enc, emu)
    logits = linear(hdec, p["proj_out.weight"], None, emu)
    if ts is not None:
        loss = soft_loss(logits, labels, upp_labels, ts)
    else:
        loss = hard_loss(logits, labels, upp_labels)
    out = {"loss": loss, "logits": logits, "encoder_last_hidden_state": enc}
    if cfg.ctc_weight > 0.0:
        enc_logits = ctc_logits(p, cfg, enc, emu)
        c = ctc_loss(enc_logits, ctc_prepare_labels(labels, cfg, prefix_tokens), cfg.ctc_loss_reduction)
        out.update(dec_loss=loss, ctc_loss=c, enc_logits=enc_logits)
        out["loss"] = (1 - cfg.ctc_weight) * loss + cfg.ctc_weight * c
    return out


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> T:
    """Whisper's frozen-at-init encoder position table (HF modeling_whisper.sinusoids)."""
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    t = torch.arange(length)[:, None] * inv[None, :]
    return torch.cat([t.sin(), t.cos()], dim=1)


def init_state(cfg: OracleConfig, seed: int = 0, std: float = 0.02, fddt_random: bool = True) -> Dict[str, T]:
    """Deterministic synthetic state dict with the reference's key surface.
    Not the reference's initialiser: values are N(0, std) (LayerNorm weight 1+N, FDDT weights
    near their 'suppressive' values plus noise when ``fddt_random``) so that every parameter
    influences the output in parity tests."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, s=std):
        return torch.randn(*shape, generator=g) * s

    p: Dict[str, T] = {}
    D, F_, M = cfg.d_model, cfg.encoder_ffn_dim, cfg.num_mel_bins
    e = "model.encoder."
    p[e + "conv1.weight"], p[e + "conv1.bias"] = rn(D, M, 3, s=0.05), rn(D)
    p[e + "conv2.weight"], p[e + "conv2.bias"] = rn(D, D, 3, s=0.03), rn(D)
    p[e + "embed_positions.weight"] = sinusoids(cfg.max_source_positions, D)

    def attn(pre, d):
        p[pre + "k_proj.weight"] = rn(d, d, s=d ** -0.5)
        for n in ("q_proj", "v_proj", "out_proj"):
            p[pre + n + ".weight"], p[pre + n + ".bias"] = rn(d, d, s=d ** -0.5), rn(d)

    def ln(pre, d):
        p[pre + "weight"], p[pre + "bias"] = 1.0 + rn(d, s=0.1), rn(d, s=0.1)

    def fd(pre, base):
        for (name, _), b0 in zip(_FDDT_CLASSES, base):
            if cfg.fddt_bias_only:
                p[pre + name] = rn(D, s=0.1)
            elif cfg.fddt_is_diagonal:
                p[pre + name + ".weight"] = b0 + (rn(D, s=0.1) if fddt_random else torch.zeros(D))
                p[pre + name + ".bias"] = rn(D, s=0.1) if fddt_random else torch.zeros(D)
            else:
                p[pre + name + ".weight"] = b0 * torch.eye(D) + rn(D, D, s=0.3 * D ** -0.5)
                p[pre + name + ".bias"] = rn(D, s=0.1)

    for i in range(cfg.encoder_layers):
        pre = f"{e}layers.{i}."
        attn(pre + "self_attn.", D)
        ln(pre + "self_attn_layer_norm.", D)
        p[pre + "fc1.weight"], p[pre + "fc1.bias"] = rn(F_, D, s=D ** -0.5), rn(F_)
        p[pre + "fc2.weight"], p[pre + "fc2.bias"] = rn(D, F_, s=F_ ** -0.5), rn(D)
        ln(pre + "final_layer_norm.", D)
    ln(e + "layer_norm.", D)
    if cfg.use_fddt:
        for i in range(cfg.n_fddts):
            fd(f"{e}fddts.{i}.", (1.0, 1.0, 1.0, 1.0))
        if cfg.use_pre_pos_fddt:
            fd(e + "initial_fddt.", (0.5, 1.0, 0.5, 1.0))
    if cfg.use_enrollments and cfg.scb_layers:
        for i in range(cfg.scb_layers):
            pre = f"{e}ca_enrolls.{i}.cae."
            attn(pre + "cross_attn.", D)
            p[pre + "cross_gate.gate"] = rn(1, s=0.5) + 0.3
            p[pre + "ffn.0.weight"], p[pre + "ffn.0.bias"] = rn(F_, 2 * D, s=(2 * D) ** -0.5), rn(F_)
            p[pre + "ffn.3.weight"], p[pre + "ffn.3.bias"] = rn(D, F_, s=F_ ** -0.5), rn(D)
    d = "model.decoder."
    Dd, Fd = cfg.d_model, cfg.decoder_ffn_dim
    p[d + "embed_tokens.weight"] = rn(cfg.vocab_size, Dd, s=0.05)
    p[d + "embed_positions.weight"] = rn(cfg.max_target_positions, Dd, s=0.02)
    for i in range(cfg.decoder_layers):
        pre = f"{d}layers.{i}."
        attn(pre + "self_attn.", Dd)
        ln(pre + "self_attn_layer_norm.", Dd)
        attn(pre + "encoder_attn.", Dd)
        ln(pre + "encoder_attn_layer_norm.", Dd)
        p[pre + "fc1.weight"], p[pre + "fc1.bias"] = rn(Fd, Dd, s=Dd ** -0.5), rn(Fd)
        p[pre + "fc2.weight"], p[pre + "fc2.bias"] = rn(Dd, Fd, s=Fd ** -0.5), rn(Dd)
        ln(pre + "final_layer_norm.", Dd)
    ln(d + "layer_norm.", Dd)
    p["proj_out.weight"] = p[d + "embed_tokens.weight"]           # tied (modeling_dicow.py:302)
    return p
