"""Decoding side of the path: encoder once, KV-cached greedy decoder, STNO seek windows.

Mirrors, for the in-training evaluation the reference runs every ``eval_steps``:
  * ``stno_seek_windows``   reference DiCoWGenerationMixin.prepare_kwargs_for_generate (src/models/dicow/generation.py
                            :73-106): the STNO mask of each active sample's current 30 s window, padded as silence;
  * ``GreedyDecoder``       the compute HF greedy search drives through the DiCoW model: the STNO-conditioned encoder runs
                            ONCE per window, the cross-attention K/V of every decoder layer are projected ONCE, and each new
                            token costs one decoder step against a per-layer self-attention KV cache; logits go through the
                            SuppressTokens / SuppressTokensAtBegin processors the reference wires (generation.py:286-306),
                            finished rows are padded, decoding stops when every row has produced eos.

  * ``timestamp_rules``     WhisperTimeStampLogitsProcessorCustom (src/models/dicow/utils.py:5-14) as one kernel per step;
  * joint CTC / attention   ``ctc=dict(...)`` switches on ctc_decoding.CtcRescorer (decoding.py) after a log-softmax, as the
                            reference wires it for greedy search (generation.py:249-268).

Every matrix product, attention and LayerNorm is a C-ABI kernel call (the training step's kernels; single-row queries go
through dicow_attn_fwd with Lq = 1, the <= 16-row products through the weight-streaming gemm_nt_skinny_kernel).  Not here:
beam search, temperature fallback and the long-form seek loop of HF's generate -- host control flow around this step
function.  No CPU fallback.
"""
from types import SimpleNamespace as NS

import torch

from . import _lib as L
from . import ops
from .engine import heads, linear_fwd

F32, BF16 = torch.float32, torch.bfloat16


def stno_seek_windows(stno_mask, seek, max_frames, batch_idx_map, num_frames=1500):
    """stno_mask [B_all, 4, T_total] (any device); seek / max_frames: feature-frame positions per ORIGINAL sample;
    batch_idx_map: original index of each still-active sample.  Returns [len(map), 4, num_frames] on stno_mask's device."""
    out = stno_mask.new_zeros((len(batch_idx_map), stno_mask.shape[1], num_frames))
    out[:, 0, :] = 1.0                                                # padding frames are silence
    for i, prev in enumerate(int(b) for b in batch_idx_map):
        s = int(seek[prev]) // 2
        n = max(0, min(int(max_frames[prev]) // 2 - s, num_frames))
        out[i, :, :n] = stno_mask[prev, :, s:s + n]
    return out


def timestamp_rules(input_ids, scores, begin_index, eos_token_id, no_timestamps_token_id, max_initial_timestamp_index=None,
                    detect_from_logprob=True):
    """WhisperTimeStampLogitsProcessorCustom.__call__ (reference src/models/dicow/utils.py:5-14): scores fp32 [B, V] on the
    GPU are constrained IN PLACE (and returned); input_ids int64 [B, L] = prompt (begin_index tokens) + generated tokens."""
    if not scores.is_cuda or scores.dtype != F32 or scores.stride(1) != 1:
        raise L.DicowError("timestamp_rules: scores must be fp32 rows on the GPU (no CPU fallback)")
    ids = input_ids.to(device=scores.device, dtype=torch.int64).contiguous()
    B, V = scores.shape
    L.call("dicow_whisper_timestamp_rules", scores.data_ptr(), scores.stride(0), B, V, ids.data_ptr(), ids.shape[1], begin_index,
           no_timestamps_token_id + 1, eos_token_id, no_timestamps_token_id,
           -1 if max_initial_timestamp_index is None else max_initial_timestamp_index, int(detect_from_logprob), L.stream())
    return scores


class GreedyDecoder:
    """``GreedyDecoder(model).generate(...)`` for a ``DiCoWForConditionalGeneration`` on the GPU."""

    def __init__(self, model, use_graphs=False):
        """use_graphs: capture each decoder position's step (~55 launches) into a hipGraph the first time it runs and replay
        it afterwards (the step is launch-bound at B = 16: 0.70 -> 0.30 ms); the KV caches and the token buffer are
        persistent per batch size so that the captured pointers stay valid across windows.  Weights must not be re-allocated
        between calls (evaluation)."""
        self.model, self.cfg = model, model.config
        self.use_graphs = use_graphs
        self._persist = {}                       # batch size -> decoding state with static buffers and captured graphs

    def _step_graphed(self, ids, t, st):
        st.ids_in.copy_(ids)
        if t not in st.graphs:
            if not st.warm:                      # first-use initialisation inside the kernels' host wrappers happens eagerly
                self._step(st.ids_in, t, st)
                st.warm = True
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=st.pool):
                out = self._step(st.ids_in, t, st)
            st.graphs[t] = (g, out)
        g, out = st.graphs[t]
        g.replay()
        return out

    # ---- one decoder step for position t over the caches
    def _step(self, ids, t, st):
        model, cfg, W = self.model, self.cfg, st.W
        dec = model.model.decoder
        B, D, H = st.B, cfg.d_model, cfg.decoder_attention_heads
        dev = ids.device
        h = torch.empty(B, D, dtype=F32, device=dev)
        ops.embed_fwd(ids.view(B, 1).contiguous(), dec.embed_tokens.weight.detach(), dec.embed_positions.weight.detach()[t:], h)
        x = torch.empty(B, D, dtype=BF16, device=dev)
        o = torch.empty(B, D, dtype=BF16, device=dev)

        def ln(src, mod):
            ops.fddt_ln_fwd(src, B, D, ln_w=mod.weight.detach(), ln_b=mod.bias.detach(), y_bf16=x)
            return x

        for i, lyr in enumerate(dec.layers):
            w, c = W.layers[i], st.layers[i]
            qkv = linear_fwd(ln(h, lyr.self_attn_layer_norm), w.sa.qkv, B, flags=L.EPI_SCALE_N, scale=0.125, scale_ncols=D)
            c.k[:, t].copy_(qkv[:, D:2 * D])
            c.v[:, t].copy_(qkv[:, 2 * D:])
            ops.attn_fwd(heads(qkv[:, :D], B, 1, H), c.k[:, :t + 1].view(B, t + 1, H, 64), c.v[:, :t + 1].view(B, t + 1, H, 64),
                         heads(o, B, 1, H))
            h = linear_fwd(o, w.sa.o, B, out_dtype=F32, residual=h)
            q = linear_fwd(ln(h, lyr.encoder_attn_layer_norm), w.ca.q, B, flags=L.EPI_SCALE_N, scale=0.125, scale_ncols=D)
            ops.attn_fwd(heads(q, B, 1, H), heads(c.ckv[:, :D], B, st.T, H), heads(c.ckv[:, D:], B, st.T, H), heads(o, B, 1, H))
            h = linear_fwd(o, w.ca.o, B, out_dtype=F32, residual=h)
            a = linear_fwd(ln(h, lyr.final_layer_norm), w.fc1, B, flags=L.EPI_GELU)
            h = linear_fwd(a, w.fc2, B, out_dtype=F32, residual=h)
        logits = torch.empty(B, W.vpad, dtype=F32, device=dev)
        ops.gemm_nt(ln(h, dec.layer_norm), W.head.w, logits, B, W.vpad, D)
        return logits[:, :cfg.vocab_size]

    @torch.no_grad()
    def encode(self, input_features, stno_mask, enrollments=None):
        """Encoder once + the cross-attention K/V of every decoder layer once.  Returns the decoding state."""
        model, cfg = self.model, self.cfg
        if not input_features.is_cuda:
            raise L.DicowError("GreedyDecoder: tensors must be on the GPU (no CPU fallback)")
        enc_out = model.model.encoder(input_features, stno_mask=stno_mask, enrollments=enrollments).last_hidden_state
        B, T, D = enc_out.shape
        W = model._engine().W
        enc_bf = ops.cast_bf16(enc_out.contiguous().to(F32)).view(B * T, D)
        Lmax = cfg.max_target_positions
        dev = enc_out.device
        st = self._persist.get(B) if self.use_graphs else None
        if st is None or st.W is not W or st.T != T:
            st = NS(B=B, T=T, W=W, layers=[], graphs={}, warm=False, pool=None, ids_in=torch.zeros(B, dtype=torch.long, device=dev))
            for w in W.layers:
                st.layers.append(NS(ckv=torch.empty(B * T, w.ca.kv.N, dtype=BF16, device=dev),
                                    k=torch.empty(B, Lmax, D, dtype=BF16, device=dev),
                                    v=torch.empty(B, Lmax, D, dtype=BF16, device=dev)))
            if self.use_graphs:
                st.pool = torch.cuda.graph_pool_handle()
                self._persist[B] = st
        st.enc_out = enc_out
        for w, c in zip(W.layers, st.layers):
            linear_fwd(enc_bf, w.ca.kv, B * T, out=c.ckv)
        return st

    @torch.no_grad()
    def generate(self, input_features, stno_mask, decoder_input_ids, max_new_tokens, eos_token_id=None, pad_token_id=None,
                 suppress_tokens=None, begin_suppress_tokens=None, enrollments=None, return_scores=False, ctc=None,
                 timestamps=None):
        """decoder_input_ids int64 [B, P]: the forced prefix (start token, language / task / timestamp tokens).
        ctc: dict(weight=..., first_timestamp=..., upper_cased=[(lo, up), ...], prefix_len=..., n_score=500) switches on the
        joint CTC / attention scoring of generation.py:249-268 (log-softmax, then ctc_decoding.CtcRescorer on the model's CTC head).
        timestamps: dict(no_timestamps_token_id=..., max_initial_timestamp_index=...) applies Whisper's timestamp rules
        (return_timestamps=True in the reference, generation.py:272-281), after the suppress lists and before the CTC term.
        Returns sequences [B, P + n] (and the processed fp32 scores [n, B, V] of the generated positions)."""
        cfg = self.cfg
        eos = cfg.eos_token_id if eos_token_id is None else eos_token_id
        pad = cfg.pad_token_id if pad_token_id is None else pad_token_id
        st = self.encode(input_features, stno_mask, enrollments)
        dev = st.enc_out.device
        ids = decoder_input_ids.to(dev)
        B, P = ids.shape
        if P < 1 or P + max_new_tokens > cfg.max_target_positions:
            raise ValueError(f"prompt {P} + max_new_tokens {max_new_tokens} exceeds max_target_positions {cfg.max_target_positions}")
        sup = None if not suppress_tokens else torch.as_tensor(list(suppress_tokens), device=dev)
        bsup = None if not begin_suppress_tokens else torch.as_tensor(list(begin_suppress_tokens), device=dev)
        rescorer = None
        if ctc is not None and ctc.get("weight", 0.0) > 0.0:
            from .ctc_decoding import CtcRescorer
            enc_logits = self.model.get_enc_logits(st.enc_out)
            rescorer = CtcRescorer(enc_logits, cfg.vocab_size, eos, ids[0, 0].item(), ctc["first_timestamp"], ctc.get("upper_cased", ()),
                                   ctc.get("prefix_len", P), ctc["weight"], ctc.get("n_score", 500))
            rows = torch.arange(B, device=dev)
        step = self._step_graphed if self.use_graphs else self._step
        for t in range(P - 1):                                       # prefill the caches with the prefix
            step(ids[:, t], t, st)
        unfinished = torch.ones(B, dtype=torch.bool, device=dev)
        seq, scores = [ids], []
        cur = ids[:, P - 1]
        for n in range(max_new_tokens):
            logits = step(cur, P - 1 + n, st)
            if sup is not None:
                logits[:, sup] = -float("inf")
            if bsup is not None and n == 0:
                logits[:, bsup] = -float("inf")
            if timestamps is not None:
                logits = timestamp_rules(torch.cat(seq, dim=1), logits, P, eos, timestamps["no_timestamps_token_id"],
                                         timestamps.get("max_initial_timestamp_index"))
            if rescorer is not None:
                logits = rescorer(torch.cat(seq, dim=1), torch.log_softmax(logits, dim=-1))
            if return_scores:
                scores.append(logits.clone())
            nxt = logits.argmax(-1)
            nxt = torch.where(unfinished, nxt, torch.full_like(nxt, pad))
            if rescorer is not None:
                rescorer.update_state(nxt, rows)
            seq.append(nxt[:, None])
            unfinished = unfinished & (nxt != eos)
            cur = nxt
            if not bool(unfinished.any()):
                break
        out = torch.cat(seq, dim=1)
        return (out, torch.stack(scores)) if return_scores else out
