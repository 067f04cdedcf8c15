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
    def encode(self, input_features, stno_mask, enrollments=None, num_beams=1):
        """Encoder once + the cross-attention K/V of every decoder layer once.  Returns the decoding state.
        num_beams > 1: the decoding state has batch x num_beams rows (the encoder and the K/V projections still run once per
        window; their result is repeated per beam)."""
        model, cfg = self.model, self.cfg
        if not input_features.is_cuda:
            raise L.DicowError("GreedyDecoder: tensors must be on the GPU (no CPU fallback)")
        enc_out = model.model.encoder(input_features, stno_mask=stno_mask, enrollments=enrollments).last_hidden_state
        B0, T, D = enc_out.shape
        B = B0 * num_beams
        W = model._engine().W
        enc_bf = ops.cast_bf16(enc_out.contiguous().to(F32)).view(B0 * T, D)
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
            if num_beams == 1:
                linear_fwd(enc_bf, w.ca.kv, B * T, out=c.ckv)
            else:
                kv = linear_fwd(enc_bf, w.ca.kv, B0 * T)
                c.ckv.view(B0, num_beams, T, -1).copy_(kv.view(B0, 1, T, -1).expand(B0, num_beams, T, kv.shape[1]))
        return st

    @torch.no_grad()
    def detect_language(self, input_features, stno_mask, lang_token_ids, decoder_start_token_id=None, enrollments=None,
                        return_logits=False):
        """DiCoWGenerationMixin.detect_language (reference src/models/dicow/generation.py:151-221): the encoder runs on the
        first window with its STNO mask (and enrollments), the decoder runs position 0 on the start token, every logit that is
        not a language token (``generation_config.lang_to_id.values()``) is masked and the argmax is returned ([B] int64)."""
        cfg = self.cfg
        n = 2 * cfg.max_source_positions
        st = self.encode(input_features[:, :, :n].contiguous(), stno_mask[:, :, :n // 2].contiguous(), enrollments)
        start = cfg.decoder_start_token_id if decoder_start_token_id is None else decoder_start_token_id
        ids = torch.full((st.B,), start, dtype=torch.long, device=input_features.device)
        logits = self._step(ids, 0, st).float()
        lang = torch.as_tensor(list(lang_token_ids), dtype=torch.long, device=logits.device)
        masked = torch.full_like(logits, float("-inf"))
        masked[:, lang] = logits[:, lang]
        best = masked.argmax(-1)
        return (best, logits) if return_logits else best

    @torch.no_grad()
    def beam_search(self, input_features, stno_mask, decoder_input_ids, max_length, num_beams, eos_token_id=None, pad_token_id=None,
                    length_penalty=1.0, early_stopping=False, suppress_tokens=None, begin_suppress_tokens=None, enrollments=None,
                    timestamps=None, ctc=None):
        """Beam search as the reference runs it (DiCoWGenerationMixin._beam_search, generation.py:815-1153, on transformers'
        vectorised beam helpers): per step the top 2K continuations over beams x vocabulary, the K best unfinished keep running,
        finished ones compete for the K result slots with length-penalised scores; processors act on log-probabilities
        (suppress lists, timestamp rules, joint CTC term without a second log-softmax, generation.py:249-268); KV caches and
        the CTC states follow ``beam_idx``.  Returns (sequences [B, <= max_length], scores [B]) of the best hypothesis."""
        cfg, K = self.cfg, int(num_beams)
        eos = cfg.eos_token_id if eos_token_id is None else eos_token_id
        pad = cfg.pad_token_id if pad_token_id is None else pad_token_id
        if self.use_graphs:
            raise L.DicowError("beam_search reorders the KV caches every step: use a decoder without graph capture")
        st = self.encode(input_features, stno_mask, enrollments, num_beams=K)
        dev = st.enc_out.device
        prompt = decoder_input_ids.to(dev)
        B, P = prompt.shape
        V = cfg.vocab_size
        max_length = min(int(max_length), cfg.max_target_positions)
        if P < 1 or P >= max_length:
            raise ValueError(f"prompt length {P} leaves no room below max_length {max_length}")
        sup = None if not suppress_tokens else torch.as_tensor(list(suppress_tokens), device=dev)
        bsup = None if not begin_suppress_tokens else torch.as_tensor(list(begin_suppress_tokens), device=dev)
        rescorer = None
        if ctc is not None and ctc.get("weight", 0.0) > 0.0:
            from .ctc_decoding import CtcRescorer
            rescorer = CtcRescorer(self.model.get_enc_logits(st.enc_out), V, eos, prompt[0, 0].item(), ctc["first_timestamp"],
                                   ctc.get("upper_cased", ()), ctc.get("prefix_len", P), ctc["weight"], ctc.get("n_score", 500), num_beams=K)
        fill = pad if pad is not None else eos
        run_seq = torch.full((B, K, max_length), fill, dtype=torch.long, device=dev)
        run_seq[:, :, :P] = prompt[:, None, :]
        seqs = run_seq.clone()
        run_sc = torch.zeros(B, K, dtype=F32, device=dev)
        run_sc[:, 1:] = -1e9
        fin_sc = torch.full((B, K), -1e9, dtype=F32, device=dev)
        is_fin = torch.zeros(B, K, dtype=torch.bool, device=dev)
        unsat = torch.ones(B, 1, dtype=torch.bool, device=dev)
        run_idx = torch.full((B, K, max_length - P), -1, dtype=torch.int32, device=dev)
        fin_idx = run_idx.clone()
        top_mask = (torch.arange(2 * K, device=dev) < K)[None, :]
        offs = (torch.arange(B, device=dev) * K)[:, None]

        def gather(t, idx):
            while idx.dim() < t.dim():
                idx = idx.unsqueeze(-1)
            return torch.take_along_dim(t, idx, dim=1)

        for t in range(P - 1):                                       # prefill: every beam of a row starts from the same prompt
            self._step(run_seq[:, :, t].reshape(-1), t, st)
        cur = P
        while True:
            flat = run_seq[:, :, :cur].reshape(B * K, cur)
            logp = torch.log_softmax(self._step(flat[:, -1].contiguous(), cur - 1, st).float(), dim=-1)
            if sup is not None:
                logp[:, sup] = -float("inf")
            if bsup is not None and cur == P:
                logp[:, bsup] = -float("inf")
            if timestamps is not None:
                logp = timestamp_rules(flat, logp, P, eos, timestamps["no_timestamps_token_id"], timestamps.get("max_initial_timestamp_index"))
            if rescorer is not None:
                logp = rescorer(flat, logp)
            acc = (logp.view(B, K, V) + run_sc[:, :, None]).reshape(B, K * V)
            tk_lp, tk = torch.topk(acc, k=2 * K)
            src = tk // V
            tk_seq, tk_idx = gather(run_seq, src).clone(), gather(run_idx, src).clone()
            tk_seq[:, :, cur] = tk % V
            tk_idx[:, :, cur - P] = (src + offs).to(torch.int32)
            hits = (tk_seq[:, :, cur] == eos) | (cur + 1 >= max_length)
            run_lp = tk_lp + hits.float() * -1e9
            nxt = torch.topk(run_lp, k=K)[1]
            run_seq, run_sc, run_idx = gather(tk_seq, nxt), gather(run_lp, nxt), gather(tk_idx, nxt)
            just = hits & top_mask
            fin = tk_lp / float((cur + 1 - P) ** length_penalty)
            fin = fin + (is_fin.all(dim=-1, keepdim=True) & (early_stopping is True)).float() * -1e9
            fin = fin + (~unsat).float() * -1e9 + (~just).float() * -1e9
            m_sc = torch.cat((fin_sc, fin), dim=1)
            top = torch.topk(m_sc, k=K)[1]
            seqs, fin_sc = gather(torch.cat((seqs, tk_seq), dim=1), top), gather(m_sc, top)
            fin_idx, is_fin = gather(torch.cat((fin_idx, tk_idx), dim=1), top), gather(torch.cat((is_fin, just), dim=1), top)
            beam_idx = run_idx[..., cur - P].reshape(-1).long()
            for c in st.layers:                                      # the caches follow their beams
                c.k[:, :cur].copy_(c.k[:, :cur].index_select(0, beam_idx))
                c.v[:, :cur].copy_(c.v[:, :cur].index_select(0, beam_idx))
            if rescorer is not None:
                rescorer.update_state(run_seq.reshape(B * K, -1)[:, cur], beam_idx)
            cur += 1
            hyp_len = (max_length - P) if (early_stopping == "never" and length_penalty > 0.0) else (cur - P)
            best = run_sc[:, :1] / float(hyp_len ** length_penalty)
            worst = torch.where(is_fin, fin_sc.min(dim=1, keepdim=True)[0], torch.full_like(fin_sc, -1e9))
            unsat = unsat & (best > worst).any(dim=-1, keepdim=True)
            go_on = bool(unsat.any()) and not (bool(is_fin.all()) and early_stopping is True) and not bool(hits.all())
            if not go_on:
                break
        gen = int(((fin_idx[:, :1] + 1) != 0).sum(dim=2).max())
        return seqs[:, 0, :P + gen], fin_sc[:, 0]

    @torch.no_grad()
    def generate(self, input_features, stno_mask, decoder_input_ids, max_new_tokens, eos_token_id=None, pad_token_id=None,
                 suppress_tokens=None, begin_suppress_tokens=None, enrollments=None, return_scores=False, ctc=None,
                 timestamps=None, temperature=0.0, generator=None, no_speech_token_id=None):
        """decoder_input_ids int64 [B, P]: the forced prefix (start token, language / task / timestamp tokens).
        ctc: dict(weight=..., first_timestamp=..., upper_cased=[(lo, up), ...], prefix_len=..., n_score=500) switches on the
        joint CTC / attention scoring of generation.py:249-268 (log-softmax, then ctc_decoding.CtcRescorer on the model's CTC head).
        timestamps: dict(no_timestamps_token_id=..., max_initial_timestamp_index=...) applies Whisper's timestamp rules
        (return_timestamps=True in the reference, generation.py:272-281), after the suppress lists and before the CTC term.
        temperature > 0: multinomial sampling from softmax(processed scores / temperature) (HF's sample mode, what Whisper's
        temperature fallback decodes with; `generator` makes it reproducible); the returned scores are then the temperature-scaled
        ones, as HF's are.  no_speech_token_id: also keep softmax(logits of the position after the start token)[that id] per row
        in ``self.no_speech_prob`` (HF WhisperNoSpeechDetection).
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
        self.no_speech_prob = None
        for t in range(P - 1):                                       # prefill the caches with the prefix
            lg = step(ids[:, t], t, st)
            if t == 0 and no_speech_token_id is not None:
                self.no_speech_prob = torch.softmax(lg.float(), dim=-1)[:, no_speech_token_id].clone()
        unfinished = torch.ones(B, dtype=torch.bool, device=dev)
        seq, scores = [ids], []
        cur = ids[:, P - 1]
        for n in range(max_new_tokens):
            logits = step(cur, P - 1 + n, st)
            if n == 0 and P == 1 and no_speech_token_id is not None:
                self.no_speech_prob = torch.softmax(logits.float(), dim=-1)[:, no_speech_token_id].clone()
            if sup is not None:
                logits[:, sup] = -float("inf")
            if bsup is not None and n == 0:
                logits[:, bsup] = -float("inf")
            if timestamps is not None:
                logits = timestamp_rules(torch.cat(seq, dim=1), logits, P, eos, timestamps["no_timestamps_token_id"],
                                         timestamps.get("max_initial_timestamp_index"))
            if rescorer is not None:
                logits = rescorer(torch.cat(seq, dim=1), torch.log_softmax(logits, dim=-1))
            if temperature and temperature > 0.0:
                logits = logits / temperature
            if return_scores:
                scores.append(logits.clone())
            if temperature and temperature > 0.0:
                nxt = torch.multinomial(torch.softmax(logits.float(), dim=-1), 1, generator=generator)[:, 0]
            else:
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

    @torch.no_grad()
    def generate_with_fallback(self, input_features, stno_mask, decoder_input_ids, max_new_tokens, temperatures=(0.0, 0.2, 0.4, 0.6, 0.8, 1.0),
                               compression_ratio_threshold=1.35, logprob_threshold=-1.0, no_speech_threshold=None,
                               no_speech_token_id=None, generator=None, enrollments=None, skip_by_row=False, **gen_kw):
        """Whisper's temperature fallback over one batch of windows (reference generation.py:567-611 -> transformers'
        generate_with_fallback): greedy first; windows whose output is too repetitive (zlib compression ratio) or too unlikely
        (average log-probability) are decoded again -- encoder included, as in HF -- with sampling at the next temperature; a
        window that is unlikely AND looks like silence (no_speech_prob) is skipped instead.  gen_kw: generate()'s processor
        arguments (eos / pad ids, suppress lists, timestamps, ctc).
        Returns (token lists of the generated positions without the eos, should_skip flags, temperature index per window)."""
        cfg = self.cfg
        eos = gen_kw.get("eos_token_id", None)
        eos = cfg.eos_token_id if eos is None else eos
        pad = gen_kw.get("pad_token_id", None)
        pad = cfg.pad_token_id if pad is None else pad
        B, P = decoder_input_ids.shape
        if no_speech_threshold is not None and no_speech_token_id is None:
            raise ValueError("no_speech_threshold needs no_speech_token_id")

        def decode(rows, temperature):
            idx = torch.as_tensor(rows, device=input_features.device)
            enr = None if enrollments is None else {k: v[idx] for k, v in enrollments.items()}
            seqs, scores = self.generate(input_features[idx], stno_mask[idx], decoder_input_ids[idx.to(decoder_input_ids.device)],
                                         max_new_tokens, enrollments=enr, return_scores=True, temperature=temperature,
                                         generator=generator, no_speech_token_id=no_speech_token_id, **gen_kw)
            toks = seqs[:, P:].tolist()
            return toks, [scores[:, i] for i in range(len(rows))], self.no_speech_prob

        return decode_with_fallback(decode, B, tuple(temperatures), cfg.vocab_size, pad, eos, compression_ratio_threshold,
                                    logprob_threshold, no_speech_threshold, skip_by_row=skip_by_row)


# ------------------------------------------------------------------------------------------------ temperature fallback
# The reference's generate_with_fallback (src/models/dicow/generation.py:567-611) cuts the STNO masks to the active windows and
# hands over to transformers' WhisperGenerationMixin.generate_with_fallback; these are that method's decisions (golden F18 is
# taken from the transformers code itself, driven with scripted decoder outputs).
def token_compression_ratio(tokens, vocab_size):
    """Raw bytes / zlib-compressed bytes of the token ids (HF _retrieve_compression_ratio): repetition detector."""
    import math
    import zlib
    width = int(math.log2(vocab_size) / 8) + 1
    raw = b"".join(int(t).to_bytes(width, "little") for t in tokens)
    return len(raw) / len(zlib.compress(raw))


def sequence_avg_logprob(scores, tokens, temperature):
    """Mean log-probability of the generated tokens, eos included (HF _retrieve_avg_logprobs).  scores [n, V]: the processed
    scores of the generated positions as generate(return_scores=True) returns them, i.e. divided by the temperature when
    sampling -- undone here before the fp32 log-softmax."""
    tok = torch.as_tensor(list(tokens), dtype=torch.long, device=scores.device)
    n = min(scores.shape[0], tok.numel())
    sc, tk = scores[:n], (tok if scores.shape[0] > tok.numel() else tok[-n:])[:n]
    if temperature and temperature > 0.0:
        sc = sc * temperature
    lp = torch.log_softmax(sc.float(), dim=-1).to(scores.dtype)
    return float(lp.gather(1, tk[:, None]).sum() / tok.numel())


def window_needs_fallback(tokens, scores, temperature, vocab_size, compression_ratio_threshold, logprob_threshold,
                          no_speech_threshold=None, no_speech_prob=None):
    """(needs_fallback, should_skip) of one decoded window (HF _need_fallback)."""
    needs = compression_ratio_threshold is not None and token_compression_ratio(tokens, vocab_size) > compression_ratio_threshold
    low = False
    if logprob_threshold is not None:
        low = sequence_avg_logprob(scores, tokens, temperature) < logprob_threshold
        needs = needs or low
    if no_speech_threshold is not None and low and no_speech_prob is not None and no_speech_prob > no_speech_threshold:
        return False, True
    return needs, False


def decode_with_fallback(decode, n_windows, temperatures, vocab_size, pad_token_id, eos_token_id, compression_ratio_threshold=1.35,
                         logprob_threshold=-1.0, no_speech_threshold=None, skip_by_row=False):
    """Temperature ladder over a batch of windows.  decode(rows, temperature) -> (token lists of the generated positions,
    [n, V] score tensors, no-speech probabilities or None) for the listed windows.  Every window keeps its latest result;
    those that fail the compression / log-probability test are decoded again at the next temperature.
    Returns (token lists without the eos, should_skip flags, index of the temperature each window ended with).

    skip_by_row=False reproduces transformers' `generate_with_fallback` (which the reference delegates to, generation.py:567-611)
    to the letter, INCLUDING its indexing quirk: `should_skip` is written at the window's position in the CURRENT fallback
    sub-batch, so after the first fallback round a silent window's flag lands on another window of the batch (golden F18 pins
    that behaviour).  skip_by_row=True is the corrected form: the flag is kept per original window and cleared when the
    window is decoded again."""
    final, used, skip = [None] * n_windows, [None] * n_windows, [False] * n_windows
    rows = list(range(n_windows))
    for k, temp in enumerate(temperatures):
        toks, scores, nsp = decode(list(rows), temp)
        again = []
        for i, row in enumerate(rows):
            seq = list(toks[i])
            if seq and seq[-1] == pad_token_id:       # drop the padding tail; with pad == eos one eos stays (it is scored)
                npad = sum(1 for t in seq if t == pad_token_id) - (1 if pad_token_id == eos_token_id else 0)
                if npad:
                    seq = seq[:-npad]
            needs, sk = window_needs_fallback(seq, scores[i], temp, vocab_size, compression_ratio_threshold, logprob_threshold,
                                              no_speech_threshold, None if nsp is None else float(nsp[i]))
            skip[row if skip_by_row else i] = sk      # (transformers indexes this list by the position in the CURRENT batch)
            final[row], used[row] = (seq[:-1] if seq and seq[-1] == eos_token_id else seq), k
            if needs:
                again.append(row)
        rows = again
        if not rows or k == len(temperatures) - 1:
            break
    return final, skip, used


# ------------------------------------------------------------------------------------------------ long-form decoding
def retrieve_segment(seq, time_offset, timestamp_begin, seek_num_frames, time_precision=0.02, input_stride=2):
    """DiCoWGenerationMixin._retrieve_segment (reference src/models/dicow/generation.py:416-534) on a plain list of the tokens
    decoded for one window (prompt removed): consecutive timestamp pairs close segments; a single trailing timestamp means
    "no more speech in this window" (the whole window is consumed), otherwise the seek pointer moves to the last closed
    segment; without any pair the window becomes one segment (or is rolled back when its only timestamp lies beyond 4 s from
    the start).  Returns ([dict(start, end, tokens)], frames to advance); host control flow, pinned by golden F16."""
    is_ts = [t >= timestamp_begin for t in seq]
    single_ending = is_ts[-2:] == [False, True]
    pairs = [i + 1 for i in range(len(seq) - 1) if is_ts[i] and is_ts[i + 1]]
    segments = []
    if pairs:
        slices = list(pairs)
        if single_ending:
            slices.append(len(seq))
        else:
            slices[-1] += 1
        last = 0
        for i, cur in enumerate(slices):
            tok = seq[last:cur]
            end_tok = tok[-1 if (i < len(slices) - 1 or single_ending) else -2]
            segments.append({"start": time_offset + (tok[0] - timestamp_begin) * time_precision,
                             "end": time_offset + (end_tok - timestamp_begin) * time_precision, "tokens": tok})
            last = cur
        offset = seek_num_frames if single_ending else (seq[last - 2] - timestamp_begin) * input_stride
    else:
        stamps = [t for t, f in zip(seq, is_ts) if f]
        start_pos, last_pos, skip, offset = 0.0, seek_num_frames // 2, False, seek_num_frames
        if len(stamps) > 1:
            start_pos, last_pos = stamps[-2] - timestamp_begin, stamps[-1] - timestamp_begin
        elif len(stamps) == 1:
            start_pos = stamps[-1] - timestamp_begin
            if start_pos > 200:
                offset, skip = start_pos * input_stride - 100, True
        elif len(seq) <= 1:
            skip = True
        if not skip:
            segments = [{"start": time_offset + start_pos * time_precision, "end": time_offset + last_pos * time_precision,
                         "tokens": list(seq)}]
            offset = seek_num_frames
    if offset <= 0:
        raise ValueError(f"Segment offset: {offset} <= 0. This should not happen!")
    return segments, int(offset)


_UNITS_PER_TICK = 100                  # integer time base of fix_timestamps_from_segmentation: 0.02 s = 100 units
_WINDOW_UNITS = 1500 * _UNITS_PER_TICK   # one 30 s window
_CARRY = -(_UNITS_PER_TICK + 1)         # the reference's Decimal(-0.02): one tick and a hair (4e-19 s) below zero


def _seconds_to_ticks(x):
    """Seconds -> 0.02 s ticks, half-up on the shortest decimal form of the float (what Decimal(str(x)) / ROUND_HALF_UP of
    reference generation.py:314-320 computes), in exact rational arithmetic."""
    from fractions import Fraction
    return int((Fraction(str(float(x))) * 50 + Fraction(1, 2)) // 1)


def fold_segments_to_windows(segments, first_timestamp_token, filler_token):
    """One recording's segments [dict(start, end, tokens)] (recording time, seconds) -> [(start, tokens, end)] with times in
    window-relative 0.02 s ticks (0..1500) or None for a stamp the reference prints as '<|-0.00|>' (not a timestamp token).
    Behaviour of DiCoWGenerationMixin._fix_timestamps_from_segmentation (reference generation.py:322-400) in integer
    arithmetic: a (30, filler, 30) entry is inserted when a 30 s block boundary is crossed, (0, filler, 30) entries bridge
    skipped blocks, a segment that is exactly 30 s long and would wrap is shortened by one tick and that tick is carried into
    the following segments.  The reference's carry is the inexact Decimal(-0.02) (a hair more than one tick); it decides block
    membership of times that land exactly on a boundary, so it is modelled as one extra unit at 100 units per tick.  Pinned by
    golden F17 (247 recordings run through the reference method)."""
    U, W = _UNITS_PER_TICK, _WINDOW_UNITS
    filler = [filler_token]
    out, prev_end, carry = [], None, 0

    def tick(u):
        return None if u < 0 else (u + U // 2) // U

    for seg in segments:
        toks = [int(t) for t in seg["tokens"]]
        if not toks or toks == [first_timestamp_token]:
            continue
        t0, t1 = _seconds_to_ticks(seg["start"]), _seconds_to_ticks(seg["end"])
        a, b = t0 * U + carry, t1 * U + carry
        if a < 0:
            raise ValueError("fold_segments_to_windows: negative segment start")
        if prev_end is None:
            out += [(0, filler, 1500)] * (t0 // 1500)
        else:
            here, before = a // W, max(prev_end - U // 20, 0) // W          # 1 ms below the previous end (generation.py:362)
            if here > before:
                out.append((1500, filler, 1500))
            out += [(0, filler, 1500)] * max(here - before - 1, 0)
        if a // W == b // W:
            out.append((tick(a % W), toks, tick(b % W)))
        elif b % W == 0:
            out.append((tick(a % W), toks, 1500))
            carry = 0
        else:
            lo, hi = a % W, b % W
            if t1 - t0 == 1500:
                if lo == 0 or lo >= W - 2:                                   # float(lo) % 30.0 == 0.0 in the reference
                    hi, carry = W, 0
                else:
                    carry = _CARRY
                    hi += carry
            else:
                carry = 0
            out.append((tick(lo), toks, tick(hi)))
        prev_end = t1 * U + carry
    return out


def fix_timestamps_from_segmentation(segments, first_timestamp_token, filler_token, pad_token_id, prefix_ids=(), suffix_ids=(),
                                     device=None):
    """segments: per recording, the [dict(start, end, tokens)] list LongFormDecoder.transcribe returns.  Returns a padded
    LongTensor [recordings, L] of ``prefix <|start|> text <|end|> ... suffix`` with window-relative timestamp tokens
    (reference generation.py:322-415).  The reference builds the text '<|s|>' + decode(tokens) + '<|e|>' and re-encodes it with the
    tokenizer (which adds its prefix / eos: pass them as prefix_ids / suffix_ids); here the ids are written directly --
    timestamp ids inside ``tokens`` are dropped exactly as Whisper's decode() drops them, text ids are kept as they are."""
    rows = []
    for rec in segments:
        ids = list(prefix_ids)
        for s, toks, e in fold_segments_to_windows(rec, first_timestamp_token, filler_token):
            if s is not None:
                ids.append(first_timestamp_token + s)
            ids += [t for t in toks if t < first_timestamp_token]
            if e is not None:
                ids.append(first_timestamp_token + e)
        rows.append(ids + list(suffix_ids))
    L_ = max((len(r) for r in rows), default=0)
    out = torch.full((len(rows), L_), pad_token_id, dtype=torch.long, device=device)
    for i, r in enumerate(rows):
        out[i, :len(r)] = torch.tensor(r, dtype=torch.long)
    return out


class LongFormDecoder:
    """Sequential long-form decoding of recordings longer than one window, the way the reference drives HF's Whisper
    ``generate`` (temperature 0, no conditioning on previous text -- the reference raises for it, generation.py:545): every
    still-active recording contributes its current 30 s window (features from ``seek``, zero-padded; STNO mask through
    ``stno_seek_windows``), the windows are decoded as one batch (greedy or beam, timestamp rules on), each window's tokens go
    through ``retrieve_segment`` and the recording's seek pointer advances by the returned offset."""

    def __init__(self, model, num_beams=1):
        self.model, self.cfg, self.num_beams = model, model.config, num_beams
        self.decoder = GreedyDecoder(model)

    @torch.no_grad()
    def transcribe(self, input_features, stno_mask, max_frames, decoder_input_ids, no_timestamps_token_id, eos_token_id=None,
                   pad_token_id=None, max_new_tokens=None, enrollments=None, **gen_kw):
        """input_features [B, M, T_total] (T_total a multiple of nothing in particular), stno_mask [B, 4, T_total / 2],
        max_frames [B] valid feature frames per recording, decoder_input_ids [1 or B, P] the forced prompt.
        Returns per recording a list of segments dict(start, end, tokens) in seconds / token ids."""
        cfg = self.cfg
        eos = cfg.eos_token_id if eos_token_id is None else eos_token_id
        pad = cfg.pad_token_id if pad_token_id is None else pad_token_id
        ts0 = no_timestamps_token_id + 1
        W = 2 * cfg.max_source_positions                           # feature frames per window
        B, M, _ = input_features.shape
        max_frames = [int(v) for v in max_frames]
        seek = [0] * B
        out = [[] for _ in range(B)]
        prompt = decoder_input_ids if decoder_input_ids.shape[0] == B else decoder_input_ids.expand(B, -1)
        P = prompt.shape[1]
        n_new = (cfg.max_target_positions - P) if max_new_tokens is None else max_new_tokens
        while True:
            active = [b for b in range(B) if seek[b] < max_frames[b]]
            if not active:
                break
            left = [min(max_frames[b] - seek[b], W) for b in active]
            feats = input_features.new_zeros((len(active), M, W))
            for i, b in enumerate(active):
                feats[i, :, :left[i]] = input_features[b, :, seek[b]:seek[b] + left[i]]
            stno = stno_seek_windows(stno_mask, seek, max_frames, active, num_frames=cfg.max_source_positions)
            enr = None if enrollments is None else {k: v[active] for k, v in enrollments.items()}
            kw = dict(eos_token_id=eos, pad_token_id=pad, enrollments=enr,
                      timestamps=dict(no_timestamps_token_id=no_timestamps_token_id,
                                      max_initial_timestamp_index=gen_kw.get("max_initial_timestamp_index", 50)),
                      suppress_tokens=gen_kw.get("suppress_tokens"), begin_suppress_tokens=gen_kw.get("begin_suppress_tokens"),
                      ctc=gen_kw.get("ctc"))
            skip = [False] * len(active)
            if gen_kw.get("temperatures") is not None and self.num_beams == 1:
                # temperature fallback (reference generation.py:567-611): per-window ladder, silent windows skipped
                # (skip_by_row=True: keep a silent window's skip flag with ITS recording instead of transformers' position in
                # the fallback sub-batch -- see decode_with_fallback; default = the reference's behaviour)
                fb = {k: gen_kw[k] for k in ("compression_ratio_threshold", "logprob_threshold", "no_speech_threshold",
                                             "no_speech_token_id", "generator", "skip_by_row") if k in gen_kw}
                tok_lists, skip, _ = self.decoder.generate_with_fallback(feats, stno, prompt[active], n_new,
                                                                         temperatures=gen_kw["temperatures"], **fb, **kw)
            elif self.num_beams > 1:
                seqs, _ = self.decoder.beam_search(feats, stno, prompt[active], P + n_new, self.num_beams,
                                                   length_penalty=gen_kw.get("length_penalty", 1.0),
                                                   early_stopping=gen_kw.get("early_stopping", False), **kw)
                tok_lists = [seqs[i, P:].tolist() for i in range(len(active))]
            else:
                seqs = self.decoder.generate(feats, stno, prompt[active], n_new, **kw)
                tok_lists = [seqs[i, P:].tolist() for i in range(len(active))]
            for i, b in enumerate(active):
                if skip[i]:                                          # HF: a skipped (silent) window only moves the seek pointer
                    seek[b] += left[i]
                    continue
                toks = list(tok_lists[i])
                while toks and toks[-1] in (pad, eos):              # HF strips the eos / padding tail before segmenting
                    toks.pop()
                if not toks:
                    seek[b] += left[i]
                    continue
                segs, adv = retrieve_segment(toks, seek[b] * 0.01, ts0, left[i])
                out[b].extend(segs)
                seek[b] += adv
        return out
