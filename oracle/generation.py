"""Oracle (test infrastructure, numpy / torch CPU fp32): the decoding-side pieces of SURVEY.md section 8 row f4.

  * stno_seek_windows   reference DiCoWGenerationMixin.prepare_kwargs_for_generate, src/models/dicow/generation.py:73-106:
                        per active sample the STNO mask of the current 30 s seek window, right-padded as silence.
                        Pinned against tests/golden/f12_seek.npz (the reference method itself, run on a stand-in self).
  * greedy_decode       what HF's greedy search does with the DiCoW decoder for short-form decoding: full teacher-forced
                        forward of the prefix at every step (no cache), SuppressTokens / SuppressTokensAtBegin logits
                        processors (generation.py:286-306), eos bookkeeping with pad fill.  PARITY UNPINNED against the
                        reference's own ``generate``: at this commit it only runs with enrollments or legacy
                        forced_decoder_ids (generation.py:125-149) and rides on transformers-4.55 generation internals
                        that the installed 5.x no longer has; the decoder arithmetic it steps is the one golden F7 pins.

Only tests/ may import this module.
"""
import numpy as np
import torch

from . import dicow_oracle as O


def stno_seek_windows(stno: np.ndarray, seek, max_frames, batch_idx_map, num_frames: int = 1500) -> np.ndarray:
    """stno fp32 [B_all, 4, T_total] (encoder-rate frames), seek / max_frames in feature frames (2 per STNO frame) for every
    original sample, batch_idx_map: original index of each still-active sample -> [len(map), 4, num_frames]."""
    out = []
    for prev in batch_idx_map:
        s = int(seek[prev]) // 2
        n = min(int(max_frames[prev]) // 2 - s, num_frames)
        w = stno[prev, :, s:s + n]
        if w.shape[-1] < num_frames:
            pad = np.zeros((4, num_frames - w.shape[-1]), dtype=stno.dtype)
            pad[0] = 1.0
            w = np.concatenate([w, pad], axis=-1)
        out.append(w)
    return np.stack(out)


@torch.no_grad()
def greedy_decode(p, cfg, input_features, stno_mask, prompt_ids, max_new_tokens, eos_token_id, pad_token_id,
                  suppress_tokens=None, begin_suppress_tokens=None, enrollments=None, emu=False):
    """Returns (sequences int64 [B, P + n], per-step processed fp32 scores [n, B, V])."""
    enc = O.encoder_forward(p, cfg, input_features, stno_mask, enrollments=enrollments, emu=emu)
    ids = prompt_ids.clone()
    begin = ids.shape[1]
    unfinished = torch.ones(ids.shape[0], dtype=torch.bool)
    scores = []
    for _ in range(max_new_tokens):
        logits = O.linear(O.decoder_forward(p, cfg, ids, enc, emu=emu)[:, -1, :], p["proj_out.weight"], None, emu).float()
        if suppress_tokens:
            logits[:, list(suppress_tokens)] = -float("inf")
        if begin_suppress_tokens and ids.shape[1] == begin:
            logits[:, list(begin_suppress_tokens)] = -float("inf")
        scores.append(logits)
        nxt = logits.argmax(-1)
        nxt = torch.where(unfinished, nxt, torch.full_like(nxt, pad_token_id))
        ids = torch.cat([ids, nxt[:, None]], dim=1)
        unfinished = unfinished & (nxt != eos_token_id)
        if not bool(unfinished.any()):
            break
    return ids, torch.stack(scores)


@torch.no_grad()
def detect_language(p, cfg, input_features, stno_mask, decoder_start_token_id, lang_token_ids, enrollments=None, emu=False):
    """DiCoWGenerationMixin.detect_language (reference generation.py:151-221): one decoder position on the start token, with
    the window's STNO mask (and enrollments) conditioning the encoder; every non-language logit is masked and the argmax is
    the language token.  Returns (language ids int64 [B], fp32 logits [B, V] before masking)."""
    enc = O.encoder_forward(p, cfg, input_features, stno_mask, enrollments=enrollments, emu=emu)
    ids = torch.full((input_features.shape[0], 1), decoder_start_token_id, dtype=torch.long)
    logits = O.linear(O.decoder_forward(p, cfg, ids, enc, emu=emu)[:, -1, :], p["proj_out.weight"], None, emu).float()
    masked = torch.full_like(logits, -float("inf"))
    lang = torch.as_tensor(list(lang_token_ids), dtype=torch.long)
    masked[:, lang] = logits[:, lang]
    return masked.argmax(-1), logits
