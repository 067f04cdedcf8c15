"""Oracle (test infrastructure, plain Python / torch CPU): Whisper's temperature fallback.

The reference's ``generate_with_fallback`` (src/models/dicow/generation.py:567-611) prepares the STNO windows and delegates to the
method of the same name in its third-party dependency transformers (pinned 4.55.0 in requirements.txt:19; the installed 5.x
implementation is the one the goldens were taken from -- tests/golden/make_golden.py::f18_fallback records the version):
``WhisperGenerationMixin.generate_with_fallback`` / ``_need_fallback`` / ``_retrieve_compression_ratio`` /
``_retrieve_avg_logprobs`` (transformers/models/whisper/generation_whisper.py).  Restated here:

  for each temperature in order:
      decode the still-active windows (greedy at temperature 0, sampling otherwise)
      per window: strip the padding tail (one eos stays for the log-probability), then
          needs_fallback = compression_ratio(tokens) > compression_ratio_threshold  or  avg_logprob < logprob_threshold
          if avg_logprob < logprob_threshold and no_speech_prob > no_speech_threshold:  needs_fallback = False, skip the window
      keep every window's LATEST result; the windows that need a fallback are decoded again at the next temperature

Pinned against tests/golden/f18_fallback.npz (the real transformers methods driven with scripted decoder outputs).
Only tests/ may import this module.
"""
import math
import zlib

import torch


def compression_ratio(tokens, vocab_size):
    """len(raw token bytes) / len(zlib(raw token bytes)), tokens as little-endian integers of int(log2(V) / 8) + 1 bytes."""
    length = int(math.log2(vocab_size) / 8) + 1
    raw = b"".join(int(t).to_bytes(length, "little") for t in tokens)
    return len(raw) / len(zlib.compress(raw))


def avg_logprob(scores, tokens, temperature):
    """scores: [n_steps, V] processed scores (already divided by the temperature when sampling); tokens: generated ids incl.
    the eos.  HF undoes the temperature scaling (scores * temperature), takes log-softmax in fp32 and averages the chosen
    tokens' log-probabilities over len(tokens) (the eos counts)."""
    rescale = temperature if (temperature is not None and temperature > 0.0) else 1
    scores = torch.as_tensor(scores)
    tokens = list(tokens)
    if scores.shape[0] > len(tokens):
        scores = scores[:len(tokens)]
    else:
        tokens = tokens[-scores.shape[0]:]
    lp = torch.log_softmax((scores * rescale).float(), dim=-1).to(scores.dtype)
    return float(sum(lp[i][tokens[i]] for i in range(lp.shape[0])) / len(tokens))


def need_fallback(tokens, scores, temperature, vocab_size, compression_ratio_threshold, logprob_threshold, no_speech_threshold=None,
                  no_speech_prob=None):
    needs, skip = False, False
    if compression_ratio_threshold is not None and compression_ratio(tokens, vocab_size) > compression_ratio_threshold:
        needs = True
    lp = None
    if logprob_threshold is not None:
        lp = avg_logprob(scores, tokens, temperature)
        if lp < logprob_threshold:
            needs = True
    if no_speech_threshold is not None and lp is not None and lp < logprob_threshold and no_speech_prob > no_speech_threshold:
        needs, skip = False, True
    return needs, skip


def strip_padding(seq, pad, eos):
    """HF: drop the padding tail; when pad == eos one eos stays (it counts in the average log-probability)."""
    seq = list(seq)
    if seq and seq[-1] == pad:
        n = sum(1 for t in seq if t == pad)
        if pad == eos:
            n -= 1
        if n != 0:
            seq = seq[:-n]
    return seq


def fallback_loop(decode, n_windows, temperatures, vocab_size, pad, eos, compression_ratio_threshold, logprob_threshold,
                  no_speech_threshold=None, no_speech_prob=None):
    """decode(active_rows, temperature) -> (list of token lists, list of [n_steps, V] score tensors) for those windows.
    Returns (final token lists without the eos, should_skip flags, index of the temperature each window ended with)."""
    seqs, skip, used = [None] * n_windows, [False] * n_windows, [None] * n_windows
    active = list(range(n_windows))
    for k, temp in enumerate(temperatures):
        toks, scores = decode(list(active), temp)
        nxt = []
        for i, row in enumerate(active):
            seq = strip_padding(toks[i], pad, eos)
            needs, sk = need_fallback(seq, scores[i], temp, vocab_size, compression_ratio_threshold, logprob_threshold,
                                      no_speech_threshold, None if no_speech_prob is None else no_speech_prob[i])
            skip[i] = sk                              # (HF indexes should_skip by the position in the CURRENT batch)
            if seq and seq[-1] == eos:
                seq = seq[:-1]
            seqs[row], used[row] = seq, k
            if needs:
                nxt.append(row)
        active = nxt
        if not active or k == len(temperatures) - 1:
            break
    return seqs, skip, used
