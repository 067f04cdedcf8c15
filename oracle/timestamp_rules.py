"""Oracle (test infrastructure, numpy): Whisper's timestamp rules as the reference applies them while decoding.

Restates reference src/models/dicow/utils.py:5-14 (WhisperTimeStampLogitsProcessorCustom: at the first generated position
the eos score is restored so that a silent window can end at once) on top of transformers' WhisperTimeStampLogitsProcessor
(transformers/generation/logits_process.py; third-party, version-pinned 4.55 in the reference, same rules in the installed
5.x): <|notimestamps|> never; timestamps come in pairs; they do not decrease; the first token is a timestamp not later
than max_initial_timestamp_index; if the timestamp mass beats every text token, only timestamps remain.
Pinned against tests/golden/f14_timestamp_rules.npz (the reference class itself).  Only tests/ may import this module.
"""
import numpy as np

NEG = -np.inf


def timestamp_rules(input_ids, scores, begin_index, eos, no_timestamps, max_initial_timestamp_index=None):
    ts0 = no_timestamps + 1
    out = scores.astype(np.float32).copy()
    out[:, no_timestamps] = NEG
    for k in range(out.shape[0]):
        seq = [int(t) for t in input_ids[k, begin_index:]]
        last_ts = len(seq) >= 1 and seq[-1] >= ts0
        pen_ts = len(seq) < 2 or seq[-2] >= ts0
        if last_ts:
            if pen_ts:
                out[k, ts0:] = NEG
            else:
                out[k, :eos] = NEG
        stamps = [t for t in seq if t >= ts0]
        if stamps:
            upto = stamps[-1] if (last_ts and not pen_ts) else stamps[-1] + 1
            out[k, ts0:upto] = NEG
    first = input_ids.shape[1] == begin_index
    if first:
        out[:, :ts0] = NEG
        if max_initial_timestamp_index is not None:
            out[:, ts0 + max_initial_timestamp_index + 1:] = NEG
    for k in range(out.shape[0]):
        row = out[k]
        m = row.max()
        logp = row - (m + np.log(np.exp(row - m).sum(dtype=np.float32)))
        tsp = logp[ts0:]
        mt = tsp.max()
        ts_lp = mt + np.log(np.exp(tsp - mt).sum(dtype=np.float32)) if np.isfinite(mt) else NEG
        if ts_lp > logp[:ts0].max():
            out[k, :ts0] = NEG
    if first:                                           # utils.py:10-12
        out[:, eos] = scores[:, eos]
    return out
