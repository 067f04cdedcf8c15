"""Oracle (test infrastructure, plain Python): long-form segment retrieval.

Restates DiCoWGenerationMixin._retrieve_segment (reference src/models/dicow/generation.py:416-534): how the tokens decoded for
one 30 s window become timed segments, and by how many feature frames the seek pointer advances.  Pinned against
tests/golden/f16_retrieve_segment.npz (the reference's static method itself).  Only tests/ may import this module.
"""


def retrieve_segment(seq, time_offset, timestamp_begin, seek_num_frames, time_precision=0.02, input_stride=2):
    """seq: list of generated token ids of the window (prompt removed); time_offset: seconds of the window start;
    seek_num_frames: feature frames left in this window (<= 3000).  Returns (segments, segment_offset) with
    segments = [dict(start, end, tokens)] and segment_offset in feature frames; raises ValueError when the offset is <= 0."""
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
            is_last = i == len(slices) - 1
            end_tok = tok[-1 if (not is_last or single_ending) else -2]
            segments.append({"start": time_offset + (tok[0] - timestamp_begin) * time_precision,
                             "end": time_offset + (end_tok - timestamp_begin) * time_precision, "tokens": tok})
            last = cur
        offset = seek_num_frames if single_ending else (seq[last - 2] - timestamp_begin) * input_stride
    else:
        stamps = [t for t, f in zip(seq, is_ts) if f]
        start_pos, last_pos = 0.0, seek_num_frames // 2
        skip, offset = False, seek_num_frames
        if len(stamps) > 1:
            start_pos, last_pos = stamps[-2] - timestamp_begin, stamps[-1] - timestamp_begin
        elif len(stamps) == 1:
            start_pos = stamps[-1] - timestamp_begin
            if start_pos > 200:                       # does not fit into the window: roll back to just before it
                offset, skip = start_pos * input_stride - 100, True
        elif len(seq) > 1:
            pass                                      # decoding without timestamps: one segment spanning the window
        else:
            skip = True
        if not skip:
            segments = [{"start": time_offset + start_pos * time_precision, "end": time_offset + last_pos * time_precision,
                         "tokens": list(seq)}]
            offset = seek_num_frames
    if offset <= 0:
        raise ValueError(f"segment offset {offset} <= 0")
    return segments, int(offset)


# ---------------------------------------------------------------------------------------------------------------------------
# Global-time segments -> per-window timestamp tokens (reference generation.py:313-415, _fix_timestamps_from_segmentation)

def _ticks_half_up(x):
    """Seconds -> 0.02 s ticks the way the reference rounds them (generation.py:314-320: Decimal(str(x)), ROUND_HALF_UP)."""
    from decimal import Decimal, ROUND_HALF_UP
    return (Decimal(str(x)) / Decimal("0.02")).to_integral_value(rounding=ROUND_HALF_UP) * Decimal("0.02")


def fold_segments(segments, first_timestamp_token, filler_token):
    """One recording's segments [dict(start, end, tokens)] in recording time -> [(start, tokens, end)] in window time
    (0..30 s), with the filler entries the reference inserts when a 30 s block boundary is crossed or blocks are skipped.
    Follows the reference's Decimal arithmetic literally, including its inexact ``Decimal(-0.02)`` carry (generation.py:392)."""
    from decimal import Decimal
    W = 30
    live = [s for s in segments if len(s["tokens"]) > 0 and not (len(s["tokens"]) == 1 and s["tokens"][0] == first_timestamp_token)]
    out, prev_end, carry = [], None, Decimal(0.0)
    for seg in live:
        t0, t1 = _ticks_half_up(float(seg["start"])), _ticks_half_up(float(seg["end"]))
        if prev_end is None:
            out += [(0, [filler_token], 30)] * int(t0 // W)
        else:
            here, before = (t0 + carry) // W, (prev_end - Decimal("0.001")) // W
            if here > before:
                out.append((30, [filler_token], 30))
            out += [(0, [filler_token], 30)] * max(int(here - before - 1), 0)
        a, b = t0 + carry, t1 + carry
        if a // W == b // W:
            out.append((a % W, seg["tokens"], b % W))
        elif b % W == 0:
            out.append((a % W, seg["tokens"], 30))
            carry = Decimal(0.0)
        else:
            s_new, e_new = a % W, b % W
            if t1 - t0 == 30.0:
                if float(s_new) % 30.0 == 0.0:
                    e_new, carry = Decimal(30.0), Decimal(0.0)
                else:
                    carry = Decimal(-0.02)
                    e_new += carry
            else:
                carry = Decimal(0.0)
            out.append((s_new, seg["tokens"], e_new))
        prev_end = t1 + carry
    return out


def folded_to_ids(folded, first_timestamp_token):
    """[(start, tokens, end)] -> token ids ``<|start|> text... <|end|>`` per entry.  The reference goes through the tokenizer
    (generation.py:402-405: format '<|%.2f|>', decode, re-encode); with a tokenizer whose decode/encode round trip is the
    identity on text tokens that equals this direct mapping (timestamp ids inside ``tokens`` are dropped by decode())."""
    ids = []

    def stamp(x):
        text = f"{x:.2f}"
        if text.startswith("-"):          # the inexact carry can leave -4e-19, printed '<|-0.00|>': not a timestamp token
            return []
        return [first_timestamp_token + int(round(float(text) / 0.02))]

    for s, toks, e in folded:
        ids += stamp(s) + [t for t in toks if t < first_timestamp_token] + stamp(e)
    return ids
