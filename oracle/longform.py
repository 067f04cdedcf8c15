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
