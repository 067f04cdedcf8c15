"""Oracle (test infrastructure, numpy): STNO mask construction and collation.

Restates, in plain numpy:
  * speaker-activity pooling      reference src/data/local_datasets.py:162-174
  * unknown-speaker row ("-1")    reference src/data/local_datasets.py:176-178
  * S/T/N/O formula               reference src/data/local_datasets.py:184-194
  * collator pad-as-silence       reference src/data/collators.py:155-161

Channel order is 0=silence, 1=target, 2=non-target, 3=overlap.
"""
import numpy as np

N_SAMPLES_30S = 480000          # WhisperFeatureExtractor.n_samples (30 s @ 16 kHz)
HOP_LENGTH = 160                # WhisperFeatureExtractor.hop_length
SUBSAMPLE = 2                   # conv2 stride (model_features_subsample_factor)


def pool_speaker_mask(spk_mask: np.ndarray) -> np.ndarray:
    """[S, n_samples] sample-level activity -> [S, T] frame-level mean activity.

    local_datasets.py:166-174: right-pad to a multiple of 30 s, then mean over
    windows of ``SUBSAMPLE * HOP_LENGTH`` = 320 samples.
    """
    pad_len = (N_SAMPLES_30S - spk_mask.shape[-1]) % N_SAMPLES_30S
    spk_mask = np.pad(spk_mask, ((0, 0), (0, pad_len)), mode="constant")
    win = SUBSAMPLE * HOP_LENGTH
    return spk_mask.astype(np.float32).reshape(spk_mask.shape[0], -1, win).mean(axis=-1)


def create_stno_masks(spk_mask: np.ndarray, s_index: int) -> np.ndarray:
    """[S, T] activities in [0,1] + target row index -> [T, 4] (S, T, N, O).

    local_datasets.py:184-194.  ``s_index == -1`` addresses the last row, which
    the caller appends as all-zero for an unknown speaker (:176-178).
    """
    non_target = np.ones(spk_mask.shape[0], dtype=bool)
    non_target[s_index] = False
    sil = (1 - spk_mask).prod(axis=0)
    anyone_else = (1 - spk_mask[non_target]).prod(axis=0)
    tgt = spk_mask[s_index] * anyone_else
    non = (1 - spk_mask[s_index]) * (1 - anyone_else)
    ovl = spk_mask[s_index] - tgt
    return np.stack([sil, tgt, non, ovl], axis=0).T


def get_stno_mask(spk_mask_samples: np.ndarray, speaker_index: int) -> np.ndarray:
    """Sample-level activity [S, n] -> [T, 4]; speaker_index -1 = unknown speaker."""
    m = pool_speaker_mask(spk_mask_samples)
    if speaker_index == -1:
        m = np.pad(m, ((0, 1), (0, 0)), mode="constant")
    return create_stno_masks(m, speaker_index)


def collate_stno(masks) -> np.ndarray:
    """List of [T_i, 4] -> [B, 4, T_max]; frames past T_i are silence=1, rest 0.

    collators.py:155-161 (pad_sequence with zeros, transpose, set channel 0 to 1
    on the padding).
    """
    t_max = max(m.shape[0] for m in masks)
    out = np.zeros((len(masks), 4, t_max), dtype=np.float32)
    for i, m in enumerate(masks):
        out[i, :, : m.shape[0]] = m.T
        out[i, 0, m.shape[0]:] = 1.0
    return out
