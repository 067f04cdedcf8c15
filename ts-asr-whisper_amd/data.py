"""Host-side batch semantics of the hot path: STNO mask construction / collation and synthetic batches.

Product-side restatement (numpy) of the input contract the reference's dataset + collator define:
  * STNO formula            reference src/data/local_datasets.py:184-194 (pooling :166-174, unknown speaker :176-178)
  * pad-as-silence collate  reference src/data/collators.py:155-161
  * batch dict layout       reference src/data/collators.py:163,181-186,216-221
and the synthetic workloads of BASELINE.md section 3 / SURVEY.md section 8d (configs 1-5).
"""
import numpy as np
import torch

WIN = 320                 # 2 (conv2 stride) * 160 (hop): samples per encoder frame
N_SAMPLES_30S = 480000


def pool_speaker_mask(spk_mask):
    pad = (N_SAMPLES_30S - spk_mask.shape[-1]) % N_SAMPLES_30S
    spk_mask = np.pad(spk_mask, ((0, 0), (0, pad)), mode="constant")
    return spk_mask.astype(np.float32).reshape(spk_mask.shape[0], -1, WIN).mean(axis=-1)


def create_stno_masks(spk_mask, s_index):
    """[S,T] per-speaker activity -> [T,4] silence / target / non-target / overlap (sums to 1 per frame)."""
    others = np.ones(spk_mask.shape[0], dtype=bool)
    others[s_index] = False
    sil = (1 - spk_mask).prod(axis=0)
    anyone_else = (1 - spk_mask[others]).prod(axis=0)
    tgt = spk_mask[s_index] * anyone_else
    non = (1 - spk_mask[s_index]) * (1 - anyone_else)
    ovl = spk_mask[s_index] - tgt
    return np.stack([sil, tgt, non, ovl], axis=0).T


def collate_stno(masks):
    t_max = max(m.shape[0] for m in masks)
    out = np.zeros((len(masks), 4, t_max), dtype=np.float32)
    for i, m in enumerate(masks):
        out[i, :, :m.shape[0]] = m.T
        out[i, 0, m.shape[0]:] = 1.0
    return out


def synthetic_activity(rng, n_spk, T, lo=0.2, hi=3.0, rate=50):
    """Per speaker alternating on/off segments with durations U[lo,hi] s at `rate` frames/s."""
    a = np.zeros((n_spk, T), dtype=np.float32)
    for s in range(n_spk):
        t, on = 0, rng.random() < 0.5
        while t < T:
            d = int(rng.uniform(lo, hi) * rate)
            if on:
                a[s, t:t + d] = 1.0
            t += d
            on = not on
    return a


def synthetic_batch(cfg, B, L=128, seed=0, mixed_length=False, enrollments=False, device="cuda", stno_mode="speakers"):
    """Synthetic DiCoW batch: N(0,1) mel clamped to [-1.5,1.5], STNO from a 3-speaker on/off process, random labels."""
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    T, M = cfg.max_source_positions, cfg.num_mel_bins

    def one_side(length_lo):
        x = torch.randn(B, M, 2 * T, generator=g).clamp_(-1.5, 1.5)
        if stno_mode == "softmax":
            st = torch.softmax(torch.randn(B, 4, T, generator=g), dim=1).numpy()
        else:
            st = collate_stno([create_stno_masks(synthetic_activity(rng, 3, T), 0) for _ in range(B)])
        lens = np.full(B, T)
        if mixed_length:
            lens = (rng.uniform(length_lo, 30.0, B) / 30.0 * T).astype(int)
            for i, n in enumerate(lens):
                x[i, :, 2 * n:] = -1.5                     # log-mel floor of digital silence after normalisation
                st[i, :, n:] = 0.0
                st[i, 0, n:] = 1.0
        return x, torch.from_numpy(st), lens

    x, st, lens = one_side(10.0)
    labels = torch.randint(0, min(50257, cfg.vocab_size - 1), (B, L), generator=g)
    if mixed_length:
        for i, n in enumerate(lens):
            li = max(2, int(round(L * n / T)))
            labels[i, li:] = -100
    batch = {"input_features": x.to(device), "stno_mask": st.to(device), "labels": labels.to(device),
             "upp_labels": labels.clone().to(device)}
    if enrollments:
        xe, ste, _ = one_side(5.0)
        batch["enrollments"] = {"input_features": xe.to(device), "stno_mask": ste.to(device),
                                "attention_mask": torch.ones(B, 2 * T, device=device)}
    return batch
