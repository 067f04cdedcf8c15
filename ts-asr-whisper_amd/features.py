"""GPU log-mel feature extraction (Whisper front end) for the DiCoW path.

Host side of ``dicow_logmel`` (csrc/logmel.hip): builds the window-folded DFT tables and the slaney mel filterbank once
(float64 on the host, cast to fp32) and keeps them on the device.  Mirrors the behaviour of the feature extractor the
reference calls at src/data/local_datasets.py:208-214 (HF ``WhisperFeatureExtractor``: n_fft 400, hop 160, 30 s chunks,
slaney mel scale + norm, 0-8000 Hz), including padding to a multiple of 30 s and the frame-level attention mask.
"""
import numpy as np
import torch

from . import _lib as L
from . import ops

N_FFT, HOP, SR, N_SAMPLES = 400, 160, 16000, 480000
TABLE_LD = 224                 # row length of the DFT tables (csrc/logmel.hip: LMM_LD)
FOLD_ROWS = 204                # rows of the folded tables behind the 400 plain ones (csrc/logmel.hip: LMM_NK)
_TABLES = {}


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * (27.0 / np.log(6.4)), 3.0 * f / 200.0)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), 200.0 * m / 3.0)


def mel_filter_bank(n_mels, n_freq=1 + N_FFT // 2, fmin=0.0, fmax=8000.0, sr=SR):
    """[n_freq, n_mels] slaney-scale, slaney-normalised triangular filters."""
    fft_freqs = np.linspace(0, sr // 2, n_freq)
    pts = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    diff = np.diff(pts)
    slopes = pts[None, :] - fft_freqs[:, None]
    fb = np.maximum(0.0, np.minimum(-slopes[:, :-2] / diff[:-1], slopes[:, 2:] / diff[1:]))
    return fb * (2.0 / (pts[2:n_mels + 2] - pts[:n_mels]))[None, :]


def _tables(n_mels, device):
    key = (n_mels, str(device))
    if key not in _TABLES:
        n = np.arange(N_FFT)
        win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / N_FFT)                      # periodic hann
        ang = 2.0 * np.pi * np.outer(n, np.arange(1 + N_FFT // 2)) / N_FFT
        # rows zero-padded from 201 to TABLE_LD bins: the kernel's B fragments are 32-bin blocks of a row (7 blocks).  Rows 0..399: the
        # plain window-folded DFT (the direct kernel of -DLOGMEL_DIRECT builds); rows 400..603: the same product folded about sample
        # 200 (csrc/logmel.hip: w[n] = w[400 - n], cos symmetric, sin antisymmetric) -- row 400 + n pairs with x[n] +- x[400 - n],
        # the centre row 600 holds half its value (x[200] meets itself), the rows behind the live ones are zero
        tw_c = np.zeros((N_FFT + FOLD_ROWS, TABLE_LD), np.float32)
        tw_s = np.zeros((N_FFT + FOLD_ROWS, TABLE_LD), np.float32)
        c32 = (win[:, None] * np.cos(ang)).astype(np.float32)
        s32 = (-win[:, None] * np.sin(ang)).astype(np.float32)
        tw_c[:N_FFT, :1 + N_FFT // 2] = c32
        tw_s[:N_FFT, :1 + N_FFT // 2] = s32
        tw_c[N_FFT:N_FFT + 200, :1 + N_FFT // 2] = c32[:200]
        tw_c[N_FFT + 200, :1 + N_FFT // 2] = c32[200] * np.float32(0.5)
        tw_s[N_FFT:N_FFT + 200, :1 + N_FFT // 2] = s32[:200]
        fb = mel_filter_bank(n_mels).astype(np.float32)
        # the bins where filter m is non-zero: the projection skips the exact zeros (same sum, in the same order, as the dense loop)
        rng = np.zeros((n_mels, 2), np.int32)
        for m in range(n_mels):
            nz = np.nonzero(fb[:, m])[0]
            rng[m] = (nz[0], nz[-1] + 1) if nz.size else (0, 0)
        _TABLES[key] = tuple(torch.from_numpy(np.ascontiguousarray(t)).to(device) for t in (tw_c, tw_s, fb, rng))
    return _TABLES[key]


def pad_to_30s(waves):
    """list of 1-D float arrays/tensors -> (fp32 [B, n] padded with zeros to a multiple of 30 s, attention_mask [B, n/160])."""
    n = max(int(w.shape[-1]) for w in waves)
    tot = max(N_SAMPLES, (n + N_SAMPLES - 1) // N_SAMPLES * N_SAMPLES)
    out = torch.zeros(len(waves), tot, dtype=torch.float32)
    am = torch.zeros(len(waves), tot, dtype=torch.int32)
    for i, w in enumerate(waves):
        w = torch.as_tensor(w, dtype=torch.float32)
        out[i, :w.shape[-1]] = w
        am[i, :w.shape[-1]] = 1
    return out, am[:, ::HOP]


def log_mel(wave: torch.Tensor, n_mels: int) -> torch.Tensor:
    """wave fp32 [B, n_samples] on the GPU (already padded) -> input_features fp32 [B, n_mels, n_samples/160]."""
    if not wave.is_cuda:
        raise L.DicowError("log_mel: the waveform must be on the GPU (no CPU fallback)")
    wave = wave.contiguous().to(torch.float32)
    if wave.data_ptr() % 16 != 0:         # a contiguous view with an odd storage offset: the kernel stages rows with 16-byte loads
        wave = wave.clone()
    B, n = wave.shape
    tw_c, tw_s, fb, rng = _tables(n_mels, wave.device)
    out = torch.empty(B, n_mels, n // HOP, dtype=torch.float32, device=wave.device)
    ws = ops.workspace(L.lib().dicow_logmel_ws_bytes(B, n), wave.device)
    L.call("dicow_logmel", wave.data_ptr(), B, n, tw_c.data_ptr(), tw_s.data_ptr(), fb.data_ptr(), rng.data_ptr(), n_mels, out.data_ptr(),
           ws.data_ptr(), ws.numel(), L.stream())
    return out
