"""Oracle (test infrastructure, numpy): Whisper log-mel feature extraction.

Restates the arithmetic the reference reaches at src/data/local_datasets.py:208-214
through the third-party ``transformers`` feature extractor (pinned 4.55.0 in the
reference's requirements.txt:22; source not vendored under /root/reference):
  * slaney mel filterbank           HF audio_utils.mel_filter_bank (norm="slaney", mel_scale="slaney")
  * centred reflect-padded STFT     HF feature_extraction_whisper.py:135-165 (torch.stft, hann(400) periodic, hop 160)
  * power, mel projection, log10(clamp 1e-10), max(x, max-8), (x+4)/4, drop last frame.
Pinned against fixture F2 generated from the real WhisperFeatureExtractor
(tests/golden/make_golden.py).
"""
import numpy as np

N_FFT = 400
HOP = 160
SR = 16000
N_SAMPLES = 480000


def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    min_log_hertz, min_log_mel, logstep = 1000.0, 15.0, 27.0 / np.log(6.4)
    mels = 3.0 * f / 200.0
    log_region = f >= min_log_hertz
    mels = np.where(log_region, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hertz) * logstep, mels)
    return mels


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    min_log_hertz, min_log_mel, logstep = 1000.0, 15.0, np.log(6.4) / 27.0
    f = 200.0 * m / 3.0
    log_region = m >= min_log_mel
    f = np.where(log_region, min_log_hertz * np.exp(logstep * (m - min_log_mel)), f)
    return f


def mel_filter_bank(n_mels: int, n_freq: int = 1 + N_FFT // 2, fmin=0.0, fmax=8000.0, sr=SR) -> np.ndarray:
    """[n_freq, n_mels] triangular slaney-normalised filters (float64)."""
    fft_freqs = np.linspace(0, sr // 2, n_freq)
    mel_pts = np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2)
    filt = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(filt)
    slopes = filt[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / fdiff[:-1]
    up = slopes[:, 2:] / fdiff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (filt[2: n_mels + 2] - filt[:n_mels])
    return fb * enorm[None, :]


def hann_periodic(n=N_FFT):
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)).astype(np.float32)


def pad_to_30s(wave: np.ndarray):
    """Right-pad with zeros to a multiple of 30 s; returns (padded, attention_mask[frames])."""
    n = wave.shape[-1]
    tot = max(N_SAMPLES, ((n + N_SAMPLES - 1) // N_SAMPLES) * N_SAMPLES)
    out = np.zeros(tot, dtype=np.float32)
    out[:n] = wave
    am = np.zeros(tot, dtype=np.int32)
    am[:n] = 1
    return out, am[::HOP]


def log_mel(wave: np.ndarray, n_mels: int, dtype=np.float32) -> np.ndarray:
    """[n_samples] (already padded to 30 s multiples) -> [n_mels, n_samples/160] log-mel."""
    x = np.asarray(wave, dtype=dtype)
    pad = N_FFT // 2
    xp = np.pad(x, (pad, pad), mode="reflect")
    n_frames = 1 + (xp.shape[0] - N_FFT) // HOP
    idx = np.arange(N_FFT)[None, :] + HOP * np.arange(n_frames)[:, None]
    frames = xp[idx] * hann_periodic().astype(dtype)[None, :]
    spec = np.fft.rfft(frames.astype(np.float64), axis=-1)
    mag = (spec.real ** 2 + spec.imag ** 2).astype(dtype)[:-1]          # drop last frame
    mel = mag @ mel_filter_bank(n_mels).astype(dtype)                   # [frames, n_mels]
    log_spec = np.log10(np.maximum(mel, 1e-10))
    log_spec = np.maximum(log_spec, log_spec.max() - 8.0)
    log_spec = (log_spec + 4.0) / 4.0
    return log_spec.T.astype(np.float32)
