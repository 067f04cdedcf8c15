"""Shared helpers for the test-suite (fixture loading, config parsing)."""
import ast
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def golden_cfg(z):
    from oracle.dicow_oracle import OracleConfig
    d = ast.literal_eval(str(z["cfg"]))
    return OracleConfig(**{k: v for k, v in d.items() if k in OracleConfig.__dataclass_fields__})


def golden_params(z, prefix="p.", requires_grad=False, dtype=torch.float32):
    out = {}
    for k in z.files:
        if k.startswith(prefix):
            t = torch.from_numpy(z[k]).to(dtype).clone()
            out[k[len(prefix):]] = t
    if "proj_out.weight" in out and "model.decoder.embed_tokens.weight" in out:
        out["proj_out.weight"] = out["model.decoder.embed_tokens.weight"]      # tied
    if requires_grad:
        for t in set(out.values()):
            if t.is_floating_point():
                t.requires_grad_(True)
    return out


def T(z, k, dtype=None):
    t = torch.from_numpy(np.asarray(z[k]))
    if dtype is not None:
        t = t.to(dtype)
    elif t.dtype == torch.float16:
        t = t.float()
    return t


def maxdiff(a, b):
    return float((a.double() - b.double()).abs().max())


def hashed_mel(B, M, T):
    """Deterministic pseudo-random mel in [-1, 1] from integer arithmetic only; the inputs of golden F11's SpecAug cases
    (same formula as tests/golden/make_golden.py:hashed_mel)."""
    b = np.arange(B, dtype=np.int64)[:, None, None]
    m = np.arange(M, dtype=np.int64)[None, :, None]
    t = np.arange(T, dtype=np.int64)[None, None, :]
    h = (m * 7919 + t * 104729 + b * 1299709 + (m * t) % 613 * 31) % 2001 - 1000
    return (h.astype(np.float64) / 1000.0).astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------------------
# Deterministic, integer-hashed parameters and inputs for the REAL-DIMENSION goldens (tests/golden/make_golden_realdims.py):
# the fixtures of whisper-tiny / -base / -large-v3-turbo sized models store OUTPUTS only -- both the reference run (build
# container) and the GPU test regenerate identical weights / inputs from these integer formulas (exact in int64 and in the
# final int -> float32 conversion, on any device).
def _crc(name):
    import zlib
    return zlib.crc32(name.encode()) & 0x7FFFFFFF


def hashed_uniform(name, shape, device="cpu"):
    """float32 tensor of `shape`, uniform in [-1, 1) on a 2^-23 grid, a pure function of (name, flat index)."""
    n = 1
    for d in shape:
        n *= int(d)
    seed = _crc(name)
    out = torch.empty(n, dtype=torch.float32, device=device)
    CH = 1 << 24
    for s in range(0, n, CH):
        idx = torch.arange(s, min(n, s + CH), dtype=torch.int64, device=device)
        h = (idx * 1540483477 + seed * 40503 + 12345) & 0xFFFFFFFF
        h = h ^ (h >> 15)
        h = (h * 1103515245 + 12345) & 0xFFFFFFFF
        h = h ^ (h >> 13)
        h = (h * 1540483477) & 0xFFFFFFFF
        h = h ^ (h >> 16)
        out[s:s + idx.numel()] = ((h & 0xFFFFFF).to(torch.float32) - 8388608.0) / 8388608.0
    return out.view(*shape)


def hashed_init_(model, skip=("embed_positions",)):
    """Overwrite every parameter of a (reference or product) DiCoW model with hashed values scaled like a trained model's:
    LayerNorm weights 1 + 0.1u, diagonal FDDT weights base + 0.1u (base = the reference's suppressive initialisation),
    matrices fan_in^-0.5 * u * 1.7 (unit-variance outputs), biases / vectors 0.1u.  The encoder's sinusoidal positions stay."""
    with torch.no_grad():
        seen = set()
        for n, p in model.named_parameters():
            if id(p) in seen or any(s in n and "encoder" in n for s in skip):
                continue
            seen.add(id(p))
            u = hashed_uniform(n, tuple(p.shape), device=p.device)
            if n.endswith("layer_norm.weight"):
                v = 1.0 + 0.1 * u
            elif "fddt" in n and n.endswith(".weight") and p.dim() == 1:
                base = 0.5 if ("initial_fddt" in n and ("silence" in n or "non_target" in n)) else 1.0
                v = base + 0.1 * u
            elif "fddt" in n and n.endswith(".weight") and p.dim() == 2:
                v = torch.eye(p.shape[0], device=p.device) + 0.3 * p.shape[0] ** -0.5 * u
            elif n.endswith("gate"):
                v = 0.3 + 0.5 * u
            elif p.dim() >= 2:
                v = (1.7 * p[0].numel() ** -0.5) * u
            else:
                v = 0.1 * u
            p.copy_(v.to(p.dtype))


def hashed_stno(B, T, tag="stno"):
    """[B, 4, T] soft STNO masks (rows sum to 1; ~30 % of the frames one-hot, the last T // 10 frames silence = 1 like the
    collator's padding, reference collators.py:157-161) from integer hashes only."""
    u = hashed_uniform(tag, (B, T, 5)).double()
    e = torch.exp(2.0 * u[..., :4] * 1.5)
    s = e / e.sum(-1, keepdim=True)
    hard = torch.nn.functional.one_hot(((u[..., 0] + 1.0) * 2.0).floor().clamp(0, 3).long(), 4).double()
    pick = (u[..., 4] < -0.4).double()[..., None]
    s = pick * hard + (1 - pick) * s
    s[:, -(T // 10):, :] = 0.0
    s[:, -(T // 10):, 0] = 1.0
    return s.permute(0, 2, 1).contiguous().float()


def hashed_labels(B, L, lo, hi, tag="labels", pad_rows=()):
    """int64 [B, L] labels in [lo, hi); rows listed in pad_rows get their last L // 4 positions set to -100."""
    u = hashed_uniform(tag, (B, L)).double()
    lab = (lo + ((u + 1.0) * 0.5 * (hi - lo)).floor()).clamp(lo, hi - 1).long()
    for r in pad_rows:
        lab[r, L - L // 4:] = -100
    return lab


def subsample(t, n=256):
    """Deterministic flat subsample of a tensor: n elements at a stride that is coprime with every dimension (a stride that
    is a multiple of the row length would sample ONE column; the fixture side and the test side agree on the rule)."""
    import math
    f = t.reshape(-1)
    step = max(1, f.numel() // n)
    dims = [int(d) for d in t.shape if int(d) > 1]
    while step > 1 and any(math.gcd(step, d) != 1 for d in dims):
        step -= 1
    return f[::step][:n].clone()


def sketch(t, name, k=4):
    """k inner products of the WHOLE tensor with hashed +-1 vectors: a direction-sensitive checksum of tensors too large to store."""
    f = t.reshape(-1).double()
    out = []
    for i in range(k):
        sgn = torch.where(hashed_uniform(f"{name}.sketch{i}", (f.numel(),), device=f.device) < 0, -1.0, 1.0).double()
        out.append((f * sgn).sum())
    return torch.stack(out).float()


# ---------------------------------------------------------------------------------------------------------------------------
# F20 (training trajectory): the batches of tests/golden/make_golden_trajectory.py -- two distinct hashed batches, alternating.
F20_PREFIXES = ["model.encoder.additional_layer", "model.encoder.additional_self_attention_layer", "model.encoder.lm_head",
                "model.encoder.subsample_conv1", "model.encoder.subsample_conv2", "model.encoder.fddts",
                "model.encoder.initial_fddt", "model.encoder.ca_enrolls"]          # reference configs/base.yaml:18


def f20_batches(case, K, ts0=None):
    """case "small": 80 mels, 100 frames, B = 2, 10 labels below id 400; "small_se": the same with an enrollment per row
    (SE-DiCoW); "tiny": whisper-tiny dimensions, B = 1, 32 labels with a timestamp token (ts0 = first timestamp id) in front."""
    se = case == "small_se"
    B, L, T, M, vocab_hi = (2, 10, 100, 80, 400) if case.startswith("small") else (1, 32, 1500, 80, 50257)
    mel = torch.from_numpy(hashed_mel((4 if se else 2) * B, M, 2 * T)).clone() * 1.5
    two = []
    for j in range(2):
        lab = hashed_labels(B, L, 0, vocab_hi, f"f20.{case}.{j}.labels", pad_rows=(B - 1,) if B > 1 else ())
        if ts0 is not None:
            lab[:, 0] = ts0 + 7 + j
        b = dict(input_features=mel[j * B:(j + 1) * B], stno_mask=hashed_stno(B, T, f"f20.{case}.{j}.stno"), labels=lab,
                 upp_labels=lab.clone())
        if se:
            b["enrollments"] = {"input_features": mel[(2 + j) * B:(3 + j) * B], "stno_mask": hashed_stno(B, T, f"f20.{case}.{j}.enr.stno")}
        two.append(b)
    return [two[k % 2] for k in range(K)]
