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
