"""CPU tests of the host-side logic: STNO builder bit-exactness, config, C-ABI library symbols, state-dict surface."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import amd_pkg
from tests.util import load_golden, ROOT

pkg = amd_pkg.load()


def test_product_stno_builder_bit_exact_vs_reference_golden():
    from ts_asr_whisper_amd import data
    z = load_golden("f1_stno")
    for i in range(int(z["n_cases"])):
        got = data.create_stno_masks(z[f"in_{i}"].copy(), int(z[f"idx_{i}"]))
        assert np.array_equal(got, z[f"out_{i}"]), i


def test_collate_stno_padding_is_silence():
    from ts_asr_whisper_amd import data
    out = data.collate_stno([np.full((3, 4), 0.25, np.float32), np.full((5, 4), 0.25, np.float32)])
    assert out.shape == (2, 4, 5) and np.all(out[0, 0, 3:] == 1) and np.all(out[0, 1:, 3:] == 0)


def test_library_exports_every_declared_symbol():
    """include/dicow_hip.h is the ABI: every function it declares must be exported by the built library and bound."""
    from ts_asr_whisper_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "dicow_hip.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(dicow_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 25
    assert declared == set(_lib.declared_symbols())
    lib = _lib.lib()                       # raises loudly if the .so has not been built
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.dicow_abi_version() == 1


def test_no_cpu_fallback_and_oracle_not_imported_by_product():
    src_dir = os.path.join(ROOT, "ts-asr-whisper_amd")
    for fn in os.listdir(src_dir):
        if fn.endswith(".py"):
            txt = open(os.path.join(src_dir, fn)).read()
            assert "import oracle" not in txt and "from oracle" not in txt, fn
    m = pkg.FDDT(64, is_diagonal=True)
    with pytest.raises(Exception):
        m(torch.randn(1, 4, 64), torch.rand(1, 4, 4))


def test_state_dict_surface_matches_reference_keys():
    z = load_golden("f8_e2e_se")
    import ast
    d = ast.literal_eval(str(z["cfg"]))
    model = pkg.DiCoWForConditionalGeneration(pkg.DiCoWConfig(**d))
    assert set(model.state_dict().keys()) == {k[2:] for k in z.files if k.startswith("p.")}
    assert model.proj_out.weight is model.model.decoder.embed_tokens.weight


def test_reference_init_semantics():
    cfg = pkg.DiCoWConfig(d_model=128, encoder_layers=2, encoder_attention_heads=2, decoder_layers=1, decoder_attention_heads=2,
                          encoder_ffn_dim=256, decoder_ffn_dim=256, vocab_size=512, max_source_positions=50, pad_token_id=500,
                          use_pre_pos_fddt=True, non_target_fddt_value=0.5, use_enrollments=True, scb_layers=1)
    m = pkg.DiCoWForConditionalGeneration(cfg)
    e = m.model.encoder
    for f in e.fddts:                       # per-layer FDDTs start as identity
        for c in ("silence_linear", "target_linear", "non_target_linear", "overlap_linear"):
            assert torch.all(getattr(f, c).weight == 1) and torch.all(getattr(f, c).bias == 0)
    i = e.initial_fddt
    assert torch.all(i.silence_linear.weight == 0.5) and torch.all(i.non_target_linear.weight == 0.5)
    assert torch.all(i.target_linear.weight == 1) and torch.all(i.overlap_linear.weight == 1)
    assert float(e.ca_enrolls[0].cae.cross_gate.gate) == 0.0
    assert not e.embed_positions.weight.requires_grad


def test_config_rejects_unsupported():
    with pytest.raises(ValueError):
        pkg.DiCoWConfig(d_model=100, encoder_attention_heads=2, decoder_attention_heads=2)
    with pytest.raises(ValueError):
        pkg.DiCoWConfig(dropout=0.1)


def test_product_mel_filterbank_matches_oracle():
    from ts_asr_whisper_amd import features
    from oracle import logmel as ol
    for m in (80, 128):
        assert np.allclose(features.mel_filter_bank(m), ol.mel_filter_bank(m), atol=1e-12)
