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
    stable, experimental = hdr.split("#ifdef DICOW_EXPERIMENTAL_ABI")
    assert set(re.findall(r"^int\s+(dicow_\w+)\s*\(", experimental, flags=re.M)) == set(_lib._SIGS_EXPERIMENTAL)
    hdr = stable                           # (the experimental tail is declared only under DICOW_EXPERIMENTAL_ABI and not exported by default)
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(dicow_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 25
    assert declared == set(_lib.declared_symbols())
    lib = _lib.lib()                       # raises loudly if the .so has not been built
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.dicow_abi_version() == 7
    if not _lib.has_experimental():        # the shipped build: nothing but the stable ABI is exported
        import subprocess
        out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
        exported = {ln.split()[-1] for ln in out.splitlines() if " T dicow_" in ln}
        assert exported == declared, exported ^ declared


def test_gemm_nt_is_persistent_host_logic():
    """dicow_gemm_nt_is_persistent is host logic (no launch): the headline fc2 shape runs on the persistent kernel -- the only one
    that implements DICOW_EPI_FDDT -- and a decoder-sized or ragged-K problem does not.  Also pins the ctypes layout of
    dicow_gemm_args up to the fields the predicate reads."""
    import ctypes as C
    from ts_asr_whisper_amd import _lib
    lib = _lib.lib()

    def q(M, N, K, batch=1):
        a = _lib.GemmArgs()
        a.A = a.B = a.C = 1 << 20
        a.M, a.N, a.K = M, N, K
        a.lda, a.ldb, a.ldc, a.ldr, a.ldaux, a.batch = K, K, N, N, N, batch
        return lib.dicow_gemm_nt_is_persistent(C.byref(a))

    assert q(24000, 1280, 5120) == 1 and q(24000, 5120, 1280) == 1 and q(12000, 1536, 512) == 1
    assert q(1024, 512, 512) == 0 and q(24000, 1280, 64) == 0 and q(200, 1280, 1280) == 0
    assert q(12000, 512, 2048) == 0                      # whisper-base's N = 512 shapes stay on the 128 x 128 kernel
    assert _lib.EPI_FDDT == 1024 and C.sizeof(_lib.GemmArgs) % 8 == 0


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


def test_default_preheat_group_is_the_reference_prefix_list():
    """reference configs/base.yaml:18 + configs/train/se_dicow.yaml:13: FDDTs, the CTC branch and the enrollment cross-attention
    form the preheat group (lr x multiplier, weight decay 0, the only parameters of phase 1) -- by DEFAULT, as in the recipes."""
    from ts_asr_whisper_amd.trainer import FlatStore, REFERENCE_PREHEAT_PREFIXES, freeze_by_keyword
    cfg = pkg.DiCoWConfig(d_model=128, encoder_layers=2, encoder_attention_heads=2, decoder_layers=1, decoder_attention_heads=2,
                          encoder_ffn_dim=256, decoder_ffn_dim=256, vocab_size=512, max_source_positions=50, pad_token_id=500,
                          use_pre_pos_fddt=True, use_enrollments=True, scb_layers=1, ctc_weight=0.3, additional_layer=True,
                          additional_self_attention_layer=True, pre_ctc_sub_sample=True)
    m = pkg.DiCoWForConditionalGeneration(cfg)
    freeze_by_keyword(m, ("decoder",))
    store = FlatStore(m)
    names = {id(p): n for n, p in m.named_parameters()}
    pre = {names[id(p)] for p, _, _, is_pre in store.entries if is_pre}
    want = {n for n, p in m.named_parameters() if p.requires_grad and n.startswith(REFERENCE_PREHEAT_PREFIXES)}
    assert pre == want
    for part in ("fddts", "initial_fddt", "ca_enrolls", "lm_head", "additional_layer", "additional_self_attention_layer", "subsample_conv1"):
        assert any(("model.encoder." + part) in n for n in pre), part
    assert not any("layers." in n and "additional" not in n and "ca_enrolls" not in n for n in pre)


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


# ------------------------------------------------------------------------------------------------ augmentation planner
def _next_draw():
    return float(torch.rand(1))


def test_augment_planner_consumes_the_generator_like_the_reference_restatement():
    """The planner must make the same draws in the same order as the collator (pinned by the oracle against golden F11):
    after either one runs from the same seed the generator has to be in the same state."""
    from oracle import augment as oaug
    from ts_asr_whisper_amd import augment as paug
    stno = load_golden("f11_augment")["stno"]
    B, C, Tn = stno.shape
    torch.manual_seed(3)
    oaug.soft_segment_augmentation(stno.copy(), 0.3, 4, 30)
    a = _next_draw()
    torch.manual_seed(3)
    segs, coef = paug.plan_soft_segments(B, C, Tn, 0.3, 4, 30)
    assert _next_draw() == a
    assert segs.dtype == torch.int32 and segs.shape[1] == 4 and coef.shape == (segs.shape[0], 2)
    assert (segs[:, 2] > segs[:, 1]).all() and (segs[:, 2] - segs[:, 1] <= 30).all() and (segs[:, 3] < C - 1).all()
    assert torch.allclose(coef.sum(1), torch.ones(len(coef)))

    torch.manual_seed(4)
    oaug.add_gaussian_noise_and_rescale(stno.copy(), 0.2, 0.75)
    a = _next_draw()
    torch.manual_seed(4)
    rows, noise, sd = paug.plan_gaussian_noise(B, C, Tn, 0.2, 0.75)
    assert _next_draw() == a
    assert rows.numel() == int(B * 0.75) == noise.shape[0] and len(set(rows.tolist())) == rows.numel()
    assert sd == float(np.float32(0.2 ** 0.5))
    assert paug.plan_gaussian_noise(B, C, Tn, 0.2, 0.1) is None              # int(6 * 0.1) == 0 rows: no draw at all

    x = np.zeros((3, 400, 132), np.float32)
    torch.manual_seed(5)
    oaug.spec_aug(x)
    a = _next_draw()
    torch.manual_seed(5)
    plan = paug.plan_spec_aug(3, 400, 132)
    assert _next_draw() == a
    assert plan.fmask.shape == (3, 2, 2) and plan.tmask.shape == (3, 5, 2)
    assert 0 < plan.warped < 400 and abs(plan.warped - plan.center) <= 5
    assert int(plan.fmask[..., 1].max()) < 27 and int(plan.tmask[..., 1].max()) < 20
    assert int((plan.fmask[..., 0] + plan.fmask[..., 1]).max()) <= 128


def test_augment_planner_short_inputs_skip_the_warp():
    from ts_asr_whisper_amd import augment as paug
    torch.manual_seed(0)
    plan = paug.plan_spec_aug(2, 10, 132)               # T - window <= window: no warp draw (augmentations.py:101)
    assert plan.warped < 0
    assert plan.tmask.numel() == 0                      # floor(10 * 0.05) == 0: the ratio mask is skipped as well


def test_augment_refuses_cpu_tensors():
    from ts_asr_whisper_amd import augment as paug
    with pytest.raises(Exception, match="GPU"):
        paug.add_gaussian_noise_and_rescale(torch.rand(2, 4, 8), 0.2, 1.0)
    with pytest.raises(Exception, match="GPU"):
        paug.spec_aug_joint(torch.rand(1, 80, 20), torch.rand(1, 4, 10))


# ------------------------------------------------------------------------------------------------ optimizer schedule
def test_lr_schedule_matches_hf_cosine_with_warmup():
    """k-th optimizer step of FusedAdamW uses the lr HF's Trainer would (LambdaLR steps after the optimizer)."""
    import types
    import transformers
    from oracle.optim import cosine_with_warmup_lambda
    from ts_asr_whisper_amd.trainer import FusedAdamW
    store = types.SimpleNamespace(params=torch.zeros(1), runs=[])
    opt = FusedAdamW(store, lr=2e-6, warmup_steps=7, max_steps=50)
    ref_opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=2e-6)
    sched = transformers.get_cosine_schedule_with_warmup(ref_opt, 7, 50)
    for k in range(1, 56):
        want = ref_opt.param_groups[0]["lr"]
        assert abs(opt.lr_at(k) - want) < 1e-18, k
        assert abs(2e-6 * cosine_with_warmup_lambda(k - 1, 7, 50) - want) < 1e-18
        ref_opt.step()
        sched.step()
    assert FusedAdamW(store, lr=1e-3).lr_at(1) == 1e-3            # no schedule configured: constant


def test_product_stno_seek_windows_match_reference_golden():
    from ts_asr_whisper_amd.generation import stno_seek_windows
    z = load_golden("f12_seek")
    for i in range(int(z["n_cases"])):
        got = stno_seek_windows(torch.from_numpy(z[f"c{i}.stno"]), z[f"c{i}.seek"], z[f"c{i}.max_frames"], z[f"c{i}.map"],
                                num_frames=int(z[f"c{i}.msp"]))
        assert np.array_equal(got.numpy(), z[f"c{i}.out"]), i


# ------------------------------------------------------------------------------------------------ checkpoints
def _tiny_cfg():
    return pkg.DiCoWConfig(vocab_size=256, d_model=64, encoder_layers=2, encoder_attention_heads=1, decoder_layers=1,
                           decoder_attention_heads=1, encoder_ffn_dim=128, decoder_ffn_dim=128, max_source_positions=20,
                           max_target_positions=16, pad_token_id=250, use_pre_pos_fddt=True)


def test_save_and_from_pretrained_round_trip(tmp_path):
    torch.manual_seed(0)
    m = pkg.DiCoWForConditionalGeneration(_tiny_cfg())
    m.save_pretrained(tmp_path / "ckpt")
    assert sorted(os.listdir(tmp_path / "ckpt")) == ["config.json", "generation_config.json", "model.safetensors"]      # (round 6: an HF PreTrainedModel)
    m2 = pkg.DiCoWForConditionalGeneration.from_pretrained(str(tmp_path / "ckpt"))
    assert m2._load_report == {"missing": [], "unexpected": []}
    for (n, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), n
    assert m2.proj_out.weight is m2.model.decoder.embed_tokens.weight          # tied after loading
    assert m2.config.to_dict() == m.config.to_dict()


def test_from_pretrained_plain_whisper_checkpoint_keeps_fddt_init(tmp_path):
    """A checkpoint without the FDDT keys (plain Whisper): they are reported missing and keep the reference initialisation,
    overrides reach the config (containers.py:47-50 passes the DiCoW switches as from_pretrained kwargs)."""
    from safetensors.torch import save_file
    import json
    torch.manual_seed(0)
    m = pkg.DiCoWForConditionalGeneration(_tiny_cfg())
    sd = {k: v.clone() for k, v in m.state_dict().items() if "fddt" not in k and k != "proj_out.weight"}
    os.makedirs(tmp_path / "w")
    save_file(sd, str(tmp_path / "w" / "model.safetensors"))
    plain = {k: v for k, v in m.config.to_dict().items() if "fddt" not in k}
    json.dump(plain, open(tmp_path / "w" / "config.json", "w"))
    m2 = pkg.DiCoWForConditionalGeneration.from_pretrained(str(tmp_path / "w"), use_fddt=True, use_pre_pos_fddt=True,
                                                          fddt_init="suppressive", non_target_fddt_value=0.5)
    assert m2._load_report["missing"] and all("fddt" in k for k in m2._load_report["missing"]) and not m2._load_report["unexpected"]
    assert torch.equal(m2.model.encoder.layers[0].fc1.weight, m.model.encoder.layers[0].fc1.weight)
    w = m2.model.encoder.initial_fddt.non_target_linear.weight
    assert torch.allclose(w, torch.full_like(w, 0.5))                           # suppressive init with non_target_fddt_value
    m3 = pkg.DiCoWForConditionalGeneration.from_pretrained("openai/whisper-tiny", use_fddt=True)
    assert m3.config.d_model == 384 and m3._load_report["missing"] is None


def test_product_retrieve_segment_matches_reference_golden():
    from tests.test_oracle_vs_golden import _check_retrieve
    from ts_asr_whisper_amd.generation import retrieve_segment
    _check_retrieve(retrieve_segment)


def test_product_fix_timestamps_matches_reference_golden_and_oracle():
    """Integer-tick folding of recording-time segments into 30 s windows: exact against golden F17, and against the oracle's
    Decimal restatement on a few thousand further random recordings (block boundaries, exact 30 s segments, long gaps)."""
    from tests.test_oracle_vs_golden import _f17_cases
    from oracle.longform import fold_segments, folded_to_ids
    from ts_asr_whisper_amd.generation import fix_timestamps_from_segmentation
    recs, wants = [], []
    for c, segs, want, ts0, fill, pad in _f17_cases():
        assert fix_timestamps_from_segmentation([segs], ts0, fill, pad)[0].tolist() == want, c
        recs.append(segs), wants.append(want)
    batch = fix_timestamps_from_segmentation(recs[:9], ts0, fill, pad, prefix_ids=(3, 4), suffix_ids=(5,))     # padding / specials
    for row, want in zip(batch.tolist(), wants[:9]):
        assert row[:len(want) + 3] == [3, 4] + want + [5] and all(x == pad for x in row[len(want) + 3:])
    rng = np.random.default_rng(170)
    for _ in range(3000):
        t, segs = int(rng.integers(0, 3000)) * int(rng.random() < 0.6), []
        for _ in range(int(rng.integers(1, 10))):
            if rng.random() < 0.3:
                t += int(rng.integers(0, 5000))
            if rng.random() < 0.2:
                t = max(-(-t // 1500) * 1500 - int(rng.integers(0, 3)), 0)
            r = rng.random()
            d = 1500 if r < 0.25 else (-(-(t + 1) // 1500) * 1500 - t if r < 0.4 else int(rng.integers(1, 1600)))
            odd = 0.01 if rng.random() < 0.1 else 0.0
            segs.append(dict(start=round(t * 0.02 + odd, 2), end=round((t + d) * 0.02 + odd, 2),
                             tokens=[1003] + [int(x) for x in rng.integers(10, 900, 2)] + [1040]))
            t += d
        want = folded_to_ids(fold_segments(segs, 1000, 7), 1000)
        assert fix_timestamps_from_segmentation([segs], 1000, 7, 0)[0].tolist() == want, segs
    assert fix_timestamps_from_segmentation([[]], 1000, 7, 0).shape == (1, 0)


def test_product_temperature_fallback_matches_transformers_golden():
    """generation.decode_with_fallback / token_compression_ratio / sequence_avg_logprob against golden F18: transformers' own
    generate_with_fallback (what reference generation.py:567-611 delegates to) driven with scripted decoder outputs."""
    from tests.test_oracle_vs_golden import _check_fallback
    from ts_asr_whisper_amd.generation import decode_with_fallback, token_compression_ratio, sequence_avg_logprob
    _check_fallback(decode_with_fallback, token_compression_ratio, sequence_avg_logprob)


def test_fallback_skip_flag_indexing_default_is_transformers_quirk_and_skip_by_row_fixes_it():
    """Three windows: window 0 fails the log-probability test once and is fine at the next temperature, window 1 decodes
    fine, window 2 fails it and turns out to be silence when decoded again.  In the second round the sub-batch is [0, 2]:
    transformers (hence the reference, generation.py:567-611) writes window 2's should_skip at position 1 of `should_skip`
    -- window 1's slot -- and golden F18 pins that; skip_by_row=True keeps the flag with window 2."""
    import torch
    from ts_asr_whisper_amd.generation import decode_with_fallback
    V, eos, pad = 16, 15, 14
    def scores_for(tokens, good):
        sc = torch.full((len(tokens), V), -8.0)
        for i, t in enumerate(tokens):
            sc[i, t] = 8.0 if good else -7.9
        return sc
    good = {(0, 0): False, (1, 0): True, (2, 0): False, (0, 1): True, (2, 1): False}
    def decode(rows, temp):
        k = 0 if temp == 0.0 else 1
        toks = [[3 + r, 5, 7, eos] for r in rows]
        return toks, [scores_for(t, good[(r, k)]) for r, t in zip(rows, toks)], [0.99 if (r == 2 and k == 1) else 0.1 for r in rows]
    kw = dict(compression_ratio_threshold=None, logprob_threshold=-1.0, no_speech_threshold=0.6)
    final, skip, used = decode_with_fallback(decode, 3, (0.0, 0.2), V, pad, eos, **kw)
    assert used == [1, 0, 1] and skip == [False, True, False]            # the quirk: flag of window 2 sits in window 1's slot
    final2, skip2, used2 = decode_with_fallback(decode, 3, (0.0, 0.2), V, pad, eos, skip_by_row=True, **kw)
    assert used2 == used and final2 == final and skip2 == [False, False, True]


def test_graft_entry_exposes_build_and_smoke():
    import __graft_entry__ as g
    assert callable(g.build) and callable(g.smoke)


def test_no_environment_knobs_on_the_product_dispatch_path():
    """Tuning knobs read with getenv() exist only inside `#ifdef DICOW_ABLATIONS` regions of the kernels' host code (diagnostic
    builds of tools/build_*variants.sh); the shipped library's dispatch depends on its arguments alone."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for path in sorted(glob.glob(os.path.join(root, "ts-asr-whisper_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "ts-asr-whisper_amd", "csrc", "*.inc"))):
        depth_abl, stack = 0, []
        for n, line in enumerate(open(path), 1):
            t = line.strip()
            if re.match(r"#\s*if", t):
                stack.append("DICOW_ABLATIONS" in t and not t.startswith("#ifndef"))
            elif re.match(r"#\s*else", t) and stack:
                stack[-1] = False
            elif re.match(r"#\s*endif", t) and stack:
                stack.pop()
            if "getenv(" in t and not t.startswith("//"):
                assert any(stack), f"{os.path.basename(path)}:{n}: getenv outside DICOW_ABLATIONS: {t}"


def test_logmel_folded_tables_equal_the_plain_dft():
    """features._tables: rows 400.. of the DFT tables (the product folded about sample 200 that dicow_logmel reads, ABI 5) give the same
    spectrum as the plain rows 0..399 -- x[n] + x[400 - n] against the cos rows, x[n] - x[400 - n] against the sin rows -- on random
    frames, in float64 (the table algebra, independent of the kernel)."""
    import numpy as np
    import torch
    from ts_asr_whisper_amd import features as F
    tw_c, tw_s, fb, rng = (t.numpy().astype(np.float64) for t in F._tables(80, torch.device("cpu")))
    assert tw_c.shape == (F.N_FFT + F.FOLD_ROWS, F.TABLE_LD) and tw_s.shape == tw_c.shape
    x = np.random.default_rng(0).standard_normal((7, F.N_FFT + 1))          # sample 400 belongs to the next hop: meets zero rows
    n = np.arange(F.FOLD_ROWS)
    xs, xd = x[:, n] + x[:, F.N_FFT - n], x[:, n] - x[:, F.N_FFT - n]
    re_f, im_f = xs @ tw_c[F.N_FFT:], xd @ tw_s[F.N_FFT:]
    re_p, im_p = x[:, :F.N_FFT] @ tw_c[:F.N_FFT], x[:, :F.N_FFT] @ tw_s[:F.N_FFT]
    scale = np.abs(re_p).max()
    assert np.abs(re_f - re_p).max() < 2e-6 * scale and np.abs(im_f - im_p).max() < 2e-6 * scale
    assert not tw_c[F.N_FFT + 201:].any() and not tw_s[F.N_FFT + 200:].any() and not tw_c[:, 201:].any()
    # ... and the plain rows are the windowed DFT itself
    w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(F.N_FFT) / F.N_FFT)
    ref = np.fft.rfft(x[:, :F.N_FFT] * w, axis=1)
    assert np.abs(re_p[:, :201] - ref.real).max() < 2e-6 * scale and np.abs(im_p[:, :201] - ref.imag).max() < 2e-6 * scale
