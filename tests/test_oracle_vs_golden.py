"""Pin the oracle (oracle/*.py) against the golden fixtures generated from the real reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import stno as ostno
from oracle import logmel as ologmel
from oracle import dicow_oracle as O
from oracle import augment as oaug
from tests.util import load_golden, golden_cfg, golden_params, T, maxdiff, hashed_mel


def test_f1_stno_bit_exact():
    z = load_golden("f1_stno")
    for i in range(int(z["n_cases"])):
        got = ostno.create_stno_masks(z[f"in_{i}"].copy(), int(z[f"idx_{i}"]))
        assert got.dtype == z[f"out_{i}"].dtype
        assert np.array_equal(got, z[f"out_{i}"]), f"case {i}"
        assert np.allclose(got.sum(-1), 1.0, atol=1e-6)


def test_stno_collate_pads_with_silence():
    a = np.random.default_rng(0).random((7, 4)).astype(np.float32)
    b = np.random.default_rng(1).random((10, 4)).astype(np.float32)
    out = ostno.collate_stno([a, b])
    assert out.shape == (2, 4, 10)
    assert np.array_equal(out[0, :, :7], a.T) and np.array_equal(out[1], b.T)
    assert np.all(out[0, 0, 7:] == 1) and np.all(out[0, 1:, 7:] == 0)


def test_stno_pooling_shape():
    m = np.zeros((2, 16000 * 31), dtype=np.float32)
    m[0, :8000] = 1
    p = ostno.pool_speaker_mask(m)
    assert p.shape == (2, 3000) and p[0, :25].min() == 1.0 and p[0, 25:].max() == 0.0


def test_f2_logmel():
    z = load_golden("f2_logmel")
    for i in range(int(z["n_cases"])):
        wave = z[f"wave_{i}"].astype(np.float32) / 32768.0
        padded, am = ologmel.pad_to_30s(wave)
        got = ologmel.log_mel(padded, int(z[f"mels_{i}"]))
        ref = z[f"feat_{i}"]
        assert got.shape == ref.shape
        assert int(am.sum()) == int(z[f"attn_sum_{i}"])
        # reference computes the STFT in fp32 (torch.stft); HF itself quotes 1e-5 between its own paths
        assert np.abs(got - ref).max() < 2e-4, np.abs(got - ref).max()


def _fddt_cfg(vn):
    return O.OracleConfig(
        d_model=128, fddt_is_diagonal=not vn.startswith("full"), fddt_bias_only=(vn == "bias"),
        fddt_use_silence=(vn != "diag_no_sil_ovl"), fddt_use_overlap=(vn != "diag_no_sil_ovl"),
        fddt_use_target=(vn != "full_no_tgt"))


def test_f3_fddt_fwd_bwd():
    z = load_golden("f3_fddt")
    for vn in ["diag", "full", "bias", "diag_no_sil_ovl", "full_no_tgt"]:
        cfg = _fddt_cfg(vn)
        p = {k[len(vn) + 3:]: torch.from_numpy(z[k]).clone().requires_grad_(True)
             for k in z.files if k.startswith(vn + ".p.")}
        h = T(z, vn + ".h").requires_grad_(True)
        out = O.fddt(h, T(z, vn + ".stno"), p, "", cfg)
        out.backward(T(z, vn + ".gout"))
        assert maxdiff(out, T(z, vn + ".out")) < 2e-5, vn
        assert maxdiff(h.grad, T(z, vn + ".gh")) < 2e-5, vn
        for k, t in p.items():
            ref = T(z, f"{vn}.g.{k}")
            assert maxdiff(t.grad, ref) < 1e-4 * max(1.0, float(ref.abs().max())), (vn, k)


def test_f4_conv_stem_fwd_bwd():
    """The oracle's stem (conv1d_k3 x 2 + exact GELU, oracle/dicow_oracle.py:100-111,201-203) against the reference's own conv
    modules (encoder.py:167-170), forward and every parameter gradient; both fixture cases (toy and full 3000 frames)."""
    z = load_golden("f4_conv_stem")
    for cn in ("a", "b"):
        p = {k[len(cn) + 3:]: torch.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith(cn + ".p.")}
        x = torch.from_numpy(z[cn + ".x"]).float()
        h = O.gelu_erf(O.conv1d_k3(x, p["conv1.weight"], p["conv1.bias"], 1, False))
        out = O.gelu_erf(O.conv1d_k3(h, p["conv2.weight"], p["conv2.bias"], 2, False)).permute(0, 2, 1)
        out.backward(T(z, cn + ".gout"))
        assert maxdiff(out, T(z, cn + ".out")) < 2e-5, cn
        for k, t in p.items():
            ref = T(z, f"{cn}.g.{k}")
            assert maxdiff(t.grad, ref) < 1e-4 * max(1.0, float(ref.abs().max())), (cn, k)


def test_f6_scb():
    z = load_golden("f6_scb")
    cfg = O.OracleConfig(d_model=128, encoder_attention_heads=2, encoder_ffn_dim=256, use_enrollments=True, scb_layers=1)
    p = {"blk." + k[2:]: torch.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith("p.")}
    x = T(z, "x").requires_grad_(True)
    out = O.scb(x, p, "blk.", cfg, False)
    out.backward(T(z, "gout"))
    assert maxdiff(out, T(z, "out")) < 5e-5
    assert maxdiff(x.grad, T(z, "gx")) < 5e-5
    for k, t in p.items():
        ref = T(z, "g." + k[4:])
        assert maxdiff(t.grad, ref) < 2e-4 * max(1.0, float(ref.abs().max())), k
    with torch.no_grad():
        p["blk.cae.cross_gate.gate"].zero_()
        out0 = O.scb(x, p, "blk.", cfg, False)
    assert torch.equal(out0[0::2], x[0::2]) and torch.equal(out0[1::2], x[1::2])
    assert maxdiff(out0, T(z, "out_gate0")) == 0.0


def test_f7_e2e_small_hard_and_soft():
    z = load_golden("f7_e2e_small")
    cfg = golden_cfg(z)
    p = golden_params(z, requires_grad=True)
    x, st, lab, upp = T(z, "x"), T(z, "stno"), T(z, "labels"), T(z, "upp_labels")
    out = O.model_forward(p, cfg, x, st, lab, upp)
    assert maxdiff(out["encoder_last_hidden_state"], T(z, "enc")) < 2e-4
    assert maxdiff(out["logits"], T(z, "logits")) < 2e-4
    assert abs(float(out["loss"]) - float(z["hard.loss"])) < 1e-5
    out["loss"].backward()
    n_checked = 0
    for k in z.files:
        if k.startswith("hard.g."):
            name = k[len("hard.g."):]
            if name == "proj_out.weight":
                continue
            ref = T(z, k)
            assert maxdiff(p[name].grad, ref) < 1e-5 + 2e-3 * float(ref.abs().max()), name
            n_checked += 1
    assert n_checked > 80
    # soft-label loss with the stub timestamp vocabulary
    for t in set(p.values()):
        t.grad = None
    vocab = {f"tok{i}": i for i in range(cfg.vocab_size)}
    for j in range(int(z["ts_n"])):
        vocab.pop(f"tok{int(z['ts_start']) + j}")
        vocab[f"<|{0.02 * j:.2f}|>"] = int(z["ts_start"]) + j
    ts = O.build_ts_smoothing(vocab)
    out = O.model_forward(p, cfg, x, st, lab, upp, ts=ts)
    assert abs(float(out["loss"]) - float(z["soft.loss"])) < 1e-5
    out["loss"].backward()
    for k in z.files:
        if k.startswith("soft.g."):
            ref = T(z, k)
            assert maxdiff(p[k[len("soft.g."):]].grad, ref) < 1e-5 + 2e-3 * float(ref.abs().max()), k


def test_f7_bf16_emulation_tracks_reference_autocast():
    """F9: the oracle's bf16 emulation must deviate from fp32 no more than the reference's own
    bf16 autocast does (same order of magnitude), which justifies the GPU tolerances."""
    z = load_golden("f7_e2e_small")
    cfg = golden_cfg(z)
    p = golden_params(z)
    x, st, lab, upp = T(z, "x"), T(z, "stno"), T(z, "labels"), T(z, "upp_labels")
    with torch.no_grad():
        emu = O.model_forward(p, cfg, x, st, lab, upp, emu=True)
    ref32, ref16 = T(z, "logits"), T(z, "bf16.logits")
    dev_ref = maxdiff(ref16, ref32)
    dev_emu = maxdiff(emu["logits"], ref32)
    assert dev_emu < 3 * dev_ref + 1e-3, (dev_emu, dev_ref)
    assert abs(float(emu["loss"]) - float(z["hard.loss"])) < 3 * abs(float(z["bf16.loss"]) - float(z["hard.loss"])) + 5e-3


def test_f8_e2e_se_dicow():
    z = load_golden("f8_e2e_se")
    cfg = golden_cfg(z)
    p = golden_params(z, requires_grad=True)
    enr = {"input_features": T(z, "enr.x"), "stno_mask": T(z, "enr.stno")}
    out = O.model_forward(p, cfg, T(z, "x"), T(z, "stno"), T(z, "labels"), T(z, "upp_labels"), enrollments=enr)
    assert maxdiff(out["encoder_last_hidden_state"], T(z, "enc")) < 2e-4
    assert maxdiff(out["logits"], T(z, "logits")) < 2e-4
    assert abs(float(out["loss"]) - float(z["loss"])) < 1e-5
    out["loss"].backward()
    for k in z.files:
        if k.startswith("g."):
            ref = T(z, k)
            assert maxdiff(p[k[2:]].grad, ref) < 1e-5 + 2e-3 * float(ref.abs().max()), k


def test_f5_encoder_full_length():
    z = load_golden("f5_encoder_T1500")
    cfg = golden_cfg(z)
    p = golden_params(z)
    p["model.encoder.embed_positions.weight"] = O.sinusoids(1500, cfg.d_model)
    with torch.no_grad():
        enc = O.encoder_forward(p, cfg, T(z, "x"), T(z, "stno"))
    assert maxdiff(enc[:, :48], T(z, "enc_head")) < 3e-4
    assert maxdiff(enc[:, -48:], T(z, "enc_tail")) < 3e-4
    assert maxdiff(enc.mean(-1), T(z, "enc_mean")) < 1e-4


@pytest.mark.parametrize("fixture", ["f10_ctc", "f10b_ctc_extra_layer"])
def test_f10_ctc_branch(fixture):
    """CTC auxiliary branch (recipe default ctc_weight 0.3): extra self-attention (F10) or a full extra encoder layer (F10b),
    two stride-2 convs, lm_head, CTC loss."""
    z = load_golden(fixture)
    cfg = golden_cfg(z)
    p = golden_params(z, requires_grad=True)
    vocab = {f"tok{i}": i for i in range(cfg.vocab_size)}
    for j in range(int(z["ts_n"])):
        vocab.pop(f"tok{int(z['ts_start']) + j}")
        vocab[f"<|{0.02 * j:.2f}|>"] = int(z["ts_start"]) + j
    ts = O.build_ts_smoothing(vocab)
    out = O.model_forward(p, cfg, T(z, "x"), T(z, "stno"), T(z, "labels"), T(z, "upp_labels"), ts=ts,
                          prefix_tokens=[int(v) for v in z["prefix"]])
    if "enc_logits" in z.files:
        assert maxdiff(out["enc_logits"], T(z, "enc_logits")) < 3e-4
    else:
        assert maxdiff(out["enc_logits"][:, ::6], T(z, "enc_logits_sub")) < 3e-4
    assert abs(float(out["loss"]) - float(z["loss"])) < 2e-5
    out["loss"].backward()
    n = 0
    for k in z.files:
        if k.startswith("g."):
            ref = T(z, k)
            assert maxdiff(p[k[2:]].grad, ref) < 1e-5 + 2e-3 * float(ref.abs().max()), k
            n += 1
    assert n > 30


# ------------------------------------------------------------------------------------------------ F11: batch augmentation
def test_f11_gaussian_noise_bit_exact():
    z = load_golden("f11_augment")
    stno = z["stno"]
    for i in range(int(z["n_noise"])):
        var, frac, seed = z[f"noise_{i}_cfg"]
        torch.manual_seed(int(seed))
        got = oaug.add_gaussian_noise_and_rescale(stno.copy(), float(var), float(frac))
        want = z[f"noise_{i}_out"]
        assert np.array_equal(got, want), f"case {i}"
        touched = (got != stno).any(axis=(1, 2))
        assert touched.sum() == int(stno.shape[0] * frac)
        assert np.allclose(got.sum(1), 1.0, atol=1e-5) and got.min() >= 0


def test_f11_soft_segments_bit_exact():
    z = load_golden("f11_augment")
    stno = z["stno"]
    for i in range(int(z["n_seg"])):
        cp, lo, hi, seed = z[f"seg_{i}_cfg"]
        torch.manual_seed(int(seed))
        got = oaug.soft_segment_augmentation(stno.copy(), float(cp), int(lo), int(hi))
        assert np.array_equal(got, z[f"seg_{i}_out"]), f"case {i}"
        assert np.allclose(got.sum(1), 1.0, atol=1e-5)


def test_f11_spec_aug_joint():
    """fp32 bicubic: the restatement is within 1e-6 of the reference (ATen's accumulation order is not reproduced
    bit-for-bit); masks and the random decisions must agree exactly."""
    z = load_golden("f11_augment")
    for i in range(int(z["n_spec"])):
        M, B, Tm, seed = (int(v) for v in z[f"spec_{i}_cfg"])
        mel, stno = hashed_mel(B, M, Tm), z["stno"][:B, :, :Tm // 2]
        torch.manual_seed(seed)
        mel_out, stno_out = oaug.spec_aug_joint(mel, stno)
        want_mel, want_stno = z[f"spec_{i}_mel_out"], z[f"spec_{i}_stno_out"]
        assert np.abs(mel_out - want_mel).max() < 1e-6, f"case {i}"
        assert np.abs(stno_out - want_stno).max() < 1e-6, f"case {i}"
        assert np.array_equal(mel_out == 0, want_mel == 0), f"case {i}: mask pattern"
        assert (want_mel == 0).mean() > 0.05


def test_spec_aug_without_warp_only_masks():
    torch.manual_seed(5)
    mel, stno = hashed_mel(2, 128, 200), np.full((2, 4, 100), 0.25, np.float32)
    mo, so = oaug.spec_aug_joint(mel, stno, apply_time_warp=False)
    keep = mo != 0
    assert np.array_equal(mo[keep], mel[keep]) and np.array_equal(so, stno)      # 128 mel bins: STNO rows are never masked


# ------------------------------------------------------------------------------------------------ F12: STNO seek windows
def test_f12_stno_seek_windows_bit_exact():
    from oracle import generation as ogen
    z = load_golden("f12_seek")
    for i in range(int(z["n_cases"])):
        got = ogen.stno_seek_windows(z[f"c{i}.stno"], z[f"c{i}.seek"], z[f"c{i}.max_frames"], z[f"c{i}.map"],
                                     num_frames=int(z[f"c{i}.msp"]))
        assert np.array_equal(got, z[f"c{i}.out"]), i
        assert got.shape[0] == int(z[f"c{i}.att_rows"]) and np.allclose(got.sum(1), 1.0, atol=1e-6)


# ------------------------------------------------------------------------------------------------ F13: CTC prefix scoring
def _close_lz(got, want, tol):
    real = want > -1e9
    assert np.array_equal(real, got > -1e9) and np.array_equal(got[~real], want[~real])
    return (not real.any()) or float(np.abs(got[real] - want[real]).max()) < tol


def test_f13_ctc_prefix_scorer():
    from oracle import ctc_prefix as ocp
    z = load_golden("f13_ctc_prefix")
    x, blank, eos = z["a.x"], int(z["a.blank"]), int(z["a.eos"])
    assert _close_lz(ocp.initial_state(x, blank), z["a.r0"], 1e-4)
    for s in range(int(z["a.steps"])):
        act = z[f"a.{s}.active"]
        psi, r = ocp.prefix_score(x, np.nonzero(act)[0], z[f"a.{s}.cs"][act], z[f"a.{s}.dl"][act], z[f"a.{s}.y"][act][:, -1],
                                  z[f"a.{s}.r_prev"][act], blank, eos)
        assert _close_lz(psi, z[f"a.{s}.psi"], 5e-6) and _close_lz(r, z[f"a.{s}.r"], 5e-5), s


def test_f13_ctc_rescorer_processor():
    from oracle import ctc_prefix as ocp
    z = load_golden("f13_ctc_prefix")
    V, ts0, eos, bos, pad, k = (int(v) for v in z["b.cfg"])
    p = ocp.CtcRescorer(z["b.enc_logits"], V, eos, bos, ts0, z["b.upper"], len(z["b.prefix"]), float(z["b.weight"]), k)
    for s in range(int(z["b.steps"])):
        out = p(z[f"b.{s}.ids"], z[f"b.{s}.scores"])
        assert float(np.abs(out - z[f"b.{s}.out"]).max()) < 5e-5 * max(1.0, float(np.abs(z[f"b.{s}.out"]).max()) / 1e9), s
        p.update_state(z[f"b.{s}.next"], np.arange(3))
        assert _close_lz(p.state_prev, z[f"b.{s}.state"], 5e-5) and float(np.abs(p.score_prev - z[f"b.{s}.score_prev"]).max()) < 1e-5
    out = p(z["b.4.ids"], z["b.4.scores"])
    real = z["b.4.out"] > -1e8
    assert float(np.abs(out - z["b.4.out"])[real].max()) < 5e-5 and np.allclose(out[~real], z["b.4.out"][~real], rtol=1e-6)


# ------------------------------------------------------------------------------------------------ F14: timestamp rules
def test_f14_timestamp_rules_bit_exact():
    from oracle.timestamp_rules import timestamp_rules
    z = load_golden("f14_timestamp_rules")
    V, eos, no_ts, ts0, begin = (int(v) for v in z["cfg"])
    for i in range(int(z["n_cases"])):
        mi = int(z[f"c{i}.max_init"])
        got = timestamp_rules(z[f"c{i}.ids"], z[f"c{i}.scores"], begin, eos, no_ts, None if mi < 0 else mi)
        assert np.array_equal(got, z[f"c{i}.out"]), i


# ------------------------------------------------------------------------------------------------ F15: beam search bookkeeping
def test_f15_beam_search_bookkeeping():
    """Every step's running / finished beams, scores, flags and cache indices follow transformers' helpers in the reference's
    order (slots still holding the -1e9 placeholder are unordered in torch.topk and are not compared)."""
    from oracle.beam_search import BeamState
    z = load_golden("f15_beam_search")
    for ci in range(int(z["n_cases"])):
        B, K, V, P, ml, eos = (int(v) for v in z[f"c{ci}.cfg"])
        es = {0: False, 1: True, 2: "never"}[int(z[f"c{ci}.early_stopping"])]
        st = BeamState(z[f"c{ci}.prompt"], K, V, ml, eos, float(z[f"c{ci}.length_penalty"]), es)
        for s in range(int(z[f"c{ci}.steps"])):
            assert not st.done
            bi = st.step(z[f"c{ci}.s{s}.log_probs"])
            live = z[f"c{ci}.s{s}.running_beam_scores"] > -1e8
            assert np.array_equal(st.running_sequences[live], z[f"c{ci}.s{s}.running_sequences"][live])
            assert np.array_equal(bi.reshape(B, K)[live], z[f"c{ci}.s{s}.beam_idx"].reshape(B, K)[live])
            fin = z[f"c{ci}.s{s}.beam_scores"] > -1e8
            assert np.array_equal(fin, st.beam_scores > -1e8) and np.array_equal(st.sequences[fin], z[f"c{ci}.s{s}.sequences"][fin])
            assert np.array_equal(st.is_sent_finished, z[f"c{ci}.s{s}.is_sent_finished"]) and np.array_equal(st.unsat, z[f"c{ci}.s{s}.unsat"])
            assert np.allclose(st.running_beam_scores, z[f"c{ci}.s{s}.running_beam_scores"], rtol=1e-6, atol=1e-6)
            assert np.allclose(st.beam_scores, z[f"c{ci}.s{s}.beam_scores"], rtol=1e-6, atol=1e-6)
        assert st.done
        seq, sc = st.result()
        assert np.array_equal(seq, z[f"c{ci}.final"]) and np.allclose(sc, z[f"c{ci}.final_scores"], atol=1e-6)


# ------------------------------------------------------------------------------------------------ F16: long-form segment retrieval
def _check_retrieve(fn):
    z = load_golden("f16_retrieve_segment")
    tb = int(z["timestamp_begin"])
    for i in range(int(z["n_cases"])):
        prev, seq = int(z[f"c{i}.prev"]), z[f"c{i}.seq"].tolist()
        try:
            segs, off = fn(seq, float(z["time_offset"][prev]), tb, int(z["seek_num_frames"][prev]))
        except ValueError:
            segs, off = None, -1
        assert off == int(z[f"c{i}.offset"]), i
        assert (-1 if segs is None else len(segs)) == int(z[f"c{i}.nseg"]), i
        for j, sg in enumerate(segs or []):
            assert abs(sg["start"] - float(z[f"c{i}.s{j}.start"])) < 1e-9 and abs(sg["end"] - float(z[f"c{i}.s{j}.end"])) < 1e-9
            assert list(sg["tokens"]) == z[f"c{i}.s{j}.tokens"].tolist()


def test_f16_retrieve_segment_oracle():
    from oracle.longform import retrieve_segment
    _check_retrieve(retrieve_segment)


# ------------------------------------------------------------------------------------------------ F17: window-relative timestamps
def _f17_cases():
    z = load_golden("f17_fix_timestamps")
    for c in range(int(z["n_cases"])):
        nt = z[f"c{c}.ntok"]
        off = np.concatenate([[0], np.cumsum(nt)]).astype(int)
        tk = z[f"c{c}.tokens"]
        segs = [dict(start=float(z[f"c{c}.start"][i]), end=float(z[f"c{c}.end"][i]), tokens=[int(x) for x in tk[off[i]:off[i + 1]]])
                for i in range(len(nt))]
        yield c, segs, [int(x) for x in z[f"c{c}.ids"]], int(z["first_timestamp"]), int(z["filler"]), int(z["pad"])


def test_f17_fix_timestamps_oracle():
    """247 recordings folded by the reference's _fix_timestamps_from_segmentation (stand-in tokenizer): exact ids."""
    from oracle.longform import fold_segments, folded_to_ids
    n = 0
    for c, segs, want, ts0, fill, _ in _f17_cases():
        assert folded_to_ids(fold_segments(segs, ts0, fill), ts0) == want, c
        n += 1
    assert n > 200


# ----------------------------------------------------------------------------- F18: temperature fallback (transformers' own decisions)
def _check_fallback(loop, ratio, avglp):
    """loop(decode, n, temps, V, pad, eos, cr_thr, lp_thr, ns_thr) -> (final, skip, used); decode(rows, temp) -> (toks, scores, nsp)."""
    import ast
    z = load_golden("f18_fallback")
    V, eos, R = int(z["V"]), int(z["eos"]), int(z["n_rows"])
    temps = [float(t) for t in z["temps"]]
    thr = [float(t) for t in z["thr"]]
    for r in range(R):
        for k in range(len(temps)):
            tok, sc = z[f"tok_{r}_{k}"].tolist(), torch.from_numpy(z[f"sc_{r}_{k}"])
            assert abs(ratio(tok, V) - float(z[f"cr_{r}_{k}"])) < 1e-12
            assert abs(avglp(sc, tok, temps[k]) - float(z[f"lp_{r}_{k}"])) < 2e-5
    calls = []

    def decode(rows, temp):
        k = temps.index(temp)
        calls.append((k, list(rows), temp > 0.0, temp if temp > 0.0 else 1.0))
        toks = [z[f"tok_{r}_{k}"].tolist() for r in rows]
        n = max(len(t) for t in toks)
        return [t + [eos] * (n - len(t)) for t in toks], [torch.from_numpy(z[f"sc_{r}_{k}"]) for r in rows], torch.from_numpy(z["no_speech_prob"])[:len(rows)]

    final, skip, used = loop(decode, R, temps, V, eos, eos, thr[0], thr[1], thr[2])
    assert calls == ast.literal_eval(str(z["calls"]))                     # the same windows re-decoded at the same temperatures
    assert list(skip) == [bool(x) for x in z["should_skip"]]
    for r in range(R):
        assert final[r] == z[f"final_{r}"].tolist(), r
    assert used == [0, 2, 3, 0, 1]


def test_f18_fallback_oracle():
    from oracle import fallback as OF

    def loop(decode, n, temps, V, pad, eos, a, b, c):
        nsp = torch.from_numpy(load_golden("f18_fallback")["no_speech_prob"])
        return OF.fallback_loop(lambda rows, t: decode(rows, t)[:2], n, temps, V, pad, eos, a, b, c, nsp)
    _check_fallback(loop, OF.compression_ratio, OF.avg_logprob)


def test_f9_deviation_table_matches_the_fixtures():
    """tests/golden/F9_bf16_deviation.md (the stand-alone table of SURVEY 8c) is exactly what the committed fixtures hold."""
    import importlib.util, os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_f9_table", os.path.join(here, "make_f9_table.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with open(os.path.join(here, "F9_bf16_deviation.md")) as f:
        assert f.read() == mod.render()
