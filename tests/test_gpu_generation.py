"""Decoding path: KV-cached greedy decoder and STNO seek windows on the MI355X vs the CPU oracle and
golden F12.  Run with `pytest -m gpu`."""
import numpy as np
import pytest
import torch

import amd_pkg
from oracle import dicow_oracle as O
from oracle import generation as ogen
from tests.util import load_golden, golden_cfg, golden_params, T
from tests.test_gpu_model import build_model

pytestmark = pytest.mark.gpu
amd_pkg.load()


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ts_asr_whisper_amd as p
    return p


def test_stno_seek_windows_on_gpu_match_reference():
    from ts_asr_whisper_amd.generation import stno_seek_windows
    z = load_golden("f12_seek")
    for i in range(int(z["n_cases"])):
        got = stno_seek_windows(torch.from_numpy(z[f"c{i}.stno"]).cuda(), z[f"c{i}.seek"], z[f"c{i}.max_frames"], z[f"c{i}.map"],
                                num_frames=int(z[f"c{i}.msp"]))
        assert np.array_equal(got.cpu().numpy(), z[f"c{i}.out"]), i


def _setup(pkg):
    z = load_golden("f7_e2e_small")
    model, cfg = build_model(pkg, z, requires_grad=False)
    model.eval()
    x, st = T(z, "x"), T(z, "stno")
    prompt = torch.tensor([[cfg.decoder_start_token_id, 7, 9], [cfg.decoder_start_token_id, 7, 11]])
    return z, model, cfg, x, st, prompt


def test_greedy_decode_scores_and_tokens_vs_oracle(pkg):
    """Per step: the cached single-token decoder step reproduces the oracle's full-prefix forward (bf16-path tolerance), and
    the chosen token is the oracle's argmax up to that tolerance; suppressed ids are never produced."""
    from ts_asr_whisper_amd.generation import GreedyDecoder
    z, model, cfg, x, st, prompt = _setup(pkg)
    sup, bsup, n_new = [3, 4, 5], [20, 21], 10
    seq, scores = GreedyDecoder(model).generate(x.cuda(), st.cuda(), prompt, n_new, eos_token_id=-1, pad_token_id=cfg.pad_token_id,
                                                suppress_tokens=sup, begin_suppress_tokens=bsup, return_scores=True)
    seq, scores = seq.cpu(), scores.float().cpu()
    assert seq.shape == (2, prompt.shape[1] + n_new) and torch.equal(seq[:, :3], prompt)
    assert not any(int(t) in sup for t in seq[:, 3:].flatten()) and not any(int(t) in bsup for t in seq[:, 3])
    ocfg, p = golden_cfg(z), golden_params(z)
    with torch.no_grad():
        enc = O.encoder_forward(p, ocfg, x, st, emu=True)
        full = O.linear(O.decoder_forward(p, ocfg, seq[:, :-1], enc, emu=True), p["proj_out.weight"], None, True).float()
    tol = 6e-2
    for n in range(n_new):
        want = full[:, prompt.shape[1] - 1 + n].clone()
        want[:, sup] = -float("inf")
        if n == 0:
            want[:, bsup] = -float("inf")
        fin = torch.isfinite(want)
        assert torch.equal(fin, torch.isfinite(scores[n]))
        assert float((scores[n][fin] - want[fin]).abs().max()) < tol, n
        chosen = want.gather(1, seq[:, prompt.shape[1] + n][:, None])[:, 0]
        assert bool((chosen >= want.max(-1).values - 2 * tol).all()), n
    # the oracle's own greedy loop (no cache) picks the same tokens wherever its top-2 margin exceeds the tolerance
    oseq, oscores = ogen.greedy_decode(p, ocfg, x, st, prompt, n_new, -1, cfg.pad_token_id, sup, bsup, emu=True)
    top2 = oscores.topk(2, dim=-1).values
    if float((top2[..., 0] - top2[..., 1]).min()) > 2 * tol:
        assert torch.equal(oseq, seq)


def test_kv_cache_matches_teacher_forced_forward(pkg):
    """Incremental decoding == the training path's full causal forward on the same tokens (both on the GPU kernels)."""
    from ts_asr_whisper_amd.generation import GreedyDecoder
    z, model, cfg, x, st, prompt = _setup(pkg)
    seq, scores = GreedyDecoder(model).generate(x.cuda(), st.cuda(), prompt, 8, eos_token_id=-1, return_scores=True)
    with torch.no_grad():
        full = model(input_features=x.cuda(), stno_mask=st.cuda(), decoder_input_ids=seq[:, :-1]).logits.float()
    for n in range(8):
        assert float((scores[n] - full[:, prompt.shape[1] - 1 + n]).abs().max()) < 4e-2, n


def test_eos_stops_a_row_and_pads_it(pkg):
    from ts_asr_whisper_amd.generation import GreedyDecoder
    z, model, cfg, x, st, prompt = _setup(pkg)
    dec = GreedyDecoder(model)
    free = dec.generate(x.cuda(), st.cuda(), prompt, 6, eos_token_id=-1, pad_token_id=499).cpu()
    P = prompt.shape[1]
    eos = int(free[0, P])                                      # row 0's first token becomes the eos id
    out = dec.generate(x.cuda(), st.cuda(), prompt, 6, eos_token_id=eos, pad_token_id=499).cpu()
    assert int(out[0, P]) == eos and bool((out[0, P + 1:] == 499).all())
    first = (free[1, P:] == eos).nonzero()
    stop = int(first[0]) + 1 if len(first) else free.shape[1] - P
    assert torch.equal(out[1, :P + stop], free[1, :P + stop]) and bool((out[1, P + stop:] == 499).all())
    assert out.shape[1] == P + max(1, stop)                     # decoding ends once every row is finished
    with pytest.raises(ValueError):
        dec.generate(x.cuda(), st.cuda(), prompt, cfg.max_target_positions, eos_token_id=-1)


def test_generate_full_size_shapes(pkg):
    """whisper-base dims, 30 s windows, B=4: encoder once + 24 cached steps run and stay finite."""
    from ts_asr_whisper_amd.generation import GreedyDecoder
    from ts_asr_whisper_amd.data import synthetic_batch
    cfg = pkg.DiCoWConfig.preset("whisper-base", use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True,
                                 fddt_init="suppressive", non_target_fddt_value=0.5)
    torch.manual_seed(0)
    model = pkg.DiCoWForConditionalGeneration(cfg).cuda().eval()
    model.tie_weights()
    b = synthetic_batch(cfg, 4, 8, seed=2)
    prompt = torch.full((4, 4), cfg.decoder_start_token_id, dtype=torch.long)
    seq, scores = GreedyDecoder(model).generate(b["input_features"], b["stno_mask"], prompt, 24, eos_token_id=-1, return_scores=True)
    assert seq.shape == (4, 28) and bool(torch.isfinite(scores).all())


def test_timestamp_rules_vs_reference_and_oracle(pkg):
    """dicow_whisper_timestamp_rules vs golden F14 (the reference processor) -- masks and surviving scores bit-exact -- and vs
    the oracle at the real vocabulary size with long generated suffixes."""
    from ts_asr_whisper_amd.generation import timestamp_rules
    from oracle.timestamp_rules import timestamp_rules as oracle_rules
    z = load_golden("f14_timestamp_rules")
    V, eos, no_ts, ts0, begin = (int(v) for v in z["cfg"])
    for i in range(int(z["n_cases"])):
        mi = int(z[f"c{i}.max_init"])
        got = timestamp_rules(torch.from_numpy(z[f"c{i}.ids"]).cuda(), torch.from_numpy(z[f"c{i}.scores"]).cuda(), begin, eos, no_ts,
                              None if mi < 0 else mi).cpu().numpy()
        assert np.array_equal(got, z[f"c{i}.out"]), i
    g = torch.Generator().manual_seed(9)
    V, eos, no_ts, begin, B = 51866, 50257, 50364, 4, 16
    for L_new, boost in ((0, 0.0), (1, 0.0), (40, 0.0), (41, 8.0), (200, 0.0)):
        ids = torch.randint(0, 50000, (B, begin + L_new), generator=g)
        for b in range(B):                                   # sprinkle timestamp pairs, some rows end on one or two timestamps
            for j in range(begin + 2, begin + L_new - 1, 9):
                ids[b, j] = ids[b, j + 1] = no_ts + 1 + j
            if L_new and b % 3 == 0:
                ids[b, -1] = no_ts + 300
            if L_new > 1 and b % 6 == 0:
                ids[b, -2] = no_ts + 300
        sc = torch.randn(B, 51968, generator=g)[:, :V] * 3          # a padded row, like the decoder's logits view
        sc[:, no_ts + 1:] += boost
        want = oracle_rules(ids.numpy(), sc.numpy(), begin, eos, no_ts, 50)
        got = timestamp_rules(ids.cuda(), sc.cuda(), begin, eos, no_ts, 50).cpu().numpy()
        assert np.array_equal(got, want), (L_new, boost)


def test_greedy_decode_with_timestamp_rules(pkg):
    """Sequences produced under the rules obey them: first token a timestamp (or eos), timestamps in pairs and non-decreasing."""
    from ts_asr_whisper_amd.generation import GreedyDecoder
    z, model, cfg, x, st, prompt = _setup(pkg)
    no_ts, eos = 399, 5
    seq = GreedyDecoder(model).generate(x.cuda(), st.cuda(), prompt, 12, eos_token_id=eos, pad_token_id=cfg.pad_token_id,
                                        timestamps=dict(no_timestamps_token_id=no_ts, max_initial_timestamp_index=20)).cpu()
    for row in seq[:, prompt.shape[1]:].tolist():
        assert row[0] == eos or no_ts < row[0] <= no_ts + 1 + 20
        stamps = [t for t in row if t > no_ts]
        assert stamps == sorted(stamps) and no_ts not in row


def test_graph_captured_steps_match_eager(pkg):
    """use_graphs=True: the first window captures one graph per position, the second window replays them on new data."""
    from ts_asr_whisper_amd.generation import GreedyDecoder
    z, model, cfg, x, st, prompt = _setup(pkg)
    eager, graphed = GreedyDecoder(model), GreedyDecoder(model, use_graphs=True)
    x2, st2 = x.flip(0).contiguous(), st.flip(0).contiguous()
    for xi, si in ((x, st), (x2, st2), (x, st)):
        a, sa = eager.generate(xi.cuda(), si.cuda(), prompt, 9, eos_token_id=-1, return_scores=True)
        b, sb = graphed.generate(xi.cuda(), si.cuda(), prompt, 9, eos_token_id=-1, return_scores=True)
        assert torch.equal(a, b) and torch.equal(sa, sb)
    assert len(graphed._persist[2].graphs) == prompt.shape[1] - 1 + 9


def test_model_generate_wrapper(pkg):
    """model.generate(...) with an HF-style generation_config equals the explicit GreedyDecoder call; unsupported modes say so."""
    import types
    from ts_asr_whisper_amd.generation import GreedyDecoder
    z, model, cfg, x, st, prompt = _setup(pkg)
    gc = types.SimpleNamespace(eos_token_id=5, pad_token_id=cfg.pad_token_id, suppress_tokens=[3, 4], begin_suppress_tokens=[20],
                               return_timestamps=True, no_timestamps_token_id=399, max_initial_timestamp_index=20, max_length=14,
                               decoder_start_token_id=cfg.decoder_start_token_id, ctc_weight=0.0, num_beams=1)
    model.tokenizer = types.SimpleNamespace(prefix_tokens=[cfg.decoder_start_token_id, 7, 9])
    a = model.generate(input_features=x.cuda(), stno_mask=st.cuda(), generation_config=gc)
    want_prompt = torch.tensor([[cfg.decoder_start_token_id, 7, 9]] * 2)
    b = GreedyDecoder(model).generate(x.cuda(), st.cuda(), want_prompt, 11, eos_token_id=5, pad_token_id=cfg.pad_token_id,
                                      suppress_tokens=[3, 4], begin_suppress_tokens=[20],
                                      timestamps=dict(no_timestamps_token_id=399, max_initial_timestamp_index=20))
    assert torch.equal(a, b) and a.shape[1] <= 14
    c = model.generate(input_features=x.cuda(), stno_mask=st.cuda(), generation_config=gc, num_beams=3)
    assert c.shape[0] == 2 and c.shape[1] <= 14 and torch.equal(c[:, :3].cpu(), want_prompt)
    with pytest.raises(ValueError):
        model.generate(input_features=x.cuda()[:, :, :100], stno_mask=st.cuda(), generation_config=gc)      # shorter than a window
    model.tokenizer = None


def test_beam_search_vs_oracle(pkg):
    """GPU beam search (KV caches reordered by beam_idx) vs the oracle's bookkeeping driven by the oracle decoder (full-prefix
    forward, no cache): same best hypothesis and score up to the bf16 path's tolerance; the returned score is the length-
    penalised sum of the teacher-forced log-probabilities of the returned sequence."""
    from ts_asr_whisper_amd.generation import GreedyDecoder
    from oracle.beam_search import beam_search as oracle_beam
    z, model, cfg, x, st, prompt = _setup(pkg)
    ocfg, p = golden_cfg(z), golden_params(z)
    K, max_length, eos, sup = 3, 11, 5, [3, 4]
    seq, score = GreedyDecoder(model).beam_search(x.cuda(), st.cuda(), prompt, max_length, K, eos_token_id=eos, pad_token_id=499,
                                                  suppress_tokens=sup)
    seq, score = seq.cpu(), score.cpu()
    with torch.no_grad():
        enc = O.encoder_forward(p, ocfg, x, st, emu=True)

        def score_fn(flat):
            ids = torch.from_numpy(flat)
            enc_rep = enc.repeat_interleave(K, dim=0)
            lg = O.linear(O.decoder_forward(p, ocfg, ids, enc_rep, emu=True)[:, -1], p["proj_out.weight"], None, True).float()
            lp = torch.log_softmax(lg, -1)
            lp[:, sup] = -float("inf")
            return lp.numpy()

        oseq, oscore = oracle_beam(score_fn, prompt.numpy(), K, cfg.vocab_size, max_length, eos)
        # the returned score must be the returned sequence's own (teacher-forced) score
        P = prompt.shape[1]
        for b in range(seq.shape[0]):
            row = seq[b].tolist()
            n = len(row) - P
            while n > 1 and row[P + n - 1] == 499:
                n -= 1
            ids = torch.tensor([row[:P + n]])
            lg = O.linear(O.decoder_forward(p, ocfg, ids[:, :-1], enc[b:b + 1], emu=True), p["proj_out.weight"], None, True).float()
            lp = torch.log_softmax(lg, -1)
            lp[..., sup] = -float("inf")
            tot = sum(float(lp[0, P - 1 + j, row[P + j]]) for j in range(n))
            assert abs(tot / n - float(score[b])) < 3e-2, (b, tot / n, float(score[b]))
    assert float((score - torch.from_numpy(oscore)).abs().max()) < 5e-2
    # identical hypotheses whenever the oracle's winner beats its runner-up by more than the tolerance (checked via the scores)
    assert seq.shape[1] == oseq.shape[1] or abs(float(score.min()) - float(oscore.min())) < 5e-2


def test_beam_search_with_ctc_and_timestamps_runs(pkg):
    """Beam 3 with the whole processor chain on the CTC golden model: rules hold and scores are finite / ordered."""
    import ts_asr_whisper_amd as pkg_
    from ts_asr_whisper_amd.generation import GreedyDecoder
    z = load_golden("f10_ctc")
    model, cfg = build_model(pkg_, z, requires_grad=False)
    model.eval()
    x, st = T(z, "x").cuda(), T(z, "stno").cuda()
    ts0 = int(z["ts_start"])
    prompt = torch.tensor([[cfg.decoder_start_token_id, 7]] * x.shape[0])
    seq, score = GreedyDecoder(model).beam_search(x, st, prompt, 10, 3, eos_token_id=5, pad_token_id=cfg.pad_token_id,
                                                  timestamps=dict(no_timestamps_token_id=ts0 - 1, max_initial_timestamp_index=10),
                                                  ctc=dict(weight=0.2, first_timestamp=ts0, upper_cased=[(3, 13)], prefix_len=2, n_score=12))
    assert bool(torch.isfinite(score).all()) and seq.shape[0] == x.shape[0] and seq.shape[1] <= 10
    for row in seq[:, 2:].tolist():
        assert row[0] == 5 or ts0 <= row[0] <= ts0 + 10


def test_long_form_loop(pkg):
    """Three recordings of different lengths (3.2, 1.4 and 2.0 windows of the small golden model's 2 s window): the loop
    terminates, every window's tokens equal a direct generate() on that window, seeks advance by retrieve_segment's offsets,
    segment times are non-decreasing and stay inside the recording."""
    from ts_asr_whisper_amd.generation import LongFormDecoder, GreedyDecoder, stno_seek_windows, retrieve_segment
    z, model, cfg, x, st, prompt = _setup(pkg)
    W = 2 * cfg.max_source_positions                      # 200 feature frames = one window of the small model
    g = torch.Generator().manual_seed(21)
    B, total = 3, 3 * W + 40
    feats = torch.randn(B, cfg.num_mel_bins, total, generator=g).clamp_(-1.5, 1.5).cuda()
    stno = torch.softmax(torch.randn(B, 4, total // 2, generator=g) * 2, 1).cuda()
    max_frames = [total, W + 80, 2 * W]
    no_ts, eos = 399, 5
    p1 = prompt[:1]
    lf = LongFormDecoder(model)
    segs = lf.transcribe(feats, stno, max_frames, p1, no_ts, eos_token_id=eos, pad_token_id=499, max_new_tokens=10)
    assert len(segs) == B
    for b in range(B):
        times = [(s["start"], s["end"]) for s in segs[b]]
        assert all(s <= e + 1e-9 for s, e in times) and all(times[i][0] <= times[i + 1][0] + 1e-9 for i in range(len(times) - 1))
        assert all(e <= max_frames[b] * 0.01 + W * 0.01 + 1e-6 for _, e in times)     # a random model may stamp the window's end
    # replay recording 0 window by window with the building blocks
    dec, seek, replay = GreedyDecoder(model), 0, []
    while seek < max_frames[0]:
        left = min(max_frames[0] - seek, W)
        win = feats.new_zeros(1, cfg.num_mel_bins, W)
        win[0, :, :left] = feats[0, :, seek:seek + left]
        sw = stno_seek_windows(stno, [seek, 0, 0], max_frames, [0], num_frames=cfg.max_source_positions)
        out = dec.generate(win, sw, p1, 10, eos_token_id=eos, pad_token_id=499,
                           timestamps=dict(no_timestamps_token_id=no_ts, max_initial_timestamp_index=50))
        toks = out[0, p1.shape[1]:].tolist()
        while toks and toks[-1] in (499, eos):
            toks.pop()
        if not toks:
            seek += left
            continue
        sg, adv = retrieve_segment(toks, seek * 0.01, no_ts + 1, left)
        replay.extend(sg)
        seek += adv
    assert [s["tokens"] for s in replay] == [s["tokens"] for s in segs[0]]
    assert all(abs(a["start"] - b["start"]) < 1e-9 for a, b in zip(replay, segs[0]))
    # beam search variant runs through the same loop
    segs_b = LongFormDecoder(model, num_beams=2).transcribe(feats[:1], stno[:1], max_frames[:1], p1, no_ts, eos_token_id=eos,
                                                            pad_token_id=499, max_new_tokens=8)
    assert len(segs_b) == 1


def test_greedy_decode_at_turbo_dims_vs_oracle(pkg):
    """whisper-large-v3-turbo dimensions, one 30 s window, 3 cached decoder steps (skinny weight-streaming GEMMs at the real N / K,
    single-query attention over 1500 encoder frames): per-step scores vs the oracle's full-prefix forward."""
    from ts_asr_whisper_amd.generation import GreedyDecoder
    from oracle.dicow_oracle import OracleConfig
    cfg = pkg.DiCoWConfig.preset("whisper-large-v3-turbo", use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True,
                                 fddt_init="suppressive", non_target_fddt_value=0.5)
    torch.manual_seed(0)
    model = pkg.DiCoWForConditionalGeneration(cfg).eval()
    state = {n: t.detach().clone() for n, t in model.state_dict().items()}
    model = model.cuda()
    model.tie_weights()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, cfg.num_mel_bins, 3000, generator=g).clamp_(-1.5, 1.5)
    st = torch.softmax(torch.randn(1, 4, 1500, generator=g) * 2, 1)
    prompt = torch.tensor([[50258, 50259, 50360]])
    seq, scores = GreedyDecoder(model).generate(x.cuda(), st.cuda(), prompt, 3, eos_token_id=-1, return_scores=True)
    ocfg = OracleConfig(**{k: v for k, v in cfg.to_dict().items() if k in OracleConfig.__dataclass_fields__})
    p = dict(state)
    p["proj_out.weight"] = p["model.decoder.embed_tokens.weight"]
    with torch.no_grad():
        enc = O.encoder_forward(p, ocfg, x, st, emu=True)
        full = O.linear(O.decoder_forward(p, ocfg, seq[:, :-1].cpu(), enc, emu=True), p["proj_out.weight"], None, True).float()
    for n in range(3):
        want = full[:, prompt.shape[1] - 1 + n]
        got = scores[n].float().cpu()
        assert float((got - want).abs().max()) < 0.12, n
        assert float(want.gather(1, seq[:, prompt.shape[1] + n].cpu()[:, None])) >= float(want.max()) - 0.25


def test_detect_language_vs_oracle(pkg):
    """One decoder position on the start token, non-language logits masked (reference generation.py:151-221): logits within the
    bf16-path tolerance of the oracle, the chosen language is the oracle's wherever its margin exceeds that tolerance, and
    model.detect_language reuses the STNO mask kept by generate()."""
    from types import SimpleNamespace
    from ts_asr_whisper_amd.generation import GreedyDecoder
    z, model, cfg, x, st, prompt = _setup(pkg)
    langs = [30, 31, 32, 33, 40, 41]
    got, logits = GreedyDecoder(model).detect_language(x.cuda(), st.cuda(), langs, return_logits=True)
    ocfg, p = golden_cfg(z), golden_params(z)
    want, ologits = ogen.detect_language(p, ocfg, x, st, cfg.decoder_start_token_id, langs, emu=True)
    tol = 6e-2
    assert float((logits.float().cpu() - ologits).abs().max()) < tol
    assert all(int(g) in langs for g in got)
    sub = ologits[:, langs]
    top2 = sub.topk(2, dim=-1).values
    for b in range(x.shape[0]):
        if float(top2[b, 0] - top2[b, 1]) > 2 * tol:
            assert int(got[b]) == int(want[b])
        assert float(sub[b].max() - ologits[b, int(got[b])]) <= 2 * tol
    gc = SimpleNamespace(lang_to_id={f"<|l{i}|>": t for i, t in enumerate(langs)}, decoder_start_token_id=cfg.decoder_start_token_id,
                         eos_token_id=cfg.eos_token_id, pad_token_id=cfg.pad_token_id)
    model.generate(input_features=x.cuda(), stno_mask=st.cuda(), decoder_input_ids=prompt, max_new_tokens=2, generation_config=gc)
    again = model.detect_language(input_features=x.cuda(), generation_config=gc)
    assert torch.equal(again.cpu(), got.cpu())
    with pytest.raises(Exception):
        GreedyDecoder(model).detect_language(x, st, langs)                      # CPU tensors: no fallback


def test_model_generate_long_form_returns_window_relative_sequences(pkg):
    """model.generate on recordings longer than one window = LongFormDecoder.transcribe + fix_timestamps_from_segmentation
    (what the reference's generate() returns for long-form input, generation.py:536-564): padded ids with the tokenizer's prefix /
    eos, segments kept in model.last_segments; short-form inputs still take the single-window path."""
    from types import SimpleNamespace
    from ts_asr_whisper_amd.generation import LongFormDecoder, fix_timestamps_from_segmentation
    z, model, cfg, x, st, prompt = _setup(pkg)
    W = 2 * cfg.max_source_positions
    g = torch.Generator().manual_seed(33)
    B, total = 2, 2 * W + 60
    feats = torch.randn(B, cfg.num_mel_bins, total, generator=g).clamp_(-1.5, 1.5).cuda()
    stno = torch.softmax(torch.randn(B, 4, total // 2, generator=g) * 2, 1).cuda()
    amask = torch.zeros(B, total, dtype=torch.long)
    amask[0, :total] = 1
    amask[1, :W + 30] = 1
    no_ts, eos, pad = 399, 5, 499

    class Tok:
        prefix_tokens = [int(t) for t in prompt[0]]
        pad_token_id = pad

        def get_vocab(self):
            return {"<|0.00|>": no_ts + 1, "Ġ": 7}

    model.set_tokenizer(Tok())
    gc = SimpleNamespace(eos_token_id=eos, pad_token_id=pad, no_timestamps_token_id=no_ts, max_initial_timestamp_index=50,
                         decoder_start_token_id=cfg.decoder_start_token_id)
    out = model.generate(input_features=feats, stno_mask=stno, attention_mask=amask.cuda(), generation_config=gc, max_new_tokens=10)
    segs = LongFormDecoder(model).transcribe(feats, stno, amask.sum(-1).tolist(), prompt[:1], no_ts, eos_token_id=eos,
                                             pad_token_id=pad, max_new_tokens=10)
    assert [[s["tokens"] for s in r] for r in model.last_segments] == [[s["tokens"] for s in r] for r in segs]
    want = fix_timestamps_from_segmentation(segs, no_ts + 1, 7, pad, prefix_ids=Tok.prefix_tokens, suffix_ids=[eos])
    assert out.is_cuda and torch.equal(out.cpu(), want)
    assert out.shape[0] == B and all(int(v) == eos for v in [r[r != pad][-1] for r in out.cpu()])
    with pytest.raises(ValueError):
        model.generate(input_features=feats, stno_mask=stno, generation_config=gc)          # no attention_mask
    short = model.generate(input_features=feats[:, :, :W].contiguous(), stno_mask=stno[:, :, :W // 2].contiguous(),
                           decoder_input_ids=prompt, max_new_tokens=3, generation_config=gc)
    assert short.shape == (B, prompt.shape[1] + 3)
    model.tokenizer = None


def test_generate_builds_the_prompt_from_language_and_task(pkg):
    """generation_config.language None (what the reference sets for multilingual models, containers.py:58) -> per-row language
    detection, then the task token and <|notimestamps|>; explicit language forms; the resulting sequences equal a generate() with
    that prompt given explicitly."""
    from types import SimpleNamespace
    z, model, cfg, x, st, prompt = _setup(pkg)
    model.tokenizer = None
    langs = {"<|en|>": 30, "<|de|>": 31, "<|cs|>": 32}
    gc = SimpleNamespace(lang_to_id=langs, task_to_id={"transcribe": 40, "translate": 41}, language=None, task="transcribe",
                         no_timestamps_token_id=399, return_timestamps=False, decoder_start_token_id=cfg.decoder_start_token_id,
                         eos_token_id=-1, pad_token_id=cfg.pad_token_id)
    xc, sc = x.cuda(), st.cuda()
    det = model.detect_language(xc, sc, gc).tolist()
    init = model.retrieve_init_tokens(xc, sc, gc)
    assert init.tolist() == [[cfg.decoder_start_token_id, l, 40, 399] for l in det]
    auto = model.generate(input_features=xc, stno_mask=sc, generation_config=gc, max_new_tokens=4)
    manual = model.generate(input_features=xc, stno_mask=sc, generation_config=gc, max_new_tokens=4, decoder_input_ids=init)
    assert torch.equal(auto, manual) and auto.shape == (x.shape[0], 8)
    graphed = model.generate(input_features=xc, stno_mask=sc, generation_config=gc, max_new_tokens=4, use_graphs=True)
    again = model.generate(input_features=xc, stno_mask=sc, generation_config=gc, max_new_tokens=4, use_graphs=True)    # replay
    assert torch.equal(graphed, auto) and torch.equal(again, auto)
    for form, want in (("<|de|>", 31), ("de", 31), ("German", 31), (["cs", "en"], None)):
        gc.language = form
        got = model.retrieve_init_tokens(xc, sc, gc)[:, 1].tolist()
        assert got == ([want] * 2 if want is not None else [32, 30])
    gc.language, gc.task, gc.return_timestamps = "en", None, True
    assert model.retrieve_init_tokens(xc, sc, gc).tolist() == [[cfg.decoder_start_token_id, 30, 40]] * 2   # transcribe by default
    gc.language = "xx"
    with pytest.raises(ValueError):
        model.retrieve_init_tokens(xc, sc, gc)


def test_temperature_fallback_on_the_decoder(pkg):
    """generate_with_fallback on the real KV-cached decoder (control flow pinned by golden F18 in the CPU / host tests):
      * thresholds nothing can fail -> one greedy pass, identical to generate();
      * a log-probability threshold nothing can meet -> every window climbs the whole ladder, sampling is reproducible with a
        seeded generator, and the per-window average log-probability / compression ratio computed on the GPU scores agree with
        the oracle's restatement of transformers' formulas;
      * with the no-speech test armed and a threshold of 0, unlikely windows are skipped instead of re-decoded."""
    from ts_asr_whisper_amd.generation import GreedyDecoder, sequence_avg_logprob, token_compression_ratio
    from oracle import fallback as OF
    z, model, cfg, x, st, prompt = _setup(pkg)
    dec = GreedyDecoder(model)
    n_new, eos = 8, cfg.eos_token_id
    kw = dict(eos_token_id=eos, pad_token_id=cfg.pad_token_id)
    plain = dec.generate(x.cuda(), st.cuda(), prompt, n_new, **kw)[:, prompt.shape[1]:].tolist()
    fin, skip, used = dec.generate_with_fallback(x.cuda(), st.cuda(), prompt, n_new, temperatures=(0.0, 0.5, 1.0),
                                                 compression_ratio_threshold=1e9, logprob_threshold=-1e9, **kw)
    assert used == [0, 0] and skip == [False, False]
    for r in range(2):
        want = list(plain[r])
        while want and want[-1] == cfg.pad_token_id and cfg.pad_token_id != eos:
            want.pop()
        want = OF.strip_padding(want, cfg.pad_token_id, eos)
        if want and want[-1] == eos:
            want = want[:-1]
        assert fin[r] == want
    # nothing passes: the ladder is climbed to its end, reproducibly
    runs = []
    for _ in range(2):
        g = torch.Generator(device="cuda").manual_seed(5)
        runs.append(dec.generate_with_fallback(x.cuda(), st.cuda(), prompt, n_new, temperatures=(0.0, 0.7, 1.3),
                                               compression_ratio_threshold=None, logprob_threshold=1.0, generator=g, **kw))
    assert runs[0] == runs[1] and runs[0][2] == [2, 2] and runs[0][1] == [False, False]
    # the statistics the decisions rest on, from real scores, vs the oracle's formulas
    g = torch.Generator(device="cuda").manual_seed(7)
    seqs, scores = dec.generate(x.cuda(), st.cuda(), prompt, n_new, return_scores=True, temperature=0.7, generator=g, **kw)
    for r in range(2):
        toks = OF.strip_padding(seqs[r, prompt.shape[1]:].tolist(), cfg.pad_token_id, eos)
        lp = sequence_avg_logprob(scores[:, r], toks, 0.7)
        assert abs(lp - OF.avg_logprob(scores[:, r].float().cpu(), toks, 0.7)) < 1e-4
        assert abs(token_compression_ratio(toks, cfg.vocab_size) - OF.compression_ratio(toks, cfg.vocab_size)) < 1e-12
        assert lp < 0.0
    # silence: unlikely (threshold 1.0) and no_speech_prob > 0 -> skipped at the first temperature, never re-decoded
    fin, skip, used = dec.generate_with_fallback(x.cuda(), st.cuda(), prompt, n_new, temperatures=(0.0, 0.7),
                                                 compression_ratio_threshold=None, logprob_threshold=1.0, no_speech_threshold=0.0,
                                                 no_speech_token_id=3, **kw)
    assert skip == [True, True] and used == [0, 0]
    assert dec.no_speech_prob is not None and dec.no_speech_prob.shape == (2,) and bool((dec.no_speech_prob > 0).all())


def test_generate_end_to_end_vs_transformers_whisper_generate(pkg):
    """Golden F19 (tests/golden/make_golden_generate.py): transformers' own WhisperForConditionalGeneration.generate -- the code the
    reference's generate() (generation.py:536-564) inherits -- run end to end on a small plain-Whisper model with hashed weights;
    DiCoW with FDDT off is that model.  The product's generate() must emit the same tokens: forced <|de|> / transcribe prompt,
    suppress and begin-suppress lists, eos handling, and (case c) the per-row language detection in front of it.  Every recorded
    best-vs-second score gap of HF's fp32 run is >= 0.39; a row is followed while its gap exceeds 0.15 (bf16 scores: 6e-2)."""
    import ast
    from types import SimpleNamespace
    from tests.util import hashed_init_, hashed_mel, hashed_uniform
    z = load_golden("f19_hf_generate")
    c, gen, scale = (ast.literal_eval(str(z[k])) for k in ("cfg", "gen", "scale"))
    cfg = pkg.DiCoWConfig(use_fddt=False, **c)
    model = pkg.DiCoWForConditionalGeneration(cfg)
    hashed_init_(model)                                        # same parameter names as the HF model -> same hashed values
    d = model.model.decoder
    with torch.no_grad():                                      # the golden script's apply_scale()
        from ts_asr_whisper_amd.modeling import sinusoids
        model.model.encoder.embed_positions.weight.copy_(sinusoids(cfg.max_source_positions, cfg.d_model))
        d.embed_tokens.weight.mul_(scale["embed_tokens"]); d.embed_positions.weight.mul_(scale["embed_positions"])
        d.layer_norm.weight.mul_(scale["final_ln"])
        for l in d.layers:
            l.encoder_attn.q_proj.weight.mul_(scale["cross_q"]); l.encoder_attn.out_proj.weight.mul_(scale["cross_out"])
    model = model.cuda().eval()
    model.tie_weights()
    model.tokenizer = None
    B = z["a.seq"].shape[0]
    x = torch.from_numpy(hashed_mel(B, 80, 3000)).clone() + 0.6 * hashed_uniform(f"f19.x.{int(z['variant'])}", (B, 80, 3000))
    x[1] = x[1].flip(-1) * 0.7
    x = x.clamp(-1.5, 1.5).cuda()
    st = torch.zeros(B, 4, 1500, device="cuda"); st[:, 1] = 1.0

    def follow(ours, ref, gaps, P):
        n = 0
        for b in range(B):
            for i in range(ref.shape[1]):
                if gaps[b, i] < 0.15:
                    break
                assert int(ours[b, P + i]) == int(ref[b, i]), (b, i, ours[b].tolist(), ref[b].tolist())
                n += 1
        return n

    gc = SimpleNamespace(language="de", task="transcribe", return_timestamps=False, **gen)
    out = model.generate(input_features=x, stno_mask=st, generation_config=gc, max_new_tokens=z["a.seq"].shape[1]).cpu()
    assert out[:, :4].tolist() == [[gen["decoder_start_token_id"], gen["lang_to_id"]["<|de|>"], gen["task_to_id"]["transcribe"], gen["no_timestamps_token_id"]]] * B
    assert follow(out, z["a.seq"], z["a.gaps"], 4) == z["a.seq"].size
    gc.language = None                                          # containers.py:58 -> detect per row (reference generation.py:151-221)
    assert model.detect_language(x, st, gc).cpu().tolist() == z["c.lang"].tolist()
    out = model.generate(input_features=x, stno_mask=st, generation_config=gc, max_new_tokens=z["c.seq"].shape[1]).cpu()
    assert out[:, 1].tolist() == z["c.lang"].tolist()
    assert follow(out, z["c.seq"], z["c.gaps"], 4) == z["c.seq"].size


def test_generate_one_window_with_timestamps_runs_the_seek_loop(pkg):
    """HF's generate -- which the reference's generate() calls (generation.py:558) -- runs its seek loop on every input: with
    timestamp prediction on and a tokenizer set, a ONE-window input goes through LongFormDecoder (a second pass over the tail
    when the window ends in an open timestamp) and comes back as the fix-up matrix, exactly like a long recording; without
    timestamps it stays a single pass returning prompt + tokens."""
    from types import SimpleNamespace
    from ts_asr_whisper_amd.generation import LongFormDecoder, fix_timestamps_from_segmentation
    z, model, cfg, x, st, prompt = _setup(pkg)
    W = 2 * cfg.max_source_positions
    assert x.shape[-1] == W
    no_ts, eos, pad = 399, 5, 499

    class Tok:
        prefix_tokens = [int(t) for t in prompt[0]]
        pad_token_id = pad

        def get_vocab(self):
            return {"<|0.00|>": no_ts + 1, "Ġ": 7}

    model.set_tokenizer(Tok())
    gc = SimpleNamespace(eos_token_id=eos, pad_token_id=pad, no_timestamps_token_id=no_ts, max_initial_timestamp_index=50,
                         decoder_start_token_id=cfg.decoder_start_token_id, return_timestamps=True)
    xc, sc = x.cuda(), st.cuda()
    out = model.generate(input_features=xc, stno_mask=sc, generation_config=gc, max_new_tokens=10)
    segs = LongFormDecoder(model).transcribe(xc, sc, [W] * x.shape[0], prompt[:1], no_ts, eos_token_id=eos, pad_token_id=pad, max_new_tokens=10)
    want = fix_timestamps_from_segmentation(segs, no_ts + 1, 7, pad, prefix_ids=Tok.prefix_tokens, suffix_ids=[eos])
    assert torch.equal(out.cpu(), want)
    assert [[s["tokens"] for s in r] for r in model.last_segments] == [[s["tokens"] for s in r] for r in segs]
    gc.return_timestamps = False
    plain = model.generate(input_features=xc, stno_mask=sc, generation_config=gc, decoder_input_ids=prompt, max_new_tokens=3)
    assert plain.shape == (x.shape[0], prompt.shape[1] + 3) and torch.equal(plain[:, :prompt.shape[1]].cpu(), prompt)
    model.tokenizer = None
