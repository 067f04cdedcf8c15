"""CTC prefix scoring / joint CTC-attention decoding on the MI355X (csrc/ctc_prefix.hip through the C ABI) vs golden F13
(the reference's CTCPrefixScore / CTCRescorerLogitsProcessor themselves) and the oracle.  Run with `pytest -m gpu`."""
import numpy as np
import pytest
import torch

import amd_pkg
from oracle import ctc_prefix as ocp
from tests.util import load_golden, T
from tests.test_gpu_model import build_model

pytestmark = pytest.mark.gpu
amd_pkg.load()


@pytest.fixture(scope="module")
def cd():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ts_asr_whisper_amd import ctc_decoding
    return ctc_decoding


def close_with_logzero(got, want, tol):
    """Real entries within tol; logzero entries (-1e10, where the reference adds log-probabilities to -1e10) exactly."""
    got, want = np.asarray(got), np.asarray(want)
    real = want > -1e9
    assert np.array_equal(real, got > -1e9)
    assert np.array_equal(got[~real], want[~real])
    return float(np.abs(got[real] - want[real]).max()) < tol if real.any() else True


def test_prefix_scorer_vs_reference(cd):
    z = load_golden("f13_ctc_prefix")
    x, blank, eos = torch.from_numpy(z["a.x"]).cuda(), int(z["a.blank"]), int(z["a.eos"])
    sc = cd.CtcPrefixScorer(x, blank, eos)                     # x holds log-probabilities: the normaliser is ~0
    assert float(sc.lse.abs().max()) < 1e-5
    assert close_with_logzero(sc.initial_state().cpu(), z["a.r0"], 1e-4)
    for s in range(int(z["a.steps"])):
        act = torch.from_numpy(z[f"a.{s}.active"])
        rows = act.nonzero()[:, 0]
        y, cs, dl = (torch.from_numpy(z[f"a.{s}.{k}"])[act] for k in ("y", "cs", "dl"))
        psi, r = sc(rows.cuda(), cs.cuda(), dl.cuda(), y[:, -1].cuda(), torch.from_numpy(z[f"a.{s}.r_prev"])[act].cuda())
        assert close_with_logzero(psi.cpu(), z[f"a.{s}.psi"], 2e-5), s
        assert close_with_logzero(r.cpu(), z[f"a.{s}.r"], 1e-4), s


def _rescorer(cd, z, enc_logits):
    V, ts0, eos, bos, pad, k = (int(v) for v in z["b.cfg"])
    return cd.CtcRescorer(enc_logits, V, eos, bos, ts0, [tuple(p) for p in z["b.upper"]], len(z["b.prefix"]), float(z["b.weight"]), k)


def test_rescorer_vs_reference(cd):
    """The logits processor over 5 greedy steps (a timestamp step, a forced eos, a finished row): combined scores, kept
    states and prefix scores must follow the reference's."""
    z = load_golden("f13_ctc_prefix")
    proc = _rescorer(cd, z, torch.from_numpy(z["b.enc_logits"]).cuda())
    rows = torch.arange(3).cuda()
    for s in range(int(z["b.steps"])):
        out = proc(torch.from_numpy(z[f"b.{s}.ids"]).cuda(), torch.from_numpy(z[f"b.{s}.scores"]).cuda())
        want = z[f"b.{s}.out"]
        assert close_with_logzero(np.where(out.cpu().numpy() < -1e8, -1e10, out.cpu().numpy()), np.where(want < -1e8, -1e10, want), 1e-4), s
        proc.update_state(torch.from_numpy(z[f"b.{s}.next"]).cuda(), rows)
        assert close_with_logzero(proc.state_prev.cpu(), z[f"b.{s}.state"], 1e-4), s
        assert float((proc.score_prev.cpu() - torch.from_numpy(z[f"b.{s}.score_prev"])).abs().max()) < 1e-4, s
    out = proc(torch.from_numpy(z["b.4.ids"]).cuda(), torch.from_numpy(z["b.4.scores"]).cuda()).cpu().numpy()
    want = z["b.4.out"]
    real = want > -1e8
    assert np.array_equal(real, out > -1e8) and float(np.abs(out[real] - want[real]).max()) < 1e-4
    assert float(np.abs(out[~real] / want[~real] - 1).max()) < 1e-5          # (1-w)*s + w*(-1e10 - prev): fp32 of a huge number


def test_bf16_padded_logits_and_alias(cd):
    """bf16 logits in 128-padded rows (what get_enc_logits returns) with tied columns vs the oracle on the same rounded values."""
    g = torch.Generator().manual_seed(3)
    B, Tn, V1, ld, C = 2, 50, 70, 128, 9
    buf = torch.zeros(B, Tn, ld, dtype=torch.bfloat16)
    buf[:, :, :V1] = (torch.randn(B, Tn, V1, generator=g) * 2).to(torch.bfloat16)
    alias = torch.arange(V1, dtype=torch.int32)
    alias[[11, 12]] = torch.tensor([1, 2], dtype=torch.int32)
    sc = cd.CtcPrefixScorer(buf.cuda()[:, :, :V1], V1 - 1, 60, alias)
    logits = buf[:, :, :V1].float()
    x = (logits - torch.logsumexp(logits, -1, keepdim=True)).numpy()
    x[..., 11], x[..., 12] = x[..., 1], x[..., 2]
    r0 = ocp.initial_state(x, V1 - 1)
    assert close_with_logzero(sc.initial_state().cpu(), r0, 1e-4)
    cs = torch.stack([torch.randperm(V1 - 1, generator=g)[:C] for _ in range(B)])
    cs[:, 0] = torch.tensor([11, 60])
    dl, last = torch.tensor([0, 0]), torch.tensor([V1 - 1, V1 - 1])
    psi, r = sc(torch.arange(B).cuda(), cs.cuda(), dl.cuda(), last.cuda(), torch.from_numpy(r0).cuda())
    opsi, orr = ocp.prefix_score(x, np.arange(B), cs.numpy(), dl.numpy(), last.numpy(), r0, V1 - 1, 60)
    assert close_with_logzero(psi.cpu(), opsi, 1e-4) and close_with_logzero(r.cpu(), orr, 2e-4)
    # second step from the kept state of candidate 1, with a repeated label among the candidates
    r1 = r[:, :, :, 1].contiguous()
    last2 = cs[:, 1]
    cs2 = cs.clone()
    cs2[:, 2] = last2
    psi2, r2 = sc(torch.arange(B).cuda(), cs2.cuda(), torch.tensor([1, 1]).cuda(), last2.cuda(), r1)
    opsi2, orr2 = ocp.prefix_score(x, np.arange(B), cs2.numpy(), np.array([1, 1]), last2.numpy(), orr[:, :, :, 1], V1 - 1, 60)
    assert close_with_logzero(psi2.cpu(), opsi2, 2e-4) and close_with_logzero(r2.cpu(), orr2, 3e-4)


def test_prefix_scorer_full_size_spot_checks(cd):
    """large-v3-turbo decoding sizes: B=16 hypotheses x 500 candidates x T=375 frames x 51867 labels; random pairs are checked
    against the oracle's loops, and prefix probabilities of the empty prefix obey psi(c) <= 0 and psi(eos) = log P(all blank)."""
    g = torch.Generator().manual_seed(4)
    B, Tn, V1, C = 16, 375, 51867, 500
    logits = (torch.randn(B, Tn, V1, generator=g) * 3).to(torch.bfloat16).cuda()
    eos = 50257
    sc = cd.CtcPrefixScorer(logits, V1 - 1, eos)
    r0 = sc.initial_state()
    cs = torch.stack([torch.randperm(50364, generator=g)[:C] for _ in range(B)])
    cs[:, -1] = eos
    dl, last = torch.zeros(B, dtype=torch.long), torch.full((B,), V1 - 1)
    psi, r = sc(torch.arange(B).cuda(), cs.cuda(), dl.cuda(), last.cuda(), r0)
    psi, r, r0c = psi.cpu(), r.cpu(), r0.cpu()
    assert bool(torch.isfinite(psi).all()) and float(psi.max()) <= 1e-4
    assert float((psi[:, -1] - r0c[:, -1, 1]).abs().max()) < 1e-3           # eos: the whole utterance is blank
    lf = logits.float().cpu()
    for b, c in ((0, 0), (3, 17), (15, 498), (7, 250)):
        xb = lf[b] - torch.logsumexp(lf[b], -1, keepdim=True)
        sub = torch.stack([xb[:, cs[b, c]], xb[:, V1 - 1]], dim=1).numpy()[None]            # 2-label problem: [cand, blank]
        opsi, orr = ocp.prefix_score(sub, np.array([0]), np.array([[0]]), np.array([0]), np.array([1]), r0c[b:b + 1].numpy(), 1, 99)
        assert abs(float(psi[b, c]) - float(opsi[0, 0])) < 2e-3 * max(1.0, abs(float(opsi[0, 0])))
        assert close_with_logzero(r[b, :, :, c], orr[0, :, :, 0], 2e-2)


def test_greedy_decode_with_ctc_rescoring(cd):
    """End to end on the golden CTC model: the decoder's combined scores equal (1-w) log_softmax(attention logits) + w * CTC
    term, replayed with the oracle rescorer on the model's own CTC logits and the teacher-forced attention logits."""
    import ts_asr_whisper_amd as pkg
    from ts_asr_whisper_amd.generation import GreedyDecoder
    z = load_golden("f10_ctc")
    model, cfg = build_model(pkg, z, requires_grad=False)
    model.eval()
    x, st = T(z, "x").cuda(), T(z, "stno").cuda()
    B = x.shape[0]
    ts0 = int(z["ts_start"])
    prompt = torch.tensor([[cfg.decoder_start_token_id, 7]] * B)
    eos = 5
    ctc = dict(weight=0.3, first_timestamp=ts0, upper_cased=[(3, 13)], prefix_len=2, n_score=12)
    seq, scores = GreedyDecoder(model).generate(x, st, prompt, 6, eos_token_id=eos, pad_token_id=cfg.pad_token_id, return_scores=True, ctc=ctc)
    n_steps = scores.shape[0]
    with torch.no_grad():
        att = model(input_features=x, stno_mask=st, decoder_input_ids=seq[:, :-1]).logits.float()
        enc_out = model.model.encoder(x, stno_mask=st).last_hidden_state
        enc_logits = model.get_enc_logits(enc_out).float().cpu().numpy()
    orc = ocp.CtcRescorer(enc_logits, cfg.vocab_size, eos, cfg.decoder_start_token_id, ts0, [(3, 13)], 2, 0.3, 12)
    seq_c = seq.cpu().numpy()
    for n in range(n_steps):
        a = torch.log_softmax(att[:, prompt.shape[1] - 1 + n], dim=-1).cpu().numpy()
        want = orc(seq_c[:, :prompt.shape[1] + n], a)
        got = scores[n].cpu().numpy()
        real = (want > -1e8) & (got > -1e8)
        assert real.sum() >= B * 3
        # candidate sets can differ where attention scores tie within bf16 noise: compare where both scored the label
        assert float(np.abs(got[real] - want[real]).max()) < 8e-2, n
        orc.update_state(seq_c[:, prompt.shape[1] + n], np.arange(B))


def test_prefix_scorer_edge_sizes(cd):
    """No active hypothesis, one candidate, two frames, a repeated label at the last possible frame."""
    g = torch.Generator().manual_seed(8)
    x = torch.log_softmax(torch.randn(2, 2, 6, generator=g), -1)
    sc = cd.CtcPrefixScorer(x.cuda(), 5, 4)
    r0 = sc.initial_state()
    psi, r = sc(torch.zeros(0, dtype=torch.long).cuda(), torch.zeros(0, 3, dtype=torch.long).cuda(), torch.zeros(0).cuda(),
                torch.zeros(0).cuda(), r0[:0])
    assert psi.shape == (0, 3) and r.shape == (0, 2, 2, 3)
    cs = torch.tensor([[2], [4]])
    psi, r = sc(torch.arange(2).cuda(), cs.cuda(), torch.tensor([0, 0]).cuda(), torch.tensor([5, 5]).cuda(), r0)
    opsi, orr = ocp.prefix_score(x.numpy(), np.arange(2), cs.numpy(), np.array([0, 0]), np.array([5, 5]), r0.cpu().numpy(), 5, 4)
    assert close_with_logzero(psi.cpu(), opsi, 1e-5) and close_with_logzero(r.cpu(), orr, 1e-5)
    # a 2-label prefix in 2 frames: the last frame is the only place the next label could start
    r1 = r[:, :, :, 0].contiguous()
    psi2, r2 = sc(torch.arange(2).cuda(), cs.cuda(), torch.tensor([1, 1]).cuda(), cs[:, 0].cuda(), r1)
    opsi2, orr2 = ocp.prefix_score(x.numpy(), np.arange(2), cs.numpy(), np.array([1, 1]), cs[:, 0].numpy(), orr[:, :, :, 0], 5, 4)
    assert close_with_logzero(psi2.cpu(), opsi2, 1e-5) and close_with_logzero(r2.cpu(), orr2, 1e-5)


def test_prefix_scorer_full_size_elementwise_vs_torch_loop(cd):
    """Every state and score of a full-size call (16 hypotheses x 500 candidates x 375 frames, second decoding step with a
    repeated label among the candidates) against the reference's own formulation -- a frame loop of torch ops -- on the GPU."""
    g = torch.Generator().manual_seed(6)
    B, Tn, V1, C, eos = 16, 375, 51867, 500, 50257
    logits = (torch.randn(B, Tn, V1, generator=g) * 3).to(torch.bfloat16).cuda()
    sc = cd.CtcPrefixScorer(logits, V1 - 1, eos)
    r0 = sc.initial_state()
    rows = torch.arange(B).cuda()
    cs = torch.stack([torch.randperm(50364, generator=g)[:C] for _ in range(B)]).cuda()
    cs[:, -1] = eos
    zero, blank = torch.zeros(B, dtype=torch.long).cuda(), torch.full((B,), V1 - 1).cuda()
    psi1, r1 = sc(rows, cs, zero, blank, r0)
    last = cs[:, 3].clone()
    r_prev = r1[:, :, :, 3].contiguous()
    cs2 = cs.clone()
    cs2[:, 7] = last                                           # the repeated-label rule applies to candidate 7
    psi2, r2 = sc(rows, cs2, torch.ones(B, dtype=torch.long).cuda(), last, r_prev)
    x = torch.log_softmax(logits.float(), -1)

    def torch_loop(cand, d, last_tok, rp):
        xs = torch.gather(x, 2, cand[:, None, :].expand(-1, Tn, -1))
        xb = x[..., V1 - 1]
        rsum = torch.logaddexp(rp[..., 0], rp[..., 1])
        phi = rsum[..., None].expand(-1, -1, C).clone()
        same = (cand == last_tok[:, None]) & (d > 0)[:, None]
        phi = torch.where(same[:, None, :], rp[..., 1:2].expand(-1, -1, C), phi)
        r = torch.full((B, Tn, 2, C), -1e10, device="cuda")
        r[d == 0, 0, 0] = xs[d == 0, 0]
        start = d.clamp(min=1)
        psi = r[torch.arange(B), start - 1, 0]
        mask = torch.arange(1, Tn, device="cuda")[None, :] >= d[:, None]
        psi = torch.logaddexp(psi, torch.logsumexp(torch.where(mask[..., None], phi[:, :-1] + xs[:, 1:], torch.full_like(xs[:, 1:], -1e10)), dim=1))
        for t in range(1, Tn):
            r[:, t, 0] = torch.logaddexp(r[:, t - 1, 0], phi[:, t - 1]) + xs[:, t]
            r[:, t, 1] = torch.logaddexp(r[:, t - 1, 0], r[:, t - 1, 1]) + xb[:, t][:, None]
        psi = torch.where(cand == eos, rsum[:, -1][:, None].expand(-1, C), psi)
        return psi, r

    for (psi, r), args in (((psi1, r1), (cs, zero, blank, r0)), ((psi2, r2), (cs2, torch.ones(B, dtype=torch.long).cuda(), last, r_prev))):
        wpsi, wr = torch_loop(*args)
        real = wr > -1e9
        assert torch.equal(real, r > -1e9)
        assert float((r[real] - wr[real]).abs().max()) < 3e-3 and float((psi - wpsi).abs().max()) < 3e-3
