"""engine.SPLIT_FWD: the encoder's FORWARD as two half batches on two HIP streams.  Training: the halves write the halves of the same
full-batch activation buffers and the backward runs as one full batch on one stream; inference: the two halves run the whole forward
side by side and write the halves of the output.  Every forward kernel is row-parallel, so NOTHING may change: encoder output, logits,
loss and every parameter gradient are bit-equal to the one-stream forward, run to run."""
import pytest
import torch

import amd_pkg

pytestmark = pytest.mark.gpu
pkg = amd_pkg.load()


def _model_and_batch(B, se=False):
    from ts_asr_whisper_amd.data import synthetic_batch
    cfg = pkg.DiCoWConfig.preset("whisper-tiny", use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True, fddt_init="suppressive",
                                 non_target_fddt_value=0.5, **(dict(use_enrollments=True, scb_layers=2) if se else {}))
    torch.manual_seed(3)
    model = pkg.DiCoWForConditionalGeneration(cfg).cuda()
    with torch.no_grad():                                       # the suppressive init leaves some FDDT classes at exactly zero: move them
        for n, p in model.named_parameters():
            if "fddt" in n:
                p.add_(0.05 * torch.randn_like(p))
    return model, synthetic_batch(cfg, B, 24, seed=5, enrollments=se)


def _step(model, batch):
    model.zero_grad(set_to_none=True)
    out = model(**batch)
    out.loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    return out.encoder_last_hidden_state.detach().clone(), out.logits.detach().clone(), out.loss.detach().clone(), grads


@pytest.mark.parametrize("B,se", [(2, False), (4, False), (6, False), (2, True), (4, True)])
def test_split_forward_is_bit_equal_to_the_one_stream_forward(B, se, monkeypatch):
    """se: SE-DiCoW (B mixture + enrollment pairs, two speaker-communication layers): the fork comes behind the last of them, where the
    enrollment rows are dropped, over the plain layers that follow."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ts_asr_whisper_amd import engine
    model, batch = _model_and_batch(B, se=se)
    monkeypatch.setattr(engine, "SPLIT_FWD_MIN_ROWS", 0)
    monkeypatch.setattr(engine, "SPLIT_BWD", False)             # (this test: the forward alone; the split backward has its own below)
    monkeypatch.setattr(engine, "SPLIT_FWD", False)
    enc0, lg0, loss0, g0 = _step(model, batch)
    # the decoder's gradients pass through atomically accumulated sums (embedding scatter-add, column sums: the order of the additions
    # follows the workgroup schedule -- tests/test_gpu_fullsize.py says the same of two one-stream runs): those are held to rounding.
    # Everything the ENCODER produces and receives is fixed-order arithmetic and must not move by a bit.
    stable = [n for n in g0 if n.startswith("model.encoder.")]
    assert len(stable) > 20
    monkeypatch.setattr(engine, "SPLIT_FWD", True)
    n_side = len(engine._FWD_STREAMS)
    for rep in range(3):                                        # (races between the two streams would show as run-to-run differences)
        enc1, lg1, loss1, g1 = _step(model, batch)
        assert len(engine._FWD_STREAMS) >= max(n_side, 1)       # the side stream exists: the split path ran
        assert torch.equal(enc1, enc0) and torch.equal(lg1, lg0) and torch.equal(loss1, loss0), rep
        assert g1.keys() == g0.keys() and len(g0) > 20
        for n in g0:
            if n in stable:
                assert torch.equal(g1[n], g0[n]), (rep, n)
            else:
                assert torch.allclose(g1[n], g0[n], rtol=1e-3, atol=1e-6), (rep, n)
    torch.cuda.synchronize()


@pytest.mark.parametrize("B", [2, 4])
def test_frozen_decoder_runs_as_two_halves_too_and_nothing_moves(B, monkeypatch):
    """The reference's default recipe freezes the decoder (keyword "decoder"): then neither direction of the decoder reduces anything
    over rows, and engine.SPLIT_DEC runs its layers -- forward and backward -- as two half batches on two streams; the LM head and the
    loss stay on the full batch.  Logits, loss and every (encoder) gradient are bit-equal to the one-stream step."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ts_asr_whisper_amd import engine
    from ts_asr_whisper_amd.trainer import freeze_by_keyword
    model, batch = _model_and_batch(B)
    model.tie_weights()
    freeze_by_keyword(model, ("decoder",))
    assert not any(p.requires_grad for p in model.model.decoder.parameters()) and not model.proj_out.weight.requires_grad
    batch["labels"][B - 1, 10:] = -100
    monkeypatch.setattr(engine, "SPLIT_FWD_MIN_ROWS", 0)
    monkeypatch.setattr(engine, "SPLIT_BWD", False)
    monkeypatch.setattr(engine, "SPLIT_FWD", False)
    enc0, lg0, loss0, g0 = _step(model, batch)
    assert len(g0) > 20 and all(n.startswith("model.encoder.") for n in g0)
    seen = []
    real = engine.DecoderEngine._layers_fwd
    monkeypatch.setattr(engine.DecoderEngine, "_layers_fwd", lambda self, enc_bf, Bp, *a: (seen.append(Bp), real(self, enc_bf, Bp, *a))[1])
    monkeypatch.setattr(engine, "SPLIT_FWD", True)
    for dec_on in (True, False, True):
        monkeypatch.setattr(engine, "SPLIT_DEC", dec_on)
        del seen[:]
        enc1, lg1, loss1, g1 = _step(model, batch)
        assert seen == ([B // 2, B // 2] if dec_on else [B]), seen
        assert torch.equal(enc1, enc0) and torch.equal(lg1, lg0) and torch.equal(loss1, loss0), dec_on
        assert g1.keys() == g0.keys()
        for n in g0:
            assert torch.equal(g1[n], g0[n]), (dec_on, n)
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,wgrad,variant", [(2, "main", ""), (4, "main", ""), (6, "alt", ""), (4, "alt", ""), (4, "alt", "preheat"), (4, "alt", "se"), (2, "main", "se"), (4, "third", ""), (6, "third", "se")])
def test_split_backward_matrices_bit_equal_vectors_to_rounding_and_bit_reproducible(B, wgrad, variant, monkeypatch):
    """engine.SPLIT_BWD: the layers' backward chain as two half batches on two streams.  Weight MATRICES come from the same pooled
    full-batch launch as before (bit-equal), the gradient that flows on to the stem is row-parallel (bit-equal: conv weights, initial
    FDDT), the layers' VECTOR gradients are sums in two pieces added in a fixed order (equal to rounding, bit-reproducible run to run)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ts_asr_whisper_amd import engine
    model, batch = _model_and_batch(B, se=variant == "se")      # se: SE-DiCoW -- the split covers the layers above the speaker-communication blocks
    if variant == "preheat":                                    # the recipe's first phase: only the FDDT parameters train (no weight-gradient GEMM runs)
        for n, p in model.named_parameters():
            p.requires_grad_("fddt" in n)
    monkeypatch.setattr(engine, "SPLIT_FWD_MIN_ROWS", 0)
    monkeypatch.setattr(engine, "SPLIT_FWD", True)
    monkeypatch.setattr(engine, "SPLIT_BWD_WGRAD", wgrad)
    monkeypatch.setattr(engine, "SPLIT_BWD", False)
    enc0, lg0, loss0, g0 = _step(model, batch)
    monkeypatch.setattr(engine, "SPLIT_BWD", True)
    ran = []
    real = engine.EncoderEngine._backward_split
    monkeypatch.setattr(engine.EncoderEngine, "_backward_split", lambda self, *a: (ran.append(1), real(self, *a))[1])
    runs = [_step(model, batch) for _ in range(3)]
    assert len(ran) == 3
    named = dict(model.named_parameters())
    n_vec = n_mat = 0
    for enc1, lg1, loss1, g1 in runs:
        assert torch.equal(enc1, enc0) and torch.equal(lg1, lg0) and torch.equal(loss1, loss0)
        for n in g0:
            if not n.startswith("model.encoder."):
                assert torch.allclose(g1[n], g0[n], rtol=1e-3, atol=1e-6), n            # (decoder: atomically accumulated sums)
            elif n.startswith(("model.encoder.layers.", "model.encoder.fddts.")) and named[n].dim() == 1:
                ref = g0[n].double()
                assert float((g1[n].double() - ref).norm()) <= 2e-4 * float(ref.norm()) + 1e-12, (n, float((g1[n].double() - ref).norm()), float(ref.norm()))
                assert torch.equal(g1[n], runs[0][3][n]), n                               # run to run: not a bit moves
                n_vec += 1
            else:                                                                        # matrices, stem, initial FDDT, final LayerNorm
                assert torch.equal(g1[n], g0[n]), n
                n_mat += 1
    assert n_vec >= 3 * (8 if variant else 20) and n_mat >= 3 * (2 if variant == "preheat" else 20), (n_vec, n_mat)
    torch.cuda.synchronize()


@pytest.mark.parametrize("B", [2, 6])
def test_split_inference_forward_is_bit_equal_to_the_one_stream_forward(B, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ts_asr_whisper_amd import engine
    model, batch = _model_and_batch(B)
    model.eval()
    monkeypatch.setattr(engine, "SPLIT_FWD_MIN_ROWS", 0)
    outs = []
    with torch.no_grad():
        for on in (False, True, True, True):
            monkeypatch.setattr(engine, "SPLIT_FWD", on)
            o = model(**batch)
            outs.append((o.encoder_last_hidden_state.clone(), o.logits.clone(), o.loss.clone()))
    for e, lg, ls in outs[1:]:
        assert torch.equal(e, outs[0][0]) and torch.equal(lg, outs[0][1]) and torch.equal(ls, outs[0][2])
    torch.cuda.synchronize()


def test_split_forward_leaves_odd_batches_small_batches_and_se_inference_alone(monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ts_asr_whisper_amd import engine
    monkeypatch.setattr(engine, "SPLIT_FWD_MIN_ROWS", 0)
    monkeypatch.setattr(engine, "SPLIT_FWD", True)
    monkeypatch.setattr(engine, "SPLIT_BWD", False)             # (count the forward's forks only)
    calls = []
    real = engine.fwd_side_stream
    monkeypatch.setattr(engine, "fwd_side_stream", lambda dev, k=0: (calls.append(1), real(dev, k))[1])
    model, batch = _model_and_batch(3)                          # odd batch: one stream (training and inference)
    _step(model, batch)
    with torch.no_grad():
        model(**batch)
    assert not calls
    model, batch = _model_and_batch(3, se=True)                 # SE-DiCoW with an odd number of pairs: one stream
    _step(model, batch)
    assert not calls
    model, batch = _model_and_batch(2, se=True)                 # SE-DiCoW inference: one stream (only the training forward forks behind the SCB layers)
    with torch.no_grad():
        model(**batch)
    assert not calls
    monkeypatch.setattr(engine, "SPLIT_FWD_MIN_ROWS", 16000)    # the shipped threshold: a small batch stays on one stream
    model, batch = _model_and_batch(4)
    _step(model, batch)
    assert not calls
    monkeypatch.setattr(engine, "SPLIT_FWD_MIN_ROWS", 0)        # and the plain forward does split, training and inference
    _step(model, batch)
    assert len(calls) == 1
    with torch.no_grad():
        model(**batch)
    assert len(calls) == 2


def test_split_backward_falls_back_to_one_stream_when_its_buffers_would_not_fit(monkeypatch):
    """The split backward keeps every layer's gradient buffers until the pass ends; a batch whose buffers exceed the allowed share of the free
    memory (engine.SPLIT_BWD_MEM_FRACTION) takes the one-stream backward -- bit-equal to SPLIT_BWD off."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ts_asr_whisper_amd import engine
    model, batch = _model_and_batch(4)
    monkeypatch.setattr(engine, "SPLIT_FWD_MIN_ROWS", 0)
    monkeypatch.setattr(engine, "SPLIT_FWD", True)
    monkeypatch.setattr(engine, "SPLIT_BWD", False)
    enc0, lg0, loss0, g0 = _step(model, batch)
    monkeypatch.setattr(engine, "SPLIT_BWD", True)
    ran = []
    real = engine.EncoderEngine._backward_split
    monkeypatch.setattr(engine.EncoderEngine, "_backward_split", lambda self, *a: (ran.append(1), real(self, *a))[1])
    monkeypatch.setattr(engine, "SPLIT_BWD_MEM_FRACTION", 0.0)
    enc1, lg1, loss1, g1 = _step(model, batch)
    assert not ran
    for n in g0:
        if n.startswith("model.encoder."):
            assert torch.equal(g1[n], g0[n]), n
    monkeypatch.setattr(engine, "SPLIT_BWD_MEM_FRACTION", 0.6)
    _step(model, batch)
    assert len(ran) == 1
