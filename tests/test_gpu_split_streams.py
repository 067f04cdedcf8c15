"""TrainStep(split_streams=True): ONE batch as two half batches on two HIP streams (trainer.SplitSync).  What must hold:
  * the gradients are, bit for bit, those of the two halves run as micro-batches on one stream (the gradient-accumulation path of
    the HF Trainer, which tests/test_gpu_fullsize.py pins): the halves' shares are added in a fixed order, nothing races;
  * the loss is the loss of the whole batch -- with the reference's two normalisations: the hard-label loss is the mean over ALL label
    positions, padding included (modeling_dicow.py:312-321), the soft-label loss divides by the non-padding count (modeling_dicow.py:135-145);
  * run to run the split step is bit-reproducible; two optimizer steps leave the same parameters as the micro-batch form;
  * a trainable decoder (tied head) is ordered too, and the data-parallel hook sees every segment exactly once, from the FOLLOWER half
    (the half that completes the sum).
Run with `pytest -m gpu`."""
import pytest
import torch

import amd_pkg

pytestmark = pytest.mark.gpu
pkg = amd_pkg.load()


def _cfg():
    return pkg.DiCoWConfig.preset("whisper-tiny", use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True,
                                  fddt_init="suppressive", non_target_fddt_value=0.5)


def _batch(cfg, B, seed):
    from ts_asr_whisper_amd.data import synthetic_batch
    b = synthetic_batch(cfg, B, 16, seed=seed)
    b["labels"][1, 9:] = -100                       # unequal padding in the two halves
    b["labels"][B - 1, 4:] = -100
    return b


def _halves(b, order=(1, 0)):
    hb = b["labels"].shape[0] // 2
    return [{k: (v[i * hb:(i + 1) * hb] if torch.is_tensor(v) and v.dim() > 0 else v) for k, v in b.items()} for i in order]


def _build(frozen=("decoder",), soft=False, split=False):
    from ts_asr_whisper_amd.trainer import TrainStep
    cfg = _cfg()
    torch.manual_seed(0)
    model = pkg.DiCoWForConditionalGeneration(cfg).cuda()
    model.tie_weights()
    if soft:
        class Tok:
            def get_vocab(self):
                v = {f"tok{i}": i for i in range(cfg.vocab_size)}
                for j in range(40):
                    v.pop(f"tok{50364 + j}")
                    v[f"<|{0.02 * j:.2f}|>"] = 50364 + j
                return v
        model.set_tokenizer(Tok())
    ts = TrainStep(model, lr=1e-4, fddt_lr_multiplier=10.0, frozen_keywords=frozen, split_streams=split)
    return cfg, model, ts


@pytest.mark.parametrize("soft", [False, True])
def test_split_step_equals_the_two_halves_as_micro_batches_and_the_whole_batch_loss(soft):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cfg, model, ts = _build(soft=soft, split=True)
    b = _batch(cfg, 4, 7)
    if soft:
        b["labels"][0, 3] = 50364 + 5               # a timestamp label: the soft-target path has something to smear
    # whole batch, one stream
    ts.split_streams = False
    ts.begin_step()
    loss_whole = float(ts._micro(b, 1.0))
    ts.store.settle_first_writers()
    # the halves as micro-batches (leader half first), weighted as the normalisation demands
    hs = _halves(b)
    if soft:
        cnt = [float((h["labels"] != -100).sum()) for h in hs]
        w = [c / sum(cnt) for c in cnt]
    else:
        w = [0.5, 0.5]
    ts.begin_step()
    loss_acc = sum(float(ts._micro(h, wi)) * wi for h, wi in zip(hs, w))
    ts.store.settle_first_writers()
    g_acc = ts.store.grads.clone()
    # split
    ts.split_streams = True
    assert ts._can_split(b)
    runs = []
    for _ in range(2):
        ts.begin_step()
        loss_split = float(ts._micro(b, 1.0))
        ts.store.settle_first_writers()
        torch.cuda.synchronize()
        runs.append((loss_split, ts.store.grads.clone()))
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])              # reproducible
    assert torch.equal(runs[0][1], g_acc), float((runs[0][1] - g_acc).abs().max())       # == micro-batches, bit for bit
    assert abs(runs[0][0] - loss_acc) < 1e-6 * max(1.0, abs(loss_acc))
    assert abs(runs[0][0] - loss_whole) < 2e-5 * max(1.0, abs(loss_whole)), (runs[0][0], loss_whole)
    assert float(g_acc.norm()) > 0


def test_split_optimizer_steps_with_trainable_decoder_and_the_data_parallel_hook_order():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cfg, m1, ts1 = _build(frozen=(), split=True)
    _, m2, ts2 = _build(frozen=(), split=False)
    batches = [_batch(cfg, 4, 20 + i) for i in range(2)]
    seen = []
    enc = m1.model.encoder
    hook_enc, hook_dec = enc._segment_hook, m1._segment_hook

    def spy(name, inner):
        seen.append((name, ts1._split_sync.role, ts1.reducer.hold))
        inner(name)
    enc._segment_hook = lambda name: spy(name, hook_enc)
    m1._segment_hook = lambda name: spy(name, hook_dec)
    for step in range(2):
        l1 = float(ts1.step(batches[step]))
        l2 = float(ts2.step(_halves(batches[step])))          # HF-style accumulation of the two halves: what the split form reproduces
        assert abs(l1 - l2) < 1e-6 * max(1.0, abs(l2)), (step, l1, l2)
    torch.cuda.synchronize()
    for (n, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
        if "decoder.embed_" in n:                  # embedding gradients are atomic scatter-adds (as torch's index_add): not bit-stable in ANY form
            assert torch.allclose(p.detach(), q.detach(), rtol=1e-5, atol=1e-7), n
        else:
            assert torch.equal(p.detach(), q.detach()), n
    # every segment is announced by both halves; only the follower's announcement may let a bucket leave (hold is False there)
    names = {n for n, _, _ in seen}
    assert {"decoder", "final_ln", "stem", "layer0", f"layer{len(enc.layers) - 1}"} <= names
    for name in names:
        roles = [(r, h) for n, r, h in seen if n == name]
        assert len(roles) == 4 and roles[0] == ("lead", True) and roles[1] == ("follow", False), (name, roles)
    # a batch the split form does not cover (odd size) takes the one-stream path
    odd = {k: (v[:3] if torch.is_tensor(v) and v.dim() > 0 else v) for k, v in batches[0].items()}
    assert not ts1._can_split(odd)
    assert float(ts1.step(odd)) > 0
