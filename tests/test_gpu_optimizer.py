"""Fused AdamW + clipping + schedule + staged freezing (SURVEY 8 row f1 and the A10 freezing schedule) against the
reference's procedure run with torch.optim.AdamW on the CPU (oracle/optim.py).  Run with `pytest -m gpu`."""
import pytest
import torch

import amd_pkg
from oracle.optim import ReferenceHarness

pytestmark = pytest.mark.gpu

pkg = amd_pkg.load()


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ts_asr_whisper_amd import ops as _ops
    return _ops


def test_adamw_kernel_matches_torch_adamw(ops):
    """dicow_sumsq_f32 + dicow_adamw_f32 on a flat region vs torch.optim.AdamW + clip_grad_norm_, 5 steps, with weight
    decay, gradients large enough for the clip to act and a region length that is not a multiple of the vector width."""
    n = 1_000_003
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    p, m, v = p0.cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    gn = torch.zeros(1, device="cuda")
    for step in range(1, 6):
        grad = torch.randn(n, generator=g) * (0.01 if step == 3 else 1.0)        # step 3: norm below max_norm -> no clip
        ref.grad = grad.clone()
        torch.nn.utils.clip_grad_norm_([ref], 5.0)
        opt.step()
        gd = grad.cuda()
        gn.zero_()
        ops.sumsq(gd, gn)
        assert abs(float(gn) - float(grad.double().pow(2).sum())) < 1e-4 * float(gn)
        ops.adamw(p, gd, m, v, 3e-3, 0.9, 0.999, 1e-8, 0.05, step, gnorm_sq=gn, max_norm=5.0)
        assert float((p.cpu() - ref.detach()).abs().max()) < 2e-6, step
    st = opt.state[ref]
    assert float((m.cpu() - st["exp_avg"]).abs().max()) < 1e-6 and float((v.cpu() - st["exp_avg_sq"]).abs().max()) < 1e-6


def test_train_step_update_rule_with_staged_freezing():
    """TrainStep's optimizer side (FlatStore, two groups, cosine warm-up schedule, clip, per-run bias correction, preheat
    phase of 2 steps) fed with the same injected gradients as the reference procedure: parameters must agree after
    every step, frozen ones must not move, and the trainable set must switch at the same step."""
    from ts_asr_whisper_amd.trainer import TrainStep
    cfg = pkg.DiCoWConfig.preset("whisper-tiny", use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True,
                                 fddt_init="suppressive", non_target_fddt_value=0.5)
    torch.manual_seed(0)
    model = pkg.DiCoWForConditionalGeneration(cfg).cuda()
    model.tie_weights()
    kw = dict(lr=1e-3, fddt_lr_multiplier=10.0, weight_decay=0.01, max_grad_norm=1.0, warmup_steps=3, max_steps=10,
              frozen_keywords=("decoder",), preheat_prefixes=("model.encoder.fddts", "model.encoder.initial_fddt"),
              use_fddt_only_n_steps=2)
    start = {n: p.detach().cpu().clone() for n, p in model.named_parameters()}
    ts = TrainStep(model, **kw)
    ref = ReferenceHarness(start, **kw)
    named = dict(model.named_parameters())
    g = torch.Generator().manual_seed(1)
    for step in range(6):
        ts.begin_step()
        ref.begin_step()
        names = [n for n, p in named.items() if p.requires_grad]
        assert names == ref.trainable(), f"step {step}: trainable sets differ"
        assert ts.warmup_phase == (step < 2)
        if step < 2:
            assert all(n.startswith(kw["preheat_prefixes"]) for n in names) and names
        else:
            assert any("layers.0.fc1" in n for n in names) and not any("decoder" in n for n in names)
        grads = {}
        for n in names:
            grads[n] = torch.randn(named[n].shape, generator=g) * (0.05 if "fddt" in n else 0.002)
            named[n].grad.copy_(grads[n])                     # the flat-store view the engine accumulates into
        ts.finish_step()
        ref.step(grads)
        worst = max(float((named[n].detach().cpu() - ref.p[n].detach()).abs().max()) for n in named)
        assert worst < 2e-6, f"step {step}: {worst}"
    moved = {n for n in named if not torch.equal(named[n].detach().cpu(), start[n])}
    assert not any("decoder" in n for n in moved) and "model.encoder.layers.0.fc1.weight" in moved
    # per-run bias correction: the encoder weights were first updated at global step 3
    assert sorted(set(ts.opt.run_t)) == [4, 6]
