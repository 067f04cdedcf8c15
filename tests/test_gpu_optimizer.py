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
            named[n]._grad_overwrite = False                  # (what the engine's first-writer GEMM does: "this gradient was written")
        ts.finish_step()
        ref.step(grads)
        worst = max(float((named[n].detach().cpu() - ref.p[n].detach()).abs().max()) for n in named)
        assert worst < 2e-6, f"step {step}: {worst}"
    moved = {n for n in named if not torch.equal(named[n].detach().cpu(), start[n])}
    assert not any("decoder" in n for n in moved) and "model.encoder.layers.0.fc1.weight" in moved
    # per-run bias correction: the encoder weights were first updated at global step 3
    assert sorted(set(ts.opt.run_t)) == [4, 6]


def test_device_schedule_and_bias_corrections_match_torch_over_300_steps(ops):
    """dicow_adamw_hyper (schedule and 1 - beta^t evaluated in double on the device) + the fused update against
    torch.optim.AdamW driven by LambdaLR(get_cosine_schedule_with_warmup) -- the reference's optimizer / scheduler pair
    (containers.py:100-114, dicow_v3.yaml:66-67) -- over 300 steps that cross the warm-up: the applied learning rate equals
    torch's (as a float) at every step, lr_at() (logging) equals what was applied, parameters stay within 2e-6."""
    import math
    from ts_asr_whisper_amd.trainer import FusedAdamW

    class Store:                                            # the slice of FlatStore the optimizer reads
        pass

    n, steps, warm, total = 4099, 300, 40, 400
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(n, generator=g)
    st = Store()
    st.params, st.grads = p0.cuda(), torch.zeros(n, device="cuda")
    st.exp_avg, st.exp_avg_sq = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    st.runs = [(0, 2048, True), (2048, n, False)]
    opt = FusedAdamW(st, lr=1e-3, fddt_lr_multiplier=10.0, weight_decay=0.01, max_grad_norm=1e9, warmup_steps=warm, max_steps=total)
    ra, rb = torch.nn.Parameter(p0[:2048].clone()), torch.nn.Parameter(p0[2048:].clone())
    topt = torch.optim.AdamW([{"params": [ra], "lr": 1e-2, "weight_decay": 0.0}, {"params": [rb], "lr": 1e-3, "weight_decay": 0.01}],
                             betas=(0.9, 0.999), eps=1e-8)

    def lam(k):                                             # transformers.get_cosine_schedule_with_warmup
        if k < warm:
            return float(k) / float(max(1, warm))
        prog = float(k - warm) / float(max(1, total - warm))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 2.0 * 0.5 * prog)))
    sched = torch.optim.lr_scheduler.LambdaLR(topt, lam)
    worst = 0.0
    for k in range(1, steps + 1):
        grad = torch.randn(n, generator=g) * 0.1
        ra.grad, rb.grad = grad[:2048].clone(), grad[2048:].clone()
        lrs = [gr["lr"] for gr in topt.param_groups]
        topt.step()
        sched.step()
        st.grads.copy_(grad)
        opt.step()
        if k in (1, 2, 3, warm, warm + 1, 150, steps):
            applied = opt.applied_lr()
            for got, want in zip(applied, lrs):
                assert got == float(torch.tensor(want, dtype=torch.float32)), (k, got, want)
            assert float(torch.tensor(opt.lr_at(k) * 10.0, dtype=torch.float32)) == applied[0]
            hy = opt.hyper.cpu()
            assert float(hy[1, 2]) == float(torch.tensor(1.0 - 0.999 ** k, dtype=torch.float32)), k
            assert float(hy[1, 1]) == float(torch.tensor(1.0 - 0.9 ** k, dtype=torch.float32)), k
            ref = torch.cat([ra.detach(), rb.detach()])
            dev = float((st.params.cpu() - ref).abs().max())
            if k <= 3:
                assert dev < 5e-7, (k, dev)                  # step by step: last-bit agreement
            worst = max(worst, dev)
    # 300 updates of up to 1e-2 on values of a few units: fp32 rounding differences of the two update expressions random-walk
    assert worst < 3e-5, worst


def test_first_writer_gradients_equal_zero_fill_plus_accumulate():
    """FlatStore.zero_grad(first_writer=True): the encoder layers' weight-matrix gradients are WRITTEN by the first micro-batch's
    weight-gradient GEMM (no zero fill, no read of the old value) and accumulated by later micro-batches.  The flat gradient
    buffer must equal the zero-fill + accumulate path bit for bit -- after one micro-batch, after two (accumulation), and on a
    second step that starts from a buffer full of the previous step's gradients."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ts_asr_whisper_amd.trainer import TrainStep
    from ts_asr_whisper_amd.data import synthetic_batch
    cfg = pkg.DiCoWConfig.preset("whisper-tiny", use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True,
                                 fddt_init="suppressive", non_target_fddt_value=0.5)
    batches = [synthetic_batch(cfg, 2, 12, seed=60 + i) for i in range(3)]
    grads = {}
    for fw in (False, True):
        torch.manual_seed(0)
        model = pkg.DiCoWForConditionalGeneration(cfg).cuda()
        model.tie_weights()
        ts = TrainStep(model, lr=1e-4, fddt_lr_multiplier=10.0)
        ts.first_writer = fw
        snaps = []
        ts.begin_step()
        ts._micro(batches[0], 0.5)
        snaps.append(ts.store.grads.clone())
        ts._micro(batches[1], 0.5)                                   # second micro-batch: accumulates
        snaps.append(ts.store.grads.clone())
        ts.finish_step()
        ts.begin_step()                                              # the matrices still hold the previous step's gradients
        ts._micro(batches[2], 1.0)
        snaps.append(ts.store.grads.clone())
        if fw:
            assert ts.store._over and len(ts.store._zero_ranges) <= len(model.model.encoder.layers) + 2
            assert not any(getattr(p, "_grad_overwrite", False) for p, _, _ in ts.store._over)      # every flag was consumed
        grads[fw] = snaps
    for k, (a, b) in enumerate(zip(grads[False], grads[True])):
        assert torch.equal(a, b), f"snapshot {k}: first-writer gradients differ from zero-fill + accumulate"
    assert float(grads[True][0].abs().sum()) > 0
