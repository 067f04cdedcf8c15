"""TrainStep(graph=True): the whole step -- zero-grad, forward, backward, clip, two-group AdamW with its schedule, bf16 weight
refresh -- captured in one hipGraph per phase and replayed must apply EXACTLY the updates of the launch-by-launch step
(reference counterpart of the step: HF Trainer.training_step, src/utils/trainers.py:116-139), across the staged-freezing phase
switch (trainers.py:122-137).  Run with `pytest -m gpu`."""
import pytest
import torch

import amd_pkg

pytestmark = pytest.mark.gpu


def _run(graph, steps, n_preheat):
    pkg = amd_pkg.load()
    from ts_asr_whisper_amd.trainer import TrainStep
    from ts_asr_whisper_amd.data import synthetic_batch
    cfg = pkg.DiCoWConfig.preset("whisper-tiny", use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True,
                                 fddt_init="suppressive", non_target_fddt_value=0.5)
    torch.manual_seed(0)
    model = pkg.DiCoWForConditionalGeneration(cfg).cuda()
    model.tie_weights()
    ts = TrainStep(model, lr=1e-4, fddt_lr_multiplier=10.0, weight_decay=0.01, max_grad_norm=1.0, warmup_steps=3, max_steps=20,
                   use_fddt_only_n_steps=n_preheat, graph=graph)
    batches = [synthetic_batch(cfg, 2, 12, seed=90 + i) for i in range(3)]
    losses, snaps = [], []
    for k in range(steps):
        losses.append(float(ts.step(batches[k % 3])))
        snaps.append(ts.store.params.detach().clone())
    return losses, snaps, ts


def test_graph_replay_is_bit_identical_to_eager_across_the_phase_switch():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    steps, n_pre = 8, 3
    l_e, p_e, _ = _run(False, steps, n_pre)
    l_g, p_g, ts = _run(True, steps, n_pre)
    assert len(ts._graphs) == 2, list(ts._graphs)                      # one captured graph per phase (preheat, full)
    assert ts.opt.t == steps and int(ts.opt.counters[0]) == steps      # device counters in lock-step with the host mirrors
    assert [int(x) for x in ts.opt.counters[1:].tolist()] == ts.opt.run_t
    for k in range(steps):
        assert torch.equal(p_e[k], p_g[k]), f"parameters differ after step {k + 1}"
        assert abs(l_e[k] - l_g[k]) <= 1e-6 * max(1.0, abs(l_e[k])), (k, l_e[k], l_g[k])
    assert not torch.equal(p_e[0], p_e[-1])                            # (training moved the parameters)
    # frozen-in-phase-1 runs were first updated at step n_pre + 1: their per-run counters say so
    pre = [bool(x) for x in ts.opt.is_pre.tolist()]
    assert all(t == steps if is_pre else t == steps - n_pre for t, is_pre in zip(ts.opt.run_t, pre))


def test_graph_mode_refuses_what_it_cannot_capture():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    pkg = amd_pkg.load()
    from ts_asr_whisper_amd.trainer import TrainStep
    from ts_asr_whisper_amd.data import synthetic_batch
    cfg = pkg.DiCoWConfig.preset("whisper-tiny", use_fddt=True, fddt_is_diagonal=True, ctc_weight=0.3)
    model = pkg.DiCoWForConditionalGeneration(cfg).cuda()
    ts = TrainStep(model, graph=True)
    with pytest.raises(NotImplementedError):
        ts.step(synthetic_batch(cfg, 1, 8, seed=1))


def test_graph_cache_is_bounded_and_evicted_signatures_are_recaptured():
    """Variable label lengths = one batch signature each.  The cache holds at most max_graphs graphs (LRU, one shared memory
    pool); a signature that was evicted is captured again, and every step still applies the eager step's update bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    pkg = amd_pkg.load()
    from ts_asr_whisper_amd.trainer import TrainStep
    from ts_asr_whisper_amd.data import synthetic_batch
    cfg = pkg.DiCoWConfig.preset("whisper-tiny", use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True,
                                 fddt_init="suppressive", non_target_fddt_value=0.5)
    lengths = [8, 12, 16]
    order = [0, 0, 1, 1, 2, 2, 0, 0, 1, 2, 2]                          # each signature: eager first, captured on its second visit
    out = {}
    for graph in (False, True):
        torch.manual_seed(0)
        model = pkg.DiCoWForConditionalGeneration(cfg).cuda()
        model.tie_weights()
        ts = TrainStep(model, lr=1e-4, fddt_lr_multiplier=10.0, graph=graph)
        ts.max_graphs = 2
        batches = [synthetic_batch(cfg, 2, L, seed=70 + L) for L in lengths]
        snaps = []
        for i in order:
            ts.step(batches[i])
            snaps.append(ts.store.params.detach().clone())
            assert len(ts._graphs) <= 2
        out[graph] = snaps
        if graph:
            assert len(ts._graphs) == 2 and ts._graph_pool is not None
            sigs = [k[1] for k in ts._graphs]
            assert ts._sig_of(batches[2]) in sigs                      # the most recently replayed signature is resident
    for k, (a, b) in enumerate(zip(out[False], out[True])):
        assert torch.equal(a, b), f"parameters differ after step {k + 1}"
