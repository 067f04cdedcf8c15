"""Replica consistency at start-up (what DistributedDataParallel's constructor gives the reference for free: rank 0's parameters
and buffers on every rank -- scripts/submit_slurm.sh:34, configs/base.yaml:73 via the HF Trainer's DDP wrap), the optimizer
checkpoint's layout fingerprint, first-writer gradient hygiene, roctx ranges and the generate() temperature rule.  CPU only:
world-size-2 gloo processes with DELIBERATELY different seeds per rank."""
import os
import socket
from types import SimpleNamespace as NS

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import amd_pkg

amd_pkg.load()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(seed):
    import ts_asr_whisper_amd as pkg
    cfg = pkg.DiCoWConfig(vocab_size=256, d_model=64, encoder_layers=2, encoder_attention_heads=1, decoder_layers=1,
                          decoder_attention_heads=1, encoder_ffn_dim=128, decoder_ffn_dim=128, max_source_positions=20,
                          max_target_positions=16, pad_token_id=250, use_pre_pos_fddt=True, use_enrollments=True, scb_layers=1)
    torch.manual_seed(seed)
    m = pkg.DiCoWForConditionalGeneration(cfg)
    with torch.no_grad():                       # the default initialisation leaves some tensors identical across seeds: perturb all
        g = torch.Generator().manual_seed(seed)
        for p in m.parameters():
            p.add_(0.01 * torch.randn(p.shape, generator=g))
    m.tie_weights()
    return m


def _state(model):
    sd = {n: p.detach().clone() for n, p in model.named_parameters()}
    sd.update({"buf." + n: b.detach().clone() for n, b in model.named_buffers()})
    return sd


def _worker(rank, world, port, q, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ts_asr_whisper_amd.trainer import TrainStep
        model = _model(seed=100 + rank)                     # every rank builds a DIFFERENT model
        before = _state(model)
        try:
            ts = TrainStep(model, replica_sync=mode)
            err = None
        except RuntimeError as e:
            ts, err = None, str(e)
        after = _state(model)
        nbytes = ts.replica_sync_bytes if ts is not None else -1
        resumed = None
        if ts is not None and mode == "broadcast":
            # resume: moments differ per rank in the loaded file -> rank 0's after load_state_dict
            sd = ts.state_dict()
            sd["exp_avg"] = torch.full_like(sd["exp_avg"], float(rank + 1))
            sd["exp_avg_sq"] = torch.full_like(sd["exp_avg_sq"], float(rank + 2))
            # ... and so do the step counters of the file and (a model loaded from another file) the parameters themselves
            sd["global_step"] = 5 + rank
            sd["run_t"] = [5 + rank] * len(sd["run_t"])
            with torch.no_grad():
                ts.store.params.add_(float(rank))
            ts.load_state_dict(sd)
            resumed = (float(ts.store.exp_avg.mean()), float(ts.store.exp_avg_sq.mean()))
            resumed_more = (ts.opt.t, tuple(ts.opt.run_t), ts.opt.counters.tolist(), float(ts.store.params.double().sum()))
            q.put(("more", rank, resumed_more))
        q.put((rank, {k: v.numpy() for k, v in before.items()}, {k: v.numpy() for k, v in after.items()}, nbytes, err, resumed))
    finally:
        dist.destroy_process_group()


LAST_MORE = None


def _run(mode):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    got = []
    while len([g for g in got if g[0] != "more"]) < world:
        got.append(q.get(timeout=300))
    global LAST_MORE
    LAST_MORE = sorted([g[1:] for g in got if g[0] == "more"])
    res = [g for g in got if g[0] != "more"]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda t: t[0])
    return res


def test_trainstep_broadcasts_rank0_state_like_ddp():
    r0, r1 = _run("broadcast")
    # the ranks really started apart ...
    assert any((r0[1][k] != r1[1][k]).any() for k in r0[1])
    for k in r0[1]:
        # ... rank 0 keeps its own values bit for bit, rank 1 ends with rank 0's -- trainable (flat store), frozen decoder, buffers
        assert (r0[2][k] == r0[1][k]).all(), k
        assert (r1[2][k] == r0[1][k]).all(), k
    n_bytes = sum(v.size * v.itemsize for k, v in r0[1].items() if k != "proj_out.weight")        # tied weight: once
    assert r0[3] == r1[3] and r0[3] >= n_bytes                                               # (the flat store pads entries to 64)
    assert r0[4] is None and r1[4] is None
    assert r0[5] == (1.0, 2.0) and r1[5] == (1.0, 2.0)                                        # resumed moments: rank 0's
    # ... and the host AND device step counters and the parameters: ranks that read different files continue as one replica
    (_, m0), (_, m1) = LAST_MORE
    assert m0 == m1 and m0[0] == 5 and set(m0[1]) == {5} and m0[2][0] == 5


def test_trainstep_verify_mode_raises_on_diverged_replicas():
    r0, r1 = _run("verify")
    for r in (r0, r1):
        assert r[4] is not None and "replicas differ" in r[4] and "[1]" in r[4], r[4]
        for k in r[1]:
            assert (r[2][k] == r[1][k]).all()               # verify never overwrites


def test_sync_replicas_is_a_noop_without_a_process_group_and_rejects_unknown_modes():
    from ts_asr_whisper_amd.trainer import TrainStep, sync_replicas
    model = _model(0)
    before = _state(model)
    ts = TrainStep(model)
    assert ts.replica_sync_bytes == 0
    for k, v in _state(model).items():
        assert torch.equal(v, before[k])
    with pytest.raises(ValueError):
        sync_replicas(model, ts.store, mode="sometimes")


def test_bit_checksum_sees_a_single_flipped_bit_and_an_exchange_of_two_elements():
    from ts_asr_whisper_amd.trainer import _bit_checksum
    t = torch.randn(10000)
    c = _bit_checksum(t)
    u = t.clone()
    u.view(torch.int32)[1234] ^= 1
    assert _bit_checksum(u) != c
    w = t.clone()
    w[[5, 6]] = w[[6, 5]]
    assert _bit_checksum(w) != c and _bit_checksum(t.clone()) == c
    assert _bit_checksum(torch.zeros(0)) == 0 and _bit_checksum(t.to(torch.bfloat16)) != _bit_checksum(u.to(torch.float64))


def test_optimizer_checkpoint_carries_a_per_parameter_layout_fingerprint():
    from ts_asr_whisper_amd.trainer import TrainStep
    ts = TrainStep(_model(0))
    sd = ts.state_dict()
    names = [e[0] for e in sd["entries"]]
    assert len(names) == len(ts.store.entries) and all(isinstance(n, str) for n in names)
    offs = [e[1] for e in sd["entries"]]
    assert offs == sorted(offs)
    ts.load_state_dict(sd)                                              # round trip
    old = dict(sd)
    old.pop("entries")
    with pytest.raises(ValueError, match="fingerprint"):                # a round-3 file: same runs, unknown order inside them
        ts.load_state_dict(old)
    swapped = dict(sd)
    e = [tuple(x) for x in sd["entries"]]
    i = next(k for k in range(len(e) - 1) if e[k][2] == e[k + 1][2])    # two neighbours of equal size trade places: runs unchanged
    e[i], e[i + 1] = (e[i + 1][0], e[i][1], e[i][2]), (e[i][0], e[i + 1][1], e[i + 1][2])
    swapped["entries"] = e
    with pytest.raises(ValueError, match="different flat-buffer layout"):
        ts.load_state_dict(swapped)


def test_first_writer_zero_grad_keeps_frozen_and_unwritten_matrices_clean():
    from ts_asr_whisper_amd.trainer import TrainStep
    ts = TrainStep(_model(0))
    st = ts.store
    assert st._over
    st.grads.fill_(7.0)                                                 # "the previous step's gradients"
    frozen = st._over[0][0]
    frozen.requires_grad_(False)                                        # frozen AFTER the store was built, outside _set_phase
    st.zero_grad(first_writer=True)
    p, o, n = st._over[0]
    assert float(st.grads[o:o + n].abs().max()) == 0.0 and not p._grad_overwrite
    for a, b in st._zero_ranges:
        assert float(st.grads[a:b].abs().max()) == 0.0
    p2, o2, n2 = st._over[1]
    assert p2._grad_overwrite and float(st.grads[o2]) == 7.0            # flagged: its GEMM will overwrite the stale values ...
    assert st.settle_first_writers() == len(st._over) - 1               # ... and if no GEMM ran, the step settles them to zero
    assert float(st.grads.abs().max()) == 0.0 and not p2._grad_overwrite


def test_roctx_ranges_cover_the_four_phases_of_a_step():
    from ts_asr_whisper_amd import tracing
    from ts_asr_whisper_amd.trainer import TrainStep
    ts = TrainStep(_model(0))
    tracing.counts.clear()
    loss = torch.zeros((), requires_grad=True)
    real = ts.model
    ts.model = lambda **kw: NS(loss=loss * 1.0)                         # host logic only: no kernels on this box
    ts._micro({}, 1.0)
    ts.model = real
    ts.opt.step = lambda preheat_only=False: None
    ts.finish_step()
    assert all(tracing.counts.get(k) == 1 for k in tracing.PHASES), tracing.counts
    assert tracing.available() in (True, False)
    with tracing.range("nested"):
        with tracing.range("inner"):
            pass
    assert tracing.counts["nested"] == 1 and tracing.counts["inner"] == 1


def test_generate_ignores_the_generation_configs_scalar_temperature():
    """transformers 4.55 (the reference's pin): GenerationConfig().temperature == 1.0 by default; HF's Whisper generate takes
    `temperature` from the explicit keyword only.  Such a config must reach the decoder (here: the GPU-only refusal), an explicit
    scalar temperature > 0 is still refused as unimplemented sampling."""
    import ts_asr_whisper_amd as pkg
    from ts_asr_whisper_amd._lib import DicowError
    model = _model(0)
    cfg = model.config
    gc = NS(temperature=1.0, eos_token_id=250, pad_token_id=250)
    x, st = torch.zeros(1, cfg.num_mel_bins, 40), torch.zeros(1, 4, 20)
    with pytest.raises(DicowError, match="GPU"):
        model.generate(x, st, generation_config=gc, max_new_tokens=2)
    with pytest.raises(NotImplementedError, match="sampling"):
        model.generate(x, st, generation_config=gc, max_new_tokens=2, temperature=0.7)
    assert isinstance(pkg.DiCoWConfig, type)
