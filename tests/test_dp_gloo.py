"""Data-parallel host logic on CPU: world-size-2 gloo processes exercise FlatStore layout + GradReducer bucketing
(the same code path the RCCL run uses, minus the side stream) and check DP-vs-single-process gradient equality."""
import os
import socket

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


def _small_model():
    import ts_asr_whisper_amd as pkg
    cfg = pkg.DiCoWConfig(vocab_size=256, d_model=64, encoder_layers=3, encoder_attention_heads=1, decoder_layers=1,
                          decoder_attention_heads=1, encoder_ffn_dim=128, decoder_ffn_dim=128, max_source_positions=20,
                          max_target_positions=16, pad_token_id=250, use_pre_pos_fddt=True, use_enrollments=True, scb_layers=2)
    torch.manual_seed(0)
    return pkg.DiCoWForConditionalGeneration(cfg)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ts_asr_whisper_amd.trainer import FlatStore, GradReducer, freeze_by_keyword
        model = _small_model()
        freeze_by_keyword(model, ("decoder",))
        store = FlatStore(model, ("model.encoder.fddts", "model.encoder.initial_fddt", "model.encoder.ca_enrolls"))
        red = GradReducer(store)
        assert red.world == world and red.stream is None
        # rank-dependent "gradients"; reduce segment by segment in backward-completion order
        g = torch.Generator().manual_seed(100 + rank)
        store.grads.copy_(torch.randn(store.numel, generator=g))
        local = store.grads.clone()
        for name, a, b in store.segments:
            red.segment_ready(name)
        red.finish()
        full = store.grads.clone()
        # gradient accumulation: while `hold` is set (not the last micro-batch) nothing is exchanged
        store.grads.copy_(local)
        red.hold = True
        for name, a, b in store.segments:
            red.segment_ready(name)
        assert torch.equal(store.grads, local)
        red.hold = False
        # staged freezing, phase 1: only the preheat runs are exchanged (one coalesced all-reduce in finish())
        red.preheat_only = True
        for name, a, b in store.segments:
            red.segment_ready(name)
        red.finish()
        from types import SimpleNamespace
        from ts_asr_whisper_amd.trainer import TrainStep
        logged = float(TrainStep.logged_loss(SimpleNamespace(reducer=red), torch.tensor(1.0 + rank)))   # mean over ranks
        q.put((rank, local.numpy(), full.numpy(), [s[0] for s in store.segments], store.grads.clone().numpy(),
               [(a, b, pre) for a, b, pre in store.runs], logged))
    finally:
        dist.destroy_process_group()


def test_dp_allreduce_matches_mean_of_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda t: t[0])
    mean = torch.from_numpy(res[0][1] + res[1][1]) / 2
    assert all(abs(r[6] - 1.5) < 1e-6 for r in res)           # the logged loss is the mean of the ranks' losses (1.0, 2.0)
    for _, local, reduced, segs, pre_reduced, runs, _ in res:
        assert torch.allclose(torch.from_numpy(reduced), mean, atol=1e-6)
        want = torch.from_numpy(local).clone()
        for a, b, pre in runs:
            if pre:
                want[a:b] = mean[a:b]
        assert any(pre for _, _, pre in runs) and not all(pre for _, _, pre in runs)
        assert torch.allclose(torch.from_numpy(pre_reduced), want, atol=1e-6)
    # segments are laid out in backward-completion order: final LN, then layers N-1 .. 0, then the stem
    assert res[0][3] == ["final_ln", "layer2", "layer1", "layer0", "stem"]


def _worker_report(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ts_asr_whisper_amd.trainer import FlatStore, GradReducer, freeze_by_keyword
        model = _small_model()
        freeze_by_keyword(model, ("decoder",))
        store = FlatStore(model, ("model.encoder.fddts", "model.encoder.initial_fddt", "model.encoder.ca_enrolls"))
        red = GradReducer(store)
        red.time_buckets = True
        STEPS = 3
        # rank 1 "skipped" one flagged weight matrix (its GEMM did not run: stale values, flag still up); rank 0 wrote it
        store.zero_grad(first_writer=True)
        p_, o_, n_ = store._over[0]
        for step in range(STEPS):
            store.grads.fill_(1.0 + rank)
            for p, o, n in store._over:
                p._grad_overwrite = False
            if rank == 1:
                store.grads[o_:o_ + n_] = 123.0                   # last step's leftovers
                p_._grad_overwrite = True
            for name, a, b in store.segments:
                red.segment_ready(name)
            red.finish()
        rep = red.bucket_report(STEPS)
        q.put((rank, rep, len(store.segments), float(store.grads[o_]), float(store.grads[o_ + n_ - 1]), bool(getattr(p_, "_grad_overwrite", False)),
               red.bucket_report(STEPS)))
    finally:
        dist.destroy_process_group()


def test_bucket_report_and_per_bucket_settling_over_gloo():
    """GradReducer.bucket_report (what `bench.py --gpus N` prints per rank so that a SCALE run explains itself): buckets per step, how
    long ready buckets queued before their all-reduce started, how long the collectives took -- on the synchronous gloo path the lag
    is 0 by construction and the busy time is wall time.  And FlatStore.settle_range: a first-writer matrix ONE rank
    skipped goes on the wire as zeros, so both ranks end with the same average -- (fresh + 0) / 2 -- and the flag is consumed."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_report, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, rep, nseg, first, lastv, still, again in got:
        assert rep["buckets_per_step"] == nseg and rep["start_lag_ms"] == {"mean": 0.0, "max": 0.0, "sum_per_step": 0.0}
        assert rep["busy_ms_per_step"] > 0.0 and len(rep["longest_queued"]) == min(3, nseg)
        assert all(set(r) == {"bucket", "lag_ms", "collective_ms"} for r in rep["longest_queued"])
        assert again is None                                           # the records are consumed by the report
        assert first == 0.5 and lastv == 0.5 and not still             # rank 0's 1.0 and rank 1's zeros, averaged; not 62.0


def test_flat_store_views_and_groups():
    from ts_asr_whisper_amd.trainer import FlatStore, freeze_by_keyword
    model = _small_model()
    freeze_by_keyword(model, ("decoder",))
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    store = FlatStore(model, ("model.encoder.fddts", "model.encoder.initial_fddt", "model.encoder.ca_enrolls"))
    names = dict(model.named_parameters())
    for n, p in names.items():
        assert torch.equal(p.detach(), before[n])                       # values preserved by the re-pointing
        if "decoder" in n or n == "proj_out.weight":
            assert not p.requires_grad and getattr(p, "_direct_grad", None) is None
        else:
            assert p.requires_grad and p.grad is not None and p.grad.data_ptr() == p._direct_grad.data_ptr()
    # embed_positions becomes trainable by the keyword rule (reference containers.py:80-90)
    assert names["model.encoder.embed_positions.weight"].requires_grad
    # optimizer runs cover every trainable element exactly once, preheat runs = FDDT + SCB parameters
    covered = sum(b - a for a, b, _ in store.runs)
    assert covered == store.numel
    pre = sum(b - a for a, b, is_pre in store.runs if is_pre)
    n_pre = sum((p.numel() + 63) // 64 * 64 for n, p in names.items() if n.startswith(("model.encoder.fddts", "model.encoder.initial_fddt", "model.encoder.ca_enrolls")))
    assert pre == n_pre
    # writing through the flat buffer is visible in the parameter (same storage)
    store.params.zero_()
    assert float(names["model.encoder.layers.0.fc1.weight"].abs().sum()) == 0.0


def test_bench_launcher_spawns_the_ranks_itself():
    """`python bench.py --gpus 2` without a rendezvous in the environment starts two ranks (scripts/submit_slurm.sh:34 uses
    torchrun for this); --dry-launch stops after the process-group checks, so it runs here over gloo without a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True,
                       text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["allreduce_sum"] == d["expected_sum"] == 3.0
    assert d["config"]["global_batch"] == 32 and d["config"]["parallelism"] == "dp2"
    # a rendezvous of another size in the environment is refused instead of silently running one rank
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True,
                       text=True, timeout=120, cwd=root, env=dict(env, RANK="0", WORLD_SIZE="3"))
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr


def test_bench_launcher_returns_when_a_rank_dies_before_the_rendezvous():
    """Rank 1 exits before joining the process group: rank 0 would wait in the rendezvous for its timeout.  The launcher polls
    every child, kills the survivors on the first non-zero exit and reports the failed rank's stderr."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True,
                       text=True, timeout=300, cwd=root, env=dict(env, DICOW_BENCH_FAIL_RANK="1"))
    assert r.returncode == 1 and time.time() - t0 < 120
    assert "rank exit codes" in r.stderr and "rank 1 failing on request" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_preflight_over_gloo_and_the_exposed_wait_guard():
    """`bench.py --gpus 2 --preflight`: the dress rehearsal of a multi-GPU run (per-rank device report, environment, a timed
    all-reduce with a checked result, the bucket schedule) -- here without a GPU, over gloo: two ranks, one JSON line, exit 0.
    And the rule that fails a run whose gradient exchange is not hidden (bench.exposed_guard)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--preflight"], capture_output=True,
                       text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["preflight"] is True and d["n_gpus"] == 2 and d["backend"] == "gloo" and len(d["ranks"]) == 2
    assert [q["rank"] for q in d["ranks"]] == [0, 1]
    for q in d["ranks"]:
        assert q["allreduce_256mb"]["result_ok"] is True and q["allreduce_256mb"]["ms"] > 0 and "env" in q
    assert d["config"]["parallelism"] == "dp2" and "gradient_exchange" in d and "topology" in d
    sys.path.insert(0, root)
    import bench
    assert bench.exposed_guard([0.0, 0.1], 100.0, 0.2) is None and bench.exposed_guard([0.0, 0.1], 100.0, None) is None
    msg = bench.exposed_guard([1.0, 30.0], 100.0, 0.2)
    assert msg and "30.00 ms" in msg and "20%" in msg
    topo = "x\n==== Link Type between two GPUs ====\n       GPU0         GPU1\nGPU0   0            XGMI\nGPU1   XGMI         0\n====\n"
    assert bench._link_types(topo) == {"GPU0": ["0", "XGMI"], "GPU1": ["XGMI", "0"]}
