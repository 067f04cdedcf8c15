"""The RCCL transport itself ("nccl" backend of torch.distributed on ROCm), opened on the one GPU the test box has: a
single-rank communicator exercises rendezvous, communicator creation, the side-stream bucketed all-reduce of the flat
gradient store and the stream / event hand-over exactly as an N-rank run does (reference scripts/submit_slurm.sh:34,
configs/base.yaml:73: torchrun + DDP over NCCL); the reduction over one rank is the identity, so results must be
BIT-IDENTICAL to the run without a process group.  Run with `pytest -m gpu`."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from tests.util import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "DICOW_FORCE_REDUCE")}
    env.update(PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra)
    return env


def _rank0_env():
    return _clean_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(_free_port()), DICOW_FORCE_REDUCE="1")


def _bench(env, *extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--profile-steps", "1", "--no-cpu-baseline", "--no-power", *extra],
                       capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout      # ONE JSON line: RCCL's version banner must not reach stdout
    return json.loads(lines[0])


def test_bench_headline_workload_through_a_one_rank_rccl_communicator():
    """bench.py on the headline model (whisper-large-v3-turbo DiCoW, decoder frozen) as rank 0 of a world of one with the
    reducer forced: backend nccl, one bucket per backward segment, 4 bytes x 637.3 M trainable gradients exchanged per step,
    and the same loss as the run that never opens a process group."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    small = ("--batch", "2", "--labels", "32")
    forced = _bench(_rank0_env(), *small)
    plain = _bench(_clean_env(), *small)
    ar = forced["allreduce"]
    assert ar["backend"] == "nccl" and ar["forced_on_one_rank"] is True
    assert forced["n_gpus"] == 1 and forced["config"]["parallelism"] == "dp1"
    n_train = forced["config"]["trainable_params"]
    assert 637.0e6 < n_train < 637.6e6, n_train                          # 637.3 M with the decoder frozen
    assert ar["bytes_per_step"] == 4 * n_train
    assert ar["buckets_per_step"] >= 32                                   # >= one bucket per encoder layer
    assert len(ar["exposed_ms_per_step"]) == 1 and ar["exposed_ms_per_step"][0] >= 0.0
    assert plain["allreduce"]["backend"] is None and plain["allreduce"]["bytes_per_step"] == 0
    assert forced["loss"] == plain["loss"], (forced["loss"], plain["loss"])   # identity reduction: bit-equal training


CHILD = r"""
import os, sys, json, hashlib
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["DICOW_ROOT"])
import amd_pkg
pkg = amd_pkg.load()
from ts_asr_whisper_amd.trainer import TrainStep
from ts_asr_whisper_amd.data import synthetic_batch
use_pg = os.environ.get("RANK") is not None
torch.cuda.set_device(0)
if use_pg:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
cfg = pkg.DiCoWConfig.preset("whisper-tiny", use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True, fddt_init="suppressive",
                             non_target_fddt_value=0.5)
torch.manual_seed(0)
model = pkg.DiCoWForConditionalGeneration(cfg).cuda()
model.tie_weights()
ts = TrainStep(model, lr=1e-4, fddt_lr_multiplier=10.0, use_fddt_only_n_steps=1,
               preheat_prefixes=("model.encoder.fddts", "model.encoder.initial_fddt"))
if use_pg:
    assert ts.reducer.force and ts.reducer.stream is not None and ts.reducer.world == 1
    # start-up replica sync (DDP's constructor broadcast) ran through RCCL: flat store + frozen parameters + buffers, a no-op in value
    assert ts.replica_sync == "broadcast" and ts.replica_sync_bytes >= 4 * ts.store.numel, ts.replica_sync_bytes
else:
    assert ts.replica_sync_bytes == 0
batches = [synthetic_batch(cfg, 2, 12, seed=40 + i) for i in range(2)]
losses = [float(ts.step(batches[i % 2])) for i in range(3)]          # step 1: preheat-only exchange, then the bucketed one
losses.append(float(ts.step([batches[0], batches[1]])))              # gradient accumulation: one exchange after the last micro-batch
torch.cuda.synchronize()
h = hashlib.sha256()
for n, p in sorted(model.named_parameters()):
    h.update(n.encode()); h.update(p.detach().float().cpu().numpy().tobytes())
h.update(ts.store.exp_avg.cpu().numpy().tobytes()); h.update(ts.store.exp_avg_sq.cpu().numpy().tobytes())
print(json.dumps({"losses": losses, "sha": h.hexdigest(), "backend": dist.get_backend() if use_pg else None}))
if use_pg:
    dist.destroy_process_group()
"""


def test_train_step_bit_identical_with_and_without_the_rccl_reducer():
    """TrainStep (preheat phase -> full phase -> gradient accumulation) as rank 0 of a one-rank RCCL group with the reducer
    forced: parameters and Adam moments after 4 optimizer steps are bit-identical to the run without a process group."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    outs = []
    for env in (_rank0_env(), _clean_env()):
        env["DICOW_ROOT"] = ROOT
        r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]))
    assert outs[0]["backend"] == "nccl" and outs[1]["backend"] is None
    assert outs[0]["losses"] == outs[1]["losses"], (outs[0]["losses"], outs[1]["losses"])
    assert outs[0]["sha"] == outs[1]["sha"]


def test_dry_launch_opens_rccl_on_one_rank():
    """The launcher's rendezvous self-test (`--dry-launch`) on the GPU: nccl backend, one all-reduce, clean teardown."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-launch"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT, env=_rank0_env())
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["dry_launch"] is True and d["backend"] == "nccl" and d["n_gpus"] == 1 and d["allreduce_sum"] == 1.0
