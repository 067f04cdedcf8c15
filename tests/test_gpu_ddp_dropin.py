"""The drop-in claim of INTEGRATION.md, section 1: the module tree works under the reference's own harness pieces -- torch's
DistributedDataParallel wrapper (reference: HF Trainer under torchrun, scripts/submit_slurm.sh:34) and a torch.optim.AdamW built
the way reference src/models/containers.py:100-114 builds it (two parameter groups, the preheat prefixes at lr x multiplier with
weight decay 0) -- without TrainStep / FlatStore.  Two ranks share the one MI355X of the test box and talk over gloo (the DDP
bucketing, hooks and optimizer are the same as under "nccl").

Checks: (1) after backward the DDP-averaged gradients equal those of one process accumulating both micro-batches in TrainStep's
flat store; (2) both ranks hold bit-identical parameters after two optimizer steps; (3) those parameters match the fused
two-group AdamW of TrainStep run on the same data.  Run with `pytest -m gpu`."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
PREFIXES = ("model.encoder.fddts", "model.encoder.initial_fddt")
LR, MULT, WD = 1e-4, 10.0, 0.01


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model_and_batches():
    import amd_pkg
    pkg = amd_pkg.load()
    from ts_asr_whisper_amd.data import synthetic_batch
    cfg = pkg.DiCoWConfig.preset("whisper-tiny", use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True,
                                 fddt_init="suppressive", non_target_fddt_value=0.5)
    torch.manual_seed(0)
    model = pkg.DiCoWForConditionalGeneration(cfg).cuda()
    model.tie_weights()
    return model, [synthetic_batch(cfg, 2, 12, seed=70 + i) for i in range(2)]


def _reference_optimizer(model):
    """reference containers.py:100-114 (get_optimizer with use_custom_optimizer)."""
    named = [(n, p) for n, p in model.named_parameters()]
    base = [p for n, p in named if not any(n.startswith(pre) for pre in PREFIXES)]
    new = [p for n, p in named if any(n.startswith(pre) for pre in PREFIXES)]
    return torch.optim.AdamW([{"params": base}, {"params": new, "lr": MULT * LR, "weight_decay": 0.0}], lr=LR, weight_decay=WD)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model, batches = _model_and_batches()                       # (registers the package: amd_pkg.load())
        from ts_asr_whisper_amd.trainer import freeze_by_keyword
        freeze_by_keyword(model, ("decoder",))                      # containers.py:80-90
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0])
        opt = _reference_optimizer(model)
        grads = None
        for step in range(2):
            opt.zero_grad(set_to_none=True)
            out = ddp(**batches[rank])
            out.loss.backward()                                     # DDP all-reduces (averages) the gradients in its hooks
            if step == 0:
                grads = {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters() if p.requires_grad}
            torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.requires_grad], 1.0)
            opt.step()
        params = {n: p.detach().cpu().numpy() for n, p in model.named_parameters() if p.requires_grad}
        q.put((rank, grads, params))
    finally:
        dist.destroy_process_group()


def test_module_tree_under_torch_ddp_and_reference_adamw():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    for _ in range(world):
        try:
            res.append(q.get(timeout=300))
        except Exception:
            for p in procs:
                p.join(timeout=5)
            raise AssertionError(f"a DDP worker died: exit codes {[p.exitcode for p in procs]}")
    res.sort(key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # ---- one process, TrainStep / FlatStore, both micro-batches accumulated
    model, batches = _model_and_batches()
    from ts_asr_whisper_amd.trainer import TrainStep
    start = None
    ts = TrainStep(model, lr=LR, fddt_lr_multiplier=MULT, weight_decay=WD, max_grad_norm=1.0, warmup_steps=0, max_steps=0,
                   preheat_prefixes=PREFIXES)
    start = {n: p.detach().clone().cpu() for n, p in model.named_parameters() if p.requires_grad}
    ts.begin_step()
    for b in batches:
        ts._micro(b, 0.5)
    torch.cuda.synchronize()
    flat = {n: p.grad.detach().clone().cpu() for n, p in model.named_parameters() if p.requires_grad}
    assert set(flat) == set(res[0][1])                               # the same parameters train under both harnesses
    num = den = 0.0
    for n, g in flat.items():
        gd = torch.from_numpy(res[0][1][n])
        assert (res[0][1][n] == res[1][1][n]).all(), n               # DDP left both ranks with the same averaged gradient
        num += float((gd - g).double().pow(2).sum())
        den += float(g.double().pow(2).sum())
    assert (num / den) ** 0.5 < 2e-2, (num / den) ** 0.5             # bf16 kernels, different summation order
    ts.finish_step()
    ts.step(batches)                                                 # second optimizer step
    torch.cuda.synchronize()
    num = den = 0.0
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        assert (res[0][2][n] == res[1][2][n]).all(), n               # ranks in lock-step after two torch.optim.AdamW steps
        num += float((torch.from_numpy(res[0][2][n]) - p.detach().cpu()).double().pow(2).sum())
        den += float((p.detach().cpu() - start[n]).double().pow(2).sum())
    assert num ** 0.5 < 0.05 * den ** 0.5, (num ** 0.5, den ** 0.5)  # same update as the fused two-group AdamW + clip
