"""Probe (TIMING ONLY -- the two halves' parameter gradients race in this form): forward + backward of the bench step as ONE
B = 16 batch on one stream vs TWO B = 8 halves on two HIP streams, gradients into a trainer.FlatStore.
   python tools/dual_stream_step_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg
pkg = amd_pkg.load()
from ts_asr_whisper_amd.data import synthetic_batch
from ts_asr_whisper_amd.trainer import TrainStep, freeze_by_keyword

cfg = pkg.DiCoWConfig.preset("whisper-large-v3-turbo", use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True,
                             fddt_init="suppressive", non_target_fddt_value=0.5)
torch.manual_seed(0)
model = pkg.DiCoWForConditionalGeneration(cfg).cuda()
model.tie_weights()
freeze_by_keyword(model, ("decoder",))
ts = TrainStep(model, lr=2e-6)
B = 16
b = synthetic_batch(cfg, B, 128, seed=1)
keys = [k for k in b if torch.is_tensor(b[k]) and b[k].shape[0] == B]
halves = [{k: (b[k][i * 8:(i + 1) * 8].contiguous() if k in keys else b[k]) for k in b} for i in range(2)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)


def single():
    ts.store.zero_grad(first_writer=True)
    out = model(**b)
    out.loss.backward()
    ts.store.settle_first_writers()


def dual(order=(0, 1)):
    ts.store.zero_grad(first_writer=True)
    cur = torch.cuda.current_stream()
    losses = [None, None]
    for i, s in zip(order, (s1, s2)):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            losses[i] = model(**halves[i]).loss
    for i, s in zip(order, (s1, s2)):
        with torch.cuda.stream(s):
            (losses[i] * 0.5).backward()
    cur.wait_stream(s1); cur.wait_stream(s2)
    ts.store.settle_first_writers()


def timed(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


single()           # (refreshes the bf16 weight copies on the current stream before anything forks)
torch.cuda.synchronize()
for rep in range(2):
    print(f"single stream, B = 16 fwd + bwd:            {timed(single):7.2f} ms", flush=True)
    print(f"two streams, B = 8 + 8 fwd + bwd:           {timed(dual):7.2f} ms", flush=True)
    print(f"one stream, B = 8 twice (micro-batches):    {timed(lambda: (ts.store.zero_grad(first_writer=True), [(model(**h).loss * 0.5).backward() for h in halves], ts.store.settle_first_writers())):7.2f} ms", flush=True)
