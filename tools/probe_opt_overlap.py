"""TIMING-ONLY probe (results are garbage: dependencies ignored): what would it be worth to run the optimizer of step n (gradient norm, 65 AdamW
launches, the bf16 weight refresh) on a side stream UNDER the forward of step n+1?   python tools/probe_opt_overlap.py [steps=8] [reps=3]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg
pkg = amd_pkg.load()
from ts_asr_whisper_amd.data import synthetic_batch
from ts_asr_whisper_amd.trainer import TrainStep
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = pkg.DiCoWConfig.preset("whisper-large-v3-turbo", use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True, fddt_init="suppressive",
                             non_target_fddt_value=0.5)
torch.manual_seed(0)
model = pkg.DiCoWForConditionalGeneration(cfg).cuda()
model.tie_weights()
ts = TrainStep(model, lr=2e-6, fddt_lr_multiplier=100.0, max_grad_norm=1.0, warmup_steps=2000, max_steps=40000,
               preheat_prefixes=("model.encoder.fddts", "model.encoder.initial_fddt"), use_fddt_only_n_steps=0)
batches = [synthetic_batch(cfg, 16, 128, seed=1000 + i) for i in range(2)]
side = torch.cuda.Stream(priority=-1)
mode = {"overlap": False}
orig_finish, orig_micro = ts.finish_step, ts._micro
def finish_step():
    if not mode["overlap"]:
        return orig_finish()
    main = torch.cuda.current_stream()
    ts.reducer.finish()
    ts.store.settle_first_writers()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        ts.opt.step(preheat_only=False)
    enc = ts.model.model.encoder
    enc._sig = None
def micro(batch, scale):
    if not mode["overlap"]:
        return orig_micro(batch, scale)
    out = ts.model(**batch)
    torch.cuda.current_stream().wait_stream(side)       # the backward writes gradients: the optimizer must be through
    out.loss.backward()
    return out.loss.detach()
ts.finish_step, ts._micro = finish_step, micro
def timed(n):
    for i in range(2):
        ts.step(batches[i % 2])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        ts.step(batches[i % 2])
    torch.cuda.current_stream().wait_stream(side)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for r in range(reps):
    for name, ov in (("optimizer behind the backward (shipped)", False), ("optimizer under the next forward (racing)", True)):
        mode["overlap"] = ov
        print(f"{name:45s} {timed(steps):7.2f} ms per step", flush=True)
