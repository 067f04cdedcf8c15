"""The bench step with the pooled weight-gradient launches on the compute stream vs on a side stream (engine.WGRAD_SIDE_STREAM), interleaved:
   [DICOW_WGRAD_STREAM_PRIORITY=-1|0] python tools/ab_wgrad_stream.py [steps=6] [reps=3]
First checks that the two orders leave bit-identical gradients (same kernels, same operands, one writer per dW)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg

pkg = amd_pkg.load()
from ts_asr_whisper_amd import engine, ops
from ts_asr_whisper_amd.data import synthetic_batch
from ts_asr_whisper_amd.trainer import TrainStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
over = dict(use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True, fddt_init="suppressive", non_target_fddt_value=0.5)
se = os.environ.get("AB_SE") == "1"
if se:
    over.update(use_enrollments=True, scb_layers=8)
cfg = pkg.DiCoWConfig.preset(os.environ.get("ENC_MODEL", "whisper-large-v3-turbo"), **over)
B = int(os.environ.get("ENC_BATCH", "16"))
torch.manual_seed(0)
model = pkg.DiCoWForConditionalGeneration(cfg).cuda()
model.tie_weights()
ts = TrainStep(model, lr=2e-6, fddt_lr_multiplier=100.0, max_grad_norm=1.0, warmup_steps=2000, max_steps=40000,
               preheat_prefixes=("model.encoder.fddts", "model.encoder.initial_fddt") + (("model.encoder.ca_enrolls",) if se else ()),
               use_fddt_only_n_steps=0, split_streams=os.environ.get("AB_SPLIT") == "1")
batches = [synthetic_batch(cfg, B, 128, seed=1000 + i, mixed_length=se, enrollments=se) for i in range(2)]


def grads_of(side):
    engine.WGRAD_SIDE_STREAM = side
    ts.begin_step()
    loss = ts._micro(batches[1], 1.0)
    ts.store.settle_first_writers()
    torch.cuda.synchronize()
    return float(loss), ts.store.grads.clone()


l0, g0 = grads_of(False)
l1, g1 = grads_of(True)
l2, g2 = grads_of(True)
print(f"loss {l0:.6f} / {l1:.6f}; gradients bit-equal to the one-stream order: {bool(torch.equal(g0, g1))}, run twice: {bool(torch.equal(g1, g2))}; "
      f"fused attention status {ops.attn_bwd_fused_status()}", flush=True)


def timed(n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        ts.step(batches[i % 2])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = {False: [], True: []}
for r in range(reps):
    for side in (False, True):
        engine.WGRAD_SIDE_STREAM = side
        ts.step(batches[0])
        res[side].append(timed(steps))
for side in (False, True):
    print(f"{'weight gradients on a side stream' if side else 'one stream':36s} ms/step " + " ".join(f"{x:7.2f}" for x in res[side]) +
          f"   median {statistics.median(res[side]):7.2f}", flush=True)
