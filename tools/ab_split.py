"""The bench step as ONE stream vs as two half batches on two streams (TrainStep(split_streams=True)), interleaved in one process:
   python tools/ab_split.py [steps=6] [reps=3]
prints ms per optimizer step for each form, and first checks the split form: loss and gradient agreement with the one-stream step,
bit-reproducibility of its gradients run to run, the fused attention backward's status word."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg

pkg = amd_pkg.load()
from ts_asr_whisper_amd import ops
from ts_asr_whisper_amd.data import synthetic_batch
from ts_asr_whisper_amd.trainer import TrainStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = pkg.DiCoWConfig.preset(os.environ.get("ENC_MODEL", "whisper-large-v3-turbo"), use_fddt=True, fddt_is_diagonal=True,
                             use_pre_pos_fddt=True, fddt_init="suppressive", non_target_fddt_value=0.5)
B = int(os.environ.get("ENC_BATCH", "16"))
torch.manual_seed(0)
model = pkg.DiCoWForConditionalGeneration(cfg).cuda()
model.tie_weights()
ts = TrainStep(model, lr=2e-6, fddt_lr_multiplier=100.0, max_grad_norm=1.0, warmup_steps=2000, max_steps=40000,
               preheat_prefixes=("model.encoder.fddts", "model.encoder.initial_fddt"), use_fddt_only_n_steps=0)
batches = [synthetic_batch(cfg, B, 128, seed=1000 + i) for i in range(2)]
batches[1]["labels"][3, 100:] = -100          # unequal label counts in the two halves: the weights matter
batches[1]["labels"][12, 40:] = -100


def grads_of(split):
    ts.split_streams = split
    ts.begin_step()
    loss = ts._micro(batches[1], 1.0)
    if ts.first_writer:
        ts.store.settle_first_writers()
    torch.cuda.synchronize()
    return float(loss), ts.store.grads.clone()


def grads_accum():
    """the two halves as micro-batches on ONE stream (the gradient-accumulation path): what the split form must reproduce"""
    ts.split_streams = False
    hb = B // 2
    halves = [{k: (v[i * hb:(i + 1) * hb] if torch.is_tensor(v) and v.dim() > 0 else v) for k, v in batches[1].items()} for i in (1, 0)]
    ts.begin_step()
    loss = sum(ts._micro(h, 0.5) for h in halves) * 0.5
    if ts.first_writer:
        ts.store.settle_first_writers()
    torch.cuda.synchronize()
    return float(loss), ts.store.grads.clone()


l0, g0 = grads_of(False)
la, ga = grads_accum()
l1, g1 = grads_of(True)
l2, g2 = grads_of(True)
rel = float((g1 - g0).norm() / g0.norm())
print(f"loss one stream {l0:.6f}  split {l1:.6f}  |  gradient: rel. difference {rel:.3e}, max abs {float((g1 - g0).abs().max()):.3e} "
      f"(norm {float(g0.norm()):.4f})  |  split run twice bit-equal: {bool(torch.equal(g1, g2))}  loss equal: {l1 == l2}  |  "
      f"fused attention status {ops.attn_bwd_fused_status()}", flush=True)


print(f"two micro-batches on one stream: loss {la:.6f}; split == micro-batches bit for bit: {bool(torch.equal(g1, ga))} "
      f"(rel. difference {float((g1 - ga).norm() / ga.norm()):.3e}); micro-batches vs one batch: rel. {float((ga - g0).norm() / g0.norm()):.3e}", flush=True)
names = {id(p): n for n, p in model.named_parameters()}
worst = []
for ent in ts.store.entries:
    p, a, b = ent[0], ent[1], ent[1] + ent[2]
    d0 = g0[a:b]
    if float(d0.norm()) > 0:
        worst.append((float((g1[a:b] - d0).norm() / d0.norm()), float(d0.norm()), names[id(p)]))
worst.sort(reverse=True)
for r, n, name in worst[:8]:
    print(f"   {name:60s} rel {r:.3e}  (norm {n:.3e})", flush=True)


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
    for split in (False, True):
        ts.split_streams = split
        ts.step(batches[0])
        res[split].append(timed(steps))
for split in (False, True):
    print(f"{'two half-batch streams' if split else 'one stream':24s} ms/step " + " ".join(f"{x:7.2f}" for x in res[split]) +
          f"   median {statistics.median(res[split]):7.2f}", flush=True)
