"""A/B library builds on the whole training step (whisper-large-v3-turbo DiCoW, B=16, L=128: bench.py's default workload) AND on the
encoder forward alone, inside ONE process: model / optimizer state built once, the ctypes binding pointed at each build in turn,
interleaved rounds.   python tools/ab_step.py label=lib.so ...     (REPS rounds, default 3; STEPS per measurement, default 4)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg
pkg = amd_pkg.load()
from ts_asr_whisper_amd import _lib as L
from ts_asr_whisper_amd.trainer import TrainStep
from ts_asr_whisper_amd.data import synthetic_batch

specs = [a.split("=", 1) for a in sys.argv[1:]]
reps, steps = int(os.environ.get("REPS", "3")), int(os.environ.get("STEPS", "4"))
cfg = pkg.DiCoWConfig.preset(os.environ.get("ENC_MODEL", "whisper-large-v3-turbo"), use_fddt=True, fddt_is_diagonal=True,
                             use_pre_pos_fddt=True, fddt_init="suppressive", non_target_fddt_value=0.5)
B = int(os.environ.get("ENC_BATCH", "16"))
torch.manual_seed(0)
model = pkg.DiCoWForConditionalGeneration(cfg).cuda()
model.tie_weights()
ts = TrainStep(model, lr=2e-6, fddt_lr_multiplier=100.0, max_grad_norm=1.0, warmup_steps=2000, max_steps=40000,
               preheat_prefixes=("model.encoder.fddts", "model.encoder.initial_fddt"), use_fddt_only_n_steps=0)
batches = [synthetic_batch(cfg, B, 128, seed=1000 + i) for i in range(2)]


def use(path):
    L.LIB_PATH = os.path.abspath(path)
    L._lib = None
    L.lib()


def timed(fn, n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def enc(i):
    with torch.no_grad():
        model.model.encoder(batches[0]["input_features"], stno_mask=batches[0]["stno_mask"])


res = {k: ([], []) for k, _ in specs}
for r in range(reps):
    for label, path in specs:
        use(path)
        ts.step(batches[0])
        res[label][0].append(timed(lambda i: ts.step(batches[i % 2]), steps))
        enc(0)
        res[label][1].append(timed(enc, 5))
for label, _ in specs:
    s, e = res[label]
    print(f"{label:12s} step " + " ".join(f"{x:7.2f}" for x in s) + f"  med {statistics.median(s):7.2f} | encfwd " +
          " ".join(f"{x:6.2f}" for x in e) + f"  med {statistics.median(e):6.3f}", flush=True)
