"""A/B library builds on the FDDT-conditioned encoder forward (whisper-large-v3-turbo, B=16, torch.no_grad()) inside ONE process:
the model is built once, the ctypes binding is pointed at each build in turn, interleaved rounds.
   python tools/ab_encfwd.py label=lib.so ...        (REPS rounds, default 3; ITERS forwards per measurement, default 5)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg
pkg = amd_pkg.load()
from ts_asr_whisper_amd import _lib as L
from ts_asr_whisper_amd.data import synthetic_batch

specs = [a.split("=", 1) for a in sys.argv[1:]]
reps, iters = int(os.environ.get("REPS", "3")), int(os.environ.get("ITERS", "5"))
model_name = os.environ.get("ENC_MODEL", "whisper-large-v3-turbo")
B = int(os.environ.get("ENC_BATCH", "16"))
cfg = pkg.DiCoWConfig.preset(model_name, use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True, fddt_init="suppressive",
                             non_target_fddt_value=0.5)
torch.manual_seed(0)
model = pkg.DiCoWForConditionalGeneration(cfg).cuda()
model.tie_weights()
b0 = synthetic_batch(cfg, B, 128, seed=1000)


def use(path):
    L.LIB_PATH = os.path.abspath(path)
    L._lib = None
    L.lib()


def measure():
    with torch.no_grad():
        for _ in range(2):
            model.model.encoder(b0["input_features"], stno_mask=b0["stno_mask"])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            model.model.encoder(b0["input_features"], stno_mask=b0["stno_mask"])
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


res = {k: [] for k, _ in specs}
for r in range(reps):
    for label, path in specs:
        use(path)
        res[label].append(measure())
for label, _ in specs:
    v = res[label]
    print(f"{label:14s} " + " ".join(f"{x:7.3f}" for x in v) + f"   median {statistics.median(v):7.3f} ms", flush=True)
