"""Encoder forward (whisper-large-v3-turbo, B=16, torch.no_grad()) with the LayerNorm fold on / off, interleaved in ONE process, plus
the deviation between the two outputs and each one's deviation from the training-mode forward.
   python tools/ab_lnfold.py        (REPS rounds, default 4; ITERS forwards per measurement, default 5)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg
pkg = amd_pkg.load()
from ts_asr_whisper_amd import _lib as L
if "DICOW_HIP_LIB" not in os.environ:      # the fold lives in the experiments library only (csrc/build.sh --exp)
    L.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ts-asr-whisper_amd", "libdicow_hip_exp.so")
from ts_asr_whisper_amd import engine as E
from ts_asr_whisper_amd.data import synthetic_batch

reps, iters = int(os.environ.get("REPS", "4")), int(os.environ.get("ITERS", "5"))
B = int(os.environ.get("ENC_BATCH", "16"))
cfg = pkg.DiCoWConfig.preset("whisper-large-v3-turbo", use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True, fddt_init="suppressive",
                             non_target_fddt_value=0.5)
torch.manual_seed(0)
model = pkg.DiCoWForConditionalGeneration(cfg).cuda()
model.tie_weights()
b0 = synthetic_batch(cfg, B, 128, seed=1000)
enc = model.model.encoder


def measure(grad=False):
    with torch.set_grad_enabled(grad):
        for _ in range(2):
            o = enc(b0["input_features"], stno_mask=b0["stno_mask"]); del o
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            o = enc(b0["input_features"], stno_mask=b0["stno_mask"]); del o
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


res = {"fold": [], "nofold": [], "train": []}
for r in range(reps):
    for label in ("fold", "nofold", "train"):
        E.LN_FOLD = label == "fold"
        res[label].append(measure(grad=label == "train"))
for label, v in res.items():
    print(f"{label:8s} " + " ".join(f"{x:7.3f}" for x in v) + f"   median {statistics.median(v):7.3f} ms", flush=True)
outs = {}
with torch.no_grad():
    for label in ("fold", "nofold"):
        E.LN_FOLD = label == "fold"
        outs[label] = enc(b0["input_features"], stno_mask=b0["stno_mask"]).last_hidden_state.float()
E.LN_FOLD = True
tr = enc(b0["input_features"], stno_mask=b0["stno_mask"]).last_hidden_state.detach().float()
def rel(a, b): return float((a - b).norm() / b.norm())
print("fold vs nofold: max |diff|", float((outs["fold"] - outs["nofold"]).abs().max()), "rel-L2", rel(outs["fold"], outs["nofold"]),
      "| nofold vs train: max", float((outs["nofold"] - tr).abs().max()), "| fold vs train: max", float((outs["fold"] - tr).abs().max()),
      "rel-L2", rel(outs["fold"], tr), "| output absmax", float(tr.abs().max()))
