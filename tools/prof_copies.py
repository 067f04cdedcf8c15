"""Which host-side operations issue the device-to-device copies (__amd_rocclr_copyBuffer) of a training step?  torch.profiler over one
step (it sees the autograd thread too), memcpy / copy ops with their Python stacks."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, amd_pkg
pkg = amd_pkg.load()
from ts_asr_whisper_amd.trainer import TrainStep
from ts_asr_whisper_amd.data import synthetic_batch
from torch.profiler import profile, ProfilerActivity
cfg = pkg.DiCoWConfig.preset("whisper-large-v3-turbo", use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True, fddt_init="suppressive", non_target_fddt_value=0.5)
torch.manual_seed(0)
model = pkg.DiCoWForConditionalGeneration(cfg).cuda(); model.tie_weights()
ts = TrainStep(model, lr=2e-6, fddt_lr_multiplier=100.0, max_grad_norm=1.0, warmup_steps=2000, max_steps=40000, preheat_prefixes=("model.encoder.fddts", "model.encoder.initial_fddt"))
b = synthetic_batch(cfg, 16, 128, seed=1)
ts.step(b); ts.step(b)
mode = sys.argv[1] if len(sys.argv) > 1 else "step"
def work():
    if mode == "step":
        ts.step(b)
    else:
        with torch.set_grad_enabled(mode == "encfwd_train"):
            o = model.model.encoder(b["input_features"], stno_mask=b["stno_mask"]); del o
work(); work()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    work()
    torch.cuda.synchronize()
ev = prof.events()
names = collections.Counter(e.name for e in ev)
print("events with copy / Memcpy in the name:")
for k, v in names.most_common():
    if "copy" in k.lower() or "memcpy" in k.lower() or "contiguous" in k.lower() or "clone" in k.lower():
        print(f"  {v:5d}  {k[:120]}")
print("all device kernels / memcpy of the region:")
for k, v in names.most_common():
    if k.startswith(("Memcpy", "Memset", "void at::", "__amd")):
        print(f"  {v:5d}  {k[:120]}")
where = collections.Counter()
for e in ev:
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::to", "aten::zeros", "aten::fill_", "aten::zero_", "aten::cat"):
        st = [s for s in (e.stack or []) if "ts-asr-whisper_amd" in s or "bench" in s]
        where[(e.name, st[0] if st else "(no package frame: autograd / torch internals)", str(e.input_shapes)[:60])] += 1
for k, v in where.most_common(25):
    print(v, k)
