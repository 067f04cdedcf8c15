import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, amd_pkg
pkg = amd_pkg.load()
from ts_asr_whisper_amd.trainer import TrainStep
from ts_asr_whisper_amd.data import synthetic_batch
from torch.utils._python_dispatch import TorchDispatchMode
import collections, traceback
cfg = pkg.DiCoWConfig.preset("whisper-large-v3-turbo", use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True, fddt_init="suppressive", non_target_fddt_value=0.5)
torch.manual_seed(0)
model = pkg.DiCoWForConditionalGeneration(cfg).cuda(); model.tie_weights()
ts = TrainStep(model, lr=2e-6, fddt_lr_multiplier=100.0, max_grad_norm=1.0, warmup_steps=2000, max_steps=40000, preheat_prefixes=("model.encoder.fddts", "model.encoder.initial_fddt"))
b = synthetic_batch(cfg, 16, 128, seed=1)
ts.step(b); ts.step(b)
cnt = collections.Counter(); where = collections.defaultdict(collections.Counter)
class M(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        cnt[name] += 1
        if any(k in name for k in ("copy_", "fill_", "zero_", "clone", "cat", "_to_copy", "zeros")):
            st = traceback.extract_stack(limit=8)
            for fr in reversed(st):
                if "ts-asr-whisper_amd" in fr.filename or "bench" in fr.filename:
                    where[name][f"{os.path.basename(fr.filename)}:{fr.lineno}"] += 1; break
        return func(*args, **(kwargs or {}))
with M():
    ts.step(b)
torch.cuda.synchronize()
for k, v in cnt.most_common(14): print(v, k)
for k, c in where.items():
    print(k, c.most_common(8))
