import sys, os, ast, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg; amd_pkg.load()
from tests.util import load_golden, f20_batches
import tests.test_gpu_trajectory as TT
from ts_asr_whisper_amd.trainer import TrainStep
z = load_golden("f20_trajectory_tiny")
hp = ast.literal_eval(str(z["hp"]))
model, cfg, ts0 = TT._build("tiny", z)
ts = TrainStep(model, lr=0.0, fddt_lr_multiplier=1.0, weight_decay=0.0, max_grad_norm=1.0, warmup_steps=0, max_steps=0, frozen_keywords=("decoder",), use_fddt_only_n_steps=0)
named = dict(model.named_parameters())
b = {n: v.cuda() for n, v in f20_batches("tiny", 1, ts0)[0].items()}
snaps = []
for k in range(3):
    ts.step(b)
    snaps.append({n: p.grad.detach().clone() for n, p in named.items() if p.requires_grad and p.grad is not None})
# plain autograd-only gradient (no TrainStep): a fresh model
model2, _, _ = TT._build("tiny", z)
for p in model2.parameters(): p.requires_grad_(True)
model2.model.encoder.embed_positions.weight.requires_grad_(False)
out = model2(**b); out.loss.backward()
n2 = dict(model2.named_parameters())
bad = 0
for n in snaps[0]:
    d01 = float((snaps[0][n] - snaps[1][n]).abs().max()); d12 = float((snaps[1][n] - snaps[2][n]).abs().max())
    ref = n2[n].grad
    dr = float((snaps[0][n].float() - ref.float()).abs().max()) / max(1e-30, float(ref.float().abs().max())) if ref is not None else -1
    if d01 > 0 or d12 > 0 or dr > 1e-3:
        bad += 1
        print(f"{n:60s} step0-1 {d01:.3e} step1-2 {d12:.3e} vs plain autograd rel {dr:.3e}")
print("parameters off:", bad, "of", len(snaps[0]))
