"""Per-parameter gradient differences of the rd_turbo golden case between two builds of the library:
   python tools/cmp_model_grads.py libA.so libB.so"""
import os, subprocess, sys
CHILD = r'''
import sys, torch
sys.path.insert(0, ".")
from tests.util import load_golden
import tests.test_gpu_realdims as T
z = load_golden("rd_turbo")
model, cfg = T._build(z)
batch = T._batch(z, cfg)
out = model(**batch)
out.loss.backward()
torch.cuda.synchronize()
torch.save({"loss": float(out.loss), "enc": out.encoder_last_hidden_state.float().cpu(), "logits": out.logits.float().cpu(),
            "g": {n: p.grad.float().cpu() for n, p in model.named_parameters() if p.grad is not None}}, sys.argv[1])
'''
outs = []
for i, lib in enumerate(sys.argv[1:3]):
    o = f"/tmp/grads_{i}.pt"
    r = subprocess.run([sys.executable, "-c", CHILD, o], env=dict(os.environ, DICOW_HIP_LIB=os.path.abspath(lib)), capture_output=True, text=True)
    if r.returncode: print(r.stderr[-2000:]); sys.exit(1)
    outs.append(o)
import torch
a, b = torch.load(outs[0]), torch.load(outs[1])
print("loss", a["loss"], b["loss"], "enc maxdiff", float((a["enc"] - b["enc"]).abs().max()), "logits maxdiff", float((a["logits"] - b["logits"]).abs().max()))
rows = []
for n in a["g"]:
    ga, gb = a["g"][n], b["g"][n]
    rows.append((float((ga - gb).norm() / ga.norm().clamp_min(1e-30)), n))
rows.sort(reverse=True)
for r, n in rows[:25]: print(f"{r:10.3e}  {n}")
print("identical:", sum(1 for r, _ in rows if r == 0), "of", len(rows))
