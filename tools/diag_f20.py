"""Update deviations of the F20 trajectory on the GPU (diagnostic): python tools/diag_f20.py tiny"""
import sys, os, ast, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg; amd_pkg.load()
from tests.util import load_golden, subsample, f20_batches
import tests.test_gpu_trajectory as TT
from ts_asr_whisper_amd.trainer import TrainStep
case = sys.argv[1] if len(sys.argv) > 1 else "tiny"
z = load_golden("f20_trajectory_" + case)
hp = ast.literal_eval(str(z["hp"]))
model, cfg, ts0 = TT._build(case, z)
start = {n: p.detach().clone() for n, p in model.named_parameters()}
ts = TrainStep(model, lr=hp["lr"], fddt_lr_multiplier=hp["mult"], weight_decay=hp["wd"], max_grad_norm=hp["max_norm"],
               warmup_steps=hp["warmup"], max_steps=hp["K"], frozen_keywords=("decoder",), use_fddt_only_n_steps=hp["n_pre"])
named = dict(model.named_parameters())
for k, b in enumerate(f20_batches(case, hp["K"], ts0)):
    loss = ts.step({n: v.cuda() for n, v in b.items()})
    print(k, float(loss), float(z["loss"][k]), float(z["bf16.loss"][k]), math.sqrt(float(ts.opt.gnorm_sq)), float(z["gnorm"][k]))
for n in str(z["watched"]).split("\n"):
    upd = (named[n].detach() - start[n]).float().cpu()
    ref, nrm = torch.from_numpy(z["upd.sub." + n]), float(z["upd.norm." + n])
    if nrm == 0: continue
    r_sub = float((subsample(upd, 512) - ref).double().norm() / ref.double().norm())
    print(f"{n:60s} sub {r_sub:.4f} norm {abs(float(upd.double().norm()) - nrm) / nrm:.4f} | reference bf16 {float(z['bf16.upd.reldev.' + n]):.4f}")
for n in ["model.encoder.layers.1.fc2.bias", "model.encoder.layer_norm.bias", "model.encoder.conv2.bias"]:
    upd = (named[n].detach() - start[n]).float().cpu()
    ref = torch.from_numpy(z["upd.sub." + n])
    mine = subsample(upd, 512)
    d = (mine - ref)
    idx = d.abs().argsort(descending=True)[:8]
    print(n, "n", mine.numel(), "ref rms", float(ref.pow(2).mean().sqrt()), "diff rms", float(d.pow(2).mean().sqrt()))
    for i in idx.tolist():
        print(f"   [{i}] ours {float(mine[i]):+.3e} ref {float(ref[i]):+.3e}")
