"""Per-parameter gradient error of the HIP path against a real-dimension golden (diagnostic): python tools/diag_rd_grads.py rd_tiny"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg; amd_pkg.load()
from tests.util import load_golden, subsample
import tests.test_gpu_realdims as R
name = sys.argv[1] if len(sys.argv) > 1 else "rd_tiny"
z = load_golden(name)
model, cfg = R._build(z)
out = model(**R._batch(z, cfg))
out.loss.backward()
named = dict(model.named_parameters())
for n in str(z["watched"]).split("\n"):
    key = f"hard.g.{n}"
    if key + ".sub" not in z.files or named[n].grad is None:
        continue
    g = named[n].grad.float().cpu()
    ref_sub, ref_norm = torch.from_numpy(z[key + ".sub"]), float(z[key + ".norm"])
    r_sub = R._rel_l2(subsample(g, 512), ref_sub)
    r_norm = abs(float(g.double().norm()) - ref_norm) / max(ref_norm, 1e-30)
    print(f"{n:60s} ours sub {r_sub:.4f} norm {r_norm:.4f} | reference bf16: reldev {float(z['bf16.g.reldev.' + n]):.4f} subdev {float(z['bf16.g.subdev.' + n]):.4f}")
