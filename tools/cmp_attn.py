import sys, os, subprocess, torch
CHILD = r'''
import sys, torch
sys.path.insert(0, ".")
import amd_pkg; amd_pkg.load()
from ts_asr_whisper_amd import ops
torch.manual_seed(0)
B, H, Lq, Lk, causal = [int(x) for x in sys.argv[1:6]]
D = H * 64; bf = torch.bfloat16
q = (torch.randn(B, Lq, H, 64, device="cuda") * 0.3).to(bf); k = torch.randn(B, Lk, H, 64, device="cuda").to(bf); v = torch.randn(B, Lk, H, 64, device="cuda").to(bf)
o = torch.empty_like(q); lse = torch.empty(B, H, Lq, device="cuda"); d_o = torch.randn(B, Lq, H, 64, device="cuda").to(bf)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v); delta = torch.empty(2, B, H, Lq, device="cuda")
ops.attn_fwd(q, k, v, o, lse, causal=bool(causal))
ops.attn_bwd(q, k, v, o, d_o, lse, delta, dq, dk, dv, causal=bool(causal), dq_scale=1.0)
torch.cuda.synchronize()
qf, kf, vf, dof = (t.float().permute(0, 2, 1, 3).requires_grad_(True) for t in (q, k, v, d_o))
s = qf @ kf.transpose(-1, -2)
if causal: s = s.masked_fill(torch.ones(Lq, Lk, device="cuda", dtype=torch.bool).triu(1), float("-inf"))
p = torch.softmax(s, -1); of = p @ vf
of.backward(dof.detach())
rel = lambda a, b: float((a.float() - b).norm() / b.norm())
print("o", rel(o.permute(0,2,1,3), of.detach()), "dq", rel(dq.permute(0,2,1,3), qf.grad), "dk", rel(dk.permute(0,2,1,3), kf.grad), "dv", rel(dv.permute(0,2,1,3), vf.grad))
'''
for lib in sys.argv[1:]:
    for shape in ("1 20 128 1500 0", "1 20 128 128 1", "2 3 300 300 1", "1 2 448 448 1", "2 3 40 40 1", "1 2 64 64 1", "1 2 65 65 1", "1 2 129 129 1", "1 2 200 200 1"):
        r = subprocess.run([sys.executable, "-c", CHILD] + shape.split(), env=dict(os.environ, DICOW_HIP_LIB=os.path.abspath(lib)), capture_output=True, text=True)
        print(lib.split("/")[-1], shape, "|", r.stdout.strip() or r.stderr[-300:])
