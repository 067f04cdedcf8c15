"""Board power and shader clock while one kernel family runs back to back (rocm-smi sampled from a side thread):
   python tools/power_probe.py [attn_bwd|attn_fwd|gemm_nt|gemm_tn|copy|idle] [seconds]"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg; amd_pkg.load()
from ts_asr_whisper_amd import ops
what = sys.argv[1] if len(sys.argv) > 1 else "attn_bwd"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
bf = torch.bfloat16
B, H, L = 16, 20, 1500
D = H * 64
qkv = torch.randn(B, L, 3 * D, device="cuda").to(bf); qkv[:, :, :D] *= 0.125
q, k, v = (qkv[:, :, i * D:(i + 1) * D].view(B, L, H, 64) for i in range(3))
o = torch.empty(B, L, H, 64, dtype=bf, device="cuda"); lse = torch.empty(B, H, L, device="cuda")
d_o = (torch.randn(B, L, H, 64, device="cuda") * 0.01).to(bf)
dqkv = torch.empty_like(qkv); dq, dk, dv = (dqkv[:, :, i * D:(i + 1) * D].view(B, L, H, 64) for i in range(3))
delta = torch.empty(2, B, H, L, device="cuda")
M = B * L
A = (torch.randn(M, 1280, device="cuda") * 0.5).to(bf); W = (torch.randn(5120, 1280, device="cuda") * 0.05).to(bf)
C = torch.empty(M, 5120, dtype=bf, device="cuda"); G = torch.zeros(5120, 1280, device="cuda")
x1, x2 = torch.empty(1 << 28, dtype=torch.uint8, device="cuda"), torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
ops.attn_fwd(q, k, v, o, lse, causal=False)
fns = {"attn_fwd": lambda: ops.attn_fwd(q, k, v, o, lse, causal=False),
       "attn_bwd": lambda: ops.attn_bwd(q, k, v, o, d_o, lse, delta, dq, dk, dv, causal=False, dq_scale=0.125),
       "gemm_nt": lambda: ops.gemm_nt(A, W, C, M, 5120, 1280),
       "gemm_tn": lambda: ops.gemm_tn(C, A, G, M, 5120, 1280),
       "copy": lambda: x1.copy_(x2), "idle": lambda: time.sleep(0.01)}
fn = fns[what]
samples, stop = [], False
def sampler():
    while not stop:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True)
        samples.append(r.stdout.strip())
        time.sleep(0.3)
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(20): fn()
    torch.cuda.synchronize(); n += 20
dt = time.time() - t0
stop = True; th.join()
import json, re
print(f"{what}: {n} launches in {dt:.2f} s = {dt / n * 1e6:.1f} us each")
for s in samples[2:8]:
    try:
        d = json.loads(s); c = d[sorted(d)[0]]
        print("   ", {k: v for k, v in c.items() if re.search(r"(?i)power|sclk|mclk|fclk", k)})
    except Exception as e:
        print("   ", s[:300])
