"""The encoder's initial FDDT + positions (bf16 rows in, fp32 rows out) at B=16, T=1500, D=1280: us per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg; amd_pkg.load()
from ts_asr_whisper_amd import ops
B, T, D = 16, 1500, 1280
rows = B * T
x = torch.randn(rows, D, device="cuda").bfloat16(); h = torch.empty(rows, D, device="cuda")
stno = torch.rand(B, 4, T, device="cuda"); pos = torch.randn(T, D, device="cuda")
w = tuple(torch.randn(D, device="cuda") for _ in range(4)); b = tuple(torch.randn(D, device="cuda") for _ in range(4))
big = torch.empty(1 << 28, dtype=torch.uint8, device="cuda"); big2 = torch.empty_like(big)
def run(): ops.fddt_ln_fwd(x, rows, D, mode=ops.MODE_DIAG, stno=stno, T=T, w=w, b=b, pos=pos, h_out=h)
for _ in range(3): run()
ev = []
for _ in range(10):
    big.copy_(big2)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); run(); e.record(); ev.append((s, e))
torch.cuda.synchronize()
ts = sorted(s.elapsed_time(e) * 1e3 for s, e in ev)
print(f"initial FDDT + pos: median {ts[len(ts)//2]:.1f} us  ({(rows*D*(2+4)+T*D*4)/ts[len(ts)//2]/1e6:.2f} TB/s)")
