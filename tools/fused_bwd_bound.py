"""What a fused (5-pass) attention backward has to pay for dQ, measured: a key-block-major kernel produces one dQ partial per
key block and query (nkb = ceil(1500 / keys per workgroup) partials of B*H*L*64 values) that must be written and summed in a
fixed order (deterministic).  This times just that traffic at the encoder shape, to set against the 273-287 us the separate
dQ kernel takes in the step (profiles/r02_kernel_stats.csv):  python tools/fused_bwd_bound.py"""
import torch
B, H, L, D = 16, 20, 1500, 64
x1, x2 = torch.empty(1 << 28, dtype=torch.uint8, device="cuda"), torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize(); ev = []
    for _ in range(iters):
        x1.copy_(x2)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); ev.append((s, e))
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in ev)[len(ev) // 2] * 1e3
for keys, dt in ((128, torch.bfloat16), (128, torch.float32), (256, torch.bfloat16), (256, torch.float32)):
    nkb = -(-L // keys)
    part = torch.randn(nkb, B * L, H * D, device="cuda").to(dt)
    src = torch.randn(B * L, H * D, device="cuda").to(dt)
    out = torch.empty(B * L, H * D, device="cuda", dtype=torch.bfloat16)
    t_w = timeit(lambda: [part[i].copy_(src) for i in range(nkb)])          # the partial stores (inside the fused kernel, overlappable)
    t_r = timeit(lambda: out.copy_(torch.sum(part, dim=0, dtype=torch.float32)))   # the ordered sum afterwards (a kernel of its own)
    print(f"{keys} keys per workgroup ({nkb} partials, {str(dt).split('.')[-1]}): partial stores {part.numel() * part.element_size() / 1e6:.0f} MB "
          f"= {t_w:.0f} us of HBM time, ordered sum + cast {t_r:.0f} us")
