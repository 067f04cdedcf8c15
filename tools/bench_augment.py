"""Time the on-GPU augmentation block (host planner + the three kernels of csrc/augment.hip) at the bench batch.
usage: python tools/bench_augment.py [B]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import amd_pkg

amd_pkg.load()
from ts_asr_whisper_amd import augment as A

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M, Tn = 128, 1500
mel = torch.randn(B, M, 2 * Tn, device="cuda").clamp_(-1.5, 1.5)
stno = torch.softmax(torch.randn(B, 4, Tn, device="cuda"), 1)


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    t0 = time.perf_counter()
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return host * 1e3, ev[0].elapsed_time(ev[1]) / n


torch.manual_seed(0)
for name, fn in (("plan_soft_segments (host only)", lambda: A.plan_soft_segments(B, 4, Tn, 0.1, 5, 50)),
                 ("plan_gaussian_noise (host only)", lambda: A.plan_gaussian_noise(B, 4, Tn, 0.2, 0.75)),
                 ("plan_spec_aug (host only)", lambda: A.plan_spec_aug(B, 2 * Tn, M + 4)),
                 ("soft_segment_augmentation", lambda: A.soft_segment_augmentation(stno, 0.1, 5, 50)),
                 ("add_gaussian_noise_and_rescale", lambda: A.add_gaussian_noise_and_rescale(stno, 0.2, 0.75)),
                 ("spec_aug_joint", lambda: A.spec_aug_joint(mel, stno))):
    h, g = timed(fn)
    print(f"{name:36s} host {h:8.3f} ms   wall/gpu {g:8.3f} ms")
# kernels alone, plans prepared ahead
plan = A.plan_spec_aug(B, 2 * Tn, M + 4)
plan.fmask, plan.tmask = plan.fmask.cuda(), plan.tmask.cuda()
h, g = timed(lambda: A.spec_aug_joint(mel, stno, plan=plan), 50)
byt = 2 * mel.numel() * 4 + 2 * stno.numel() * 4
print(f"spec_aug_joint kernel (fixed plan)    host {h:8.3f} ms   gpu {g:8.3f} ms   {byt / g / 1e6:.0f} GB/s algorithmic")
