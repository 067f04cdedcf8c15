"""Do a side stream's kernels run BESIDE the compute stream's, or does their time add to it?  (round 6: the fabric emulator and the
one-rank RCCL reducer both lengthened the step by exactly their own busy time.)
Main stream: N encoder forwards of the headline batch (persistent GEMMs, attention, row kernels -- the kernels of a step).  Side stream:
a 'sleeper' -- dicow_fabric_emulate on a 64 KB buffer paced to last T ms: W workgroups that touch 64 KB and otherwise s_sleep (no
bandwidth, no power).  Timed: main alone, sleeper alone, both started together.   python tools/probe_streams.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg
pkg = amd_pkg.load()
from ts_asr_whisper_amd import ops
from ts_asr_whisper_amd.data import synthetic_batch
cfg = pkg.DiCoWConfig.preset("whisper-large-v3-turbo", use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True, fddt_init="suppressive", non_target_fddt_value=0.5)
torch.manual_seed(0)
model = pkg.DiCoWForConditionalGeneration(cfg).cuda()
enc = model.model.encoder
b = synthetic_batch(cfg, 16, 128, seed=1)
x, st = b["input_features"], b["stno_mask"]
buf = torch.zeros(16384, device="cuda")


def main_work(n=3):
    with torch.no_grad():
        for _ in range(n):
            enc(x, stno_mask=st)


def wall(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


side = torch.cuda.Stream()
t_main = wall(main_work)
print(f"main stream alone (3 encoder forwards, B=16): {t_main:.1f} ms")
for cus in (0, 240):
    ops.set_gemm_cus(cus)
    for wgs in (16, 256):
        for T in (20.0, 60.0):
            gbps = buf.numel() * 4 / (T * 1e-3) / 1e9
            def sleeper():
                with torch.cuda.stream(side):
                    ops.fabric_emulate(buf, gbps, wgs, 1)
            t_side = wall(sleeper)
            def both():
                sleeper()
                main_work()
            t_both = wall(both)
            t_m2 = wall(main_work)
            print(f"gemm_cus {cus or 256:3d}  sleeper {wgs:3d} workgroups x {T:.0f} ms: alone {t_side:6.1f} ms | main alone {t_m2:6.1f} | together {t_both:6.1f} ms "
                  f"-> {'OVERLAP' if t_both < t_m2 + 0.5 * t_side else 'SERIAL (sum %.1f)' % (t_m2 + t_side)}")
ops.set_gemm_cus(0)
