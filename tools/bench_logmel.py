"""dicow_logmel on 16 clips of 30 s: the shipped kernel (DFT on the matrix pipe) against the direct-DFT build (tools/libv_lmdirect.so,
-DLOGMEL_DIRECT) in one process -- time per call, output difference, algorithmic GB/s.   python tools/bench_logmel.py [lib ...]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg
pkg = amd_pkg.load()
from ts_asr_whisper_amd import _lib as L, features

B, M = int(os.environ.get("LM_B", "16")), int(os.environ.get("LM_MELS", "128"))
g = torch.Generator().manual_seed(0)
wave = (torch.randn(B, features.N_SAMPLES, generator=g) * 0.1).cuda()
libs = [("shipped", L.LIB_PATH)] + [(os.path.basename(p), p) for p in sys.argv[1:]]
outs = {}
for name, path in libs:
    L.LIB_PATH = os.path.abspath(path); L._lib = None; L.lib()
    for _ in range(3):
        o = features.log_mel(wave, M)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            o = features.log_mel(wave, M)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    outs[name] = o.float().cpu()
    ms = statistics.median(ts)
    mb = (wave.numel() * 4 + o.numel() * 4 * 3) / 1e6          # audio read + features written, re-read and re-written by the finalize pass
    print(f"{name:20s} {ms:7.4f} ms per call ({B} clips)   {mb / ms:7.1f} GB/s algorithmic ({mb:.0f} MB)   "
          f"{2 * B * 3000 * 400 * 402 / ms / 1e9:6.1f} TFLOP/s of the dense DFT", flush=True)
names = list(outs)
for n in names[1:]:
    print(f"max |{names[0]} - {n}| = {float((outs[names[0]] - outs[n]).abs().max()):.3e}")
