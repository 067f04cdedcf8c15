"""FDDT-conditioned encoder forward alone (whisper-large-v3-turbo, B=16, torch.no_grad()) -- the north-star headline.
   python tools/enc_fwd.py [iters]           prints ms per forward and the MFMA fraction
   (tools/prof_encfwd.sh runs it under rocprofv3 --kernel-trace --stats: per-kernel share of the forward)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg
pkg = amd_pkg.load()
from ts_asr_whisper_amd.data import synthetic_batch

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
if os.environ.get("ENC_NO_FUSE") == "1":            # A/B: the next layer's FDDT as its own row kernel again
    from ts_asr_whisper_amd import engine as _eng
    _eng.FUSE_NEXT_FDDT = False
model_name = os.environ.get("ENC_MODEL", "whisper-large-v3-turbo")
B = int(os.environ.get("ENC_BATCH", "16"))
cfg = pkg.DiCoWConfig.preset(model_name, use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True, fddt_init="suppressive",
                             non_target_fddt_value=0.5)
torch.manual_seed(0)
model = pkg.DiCoWForConditionalGeneration(cfg).cuda()
model.tie_weights()
b0 = synthetic_batch(cfg, B, 128, seed=1000)
T_, D_, F_, Le, Mm = cfg.max_source_positions, cfg.d_model, cfg.encoder_ffn_dim, cfg.encoder_layers, cfg.num_mel_bins
flops = (Le * (8 * T_ * D_ * D_ + 4 * T_ * T_ * D_ + 4 * T_ * D_ * F_) + 6 * (2 * T_) * Mm * D_ + 6 * T_ * D_ * D_) * B
with torch.no_grad():
    for _ in range(3):
        model.model.encoder(b0["input_features"], stno_mask=b0["stno_mask"])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        model.model.encoder(b0["input_features"], stno_mask=b0["stno_mask"])
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(json.dumps({"encoder_forward_ms": round(ms, 3), "iters": iters, "batch": B, "model": model_name,
                  "tflops": round(flops / ms / 1e9, 1), "mfma_frac": round(flops / ms / 1e9 / 2500.0, 4)}))
