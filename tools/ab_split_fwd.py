"""A/B of the encoder forward at the bench shape (whisper-large-v3-turbo DiCoW, B = 16): one stream against two half-batch streams
(engine.SPLIT_FWD), training form (gradients enabled: activations kept) and inference form, interleaved.  (A start stagger of the second
half -- behind the first half's qkv / attention / out-proj of layer 0 -- measured 0.3-0.4 ms slower than none: profiles/r06_split_fwd.txt.)   python tools/ab_split_fwd.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amd_pkg
pkg = amd_pkg.load()
from ts_asr_whisper_amd import engine
from ts_asr_whisper_amd.data import synthetic_batch
cfg = pkg.DiCoWConfig.preset("whisper-large-v3-turbo", use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True,
                             fddt_init="suppressive", non_target_fddt_value=0.5)
torch.manual_seed(0)
model = pkg.DiCoWForConditionalGeneration(cfg).cuda()
enc = model.model.encoder
b = synthetic_batch(cfg, 16, 128, seed=1)
T_, D_, F_, Le, Mm = cfg.max_source_positions, cfg.d_model, cfg.encoder_ffn_dim, cfg.encoder_layers, cfg.num_mel_bins
flops = (Le * (8 * T_ * D_ * D_ + 4 * T_ * T_ * D_ + 4 * T_ * D_ * F_) + 6 * (2 * T_) * Mm * D_ + 6 * T_ * D_ * D_) * 16
def timed(grad, n=5):
    with torch.set_grad_enabled(grad):
        for _ in range(2):
            o = enc(b["input_features"], stno_mask=b["stno_mask"]); del o
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            o = enc(b["input_features"], stno_mask=b["stno_mask"]); del o
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad():
    engine.SPLIT_FWD = False; ref = enc(b["input_features"], stno_mask=b["stno_mask"]).last_hidden_state.clone()
    engine.SPLIT_FWD = True; two = enc(b["input_features"], stno_mask=b["stno_mask"]).last_hidden_state.clone()
print("inference output, two streams vs one: bit-equal =", bool(torch.equal(ref, two)))
engine.SPLIT_FWD = False; r1 = enc(b["input_features"], stno_mask=b["stno_mask"]).last_hidden_state.detach().clone()
engine.SPLIT_FWD = True; r2 = enc(b["input_features"], stno_mask=b["stno_mask"]).last_hidden_state.detach().clone()
print("training-form output, two streams vs one: bit-equal =", bool(torch.equal(r1, r2)), "; vs inference form:", bool(torch.equal(r1, ref)))
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    for name, on, parts in (("one stream", False, 2), ("two streams", True, 2), ("four streams", True, 4)):
        engine.SPLIT_FWD, engine.SPLIT_FWD_PARTS = on, parts
        tt, ti = timed(True), timed(False)
        print(f"{name:14s} training form {tt:6.2f} ms = {flops / tt / 1e9 / 2500:.4f} of 2.5 PF   inference form {ti:6.2f} ms = {flops / ti / 1e9 / 2500:.4f}", flush=True)
