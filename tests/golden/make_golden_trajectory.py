#!/usr/bin/env python3
"""F20: a short TRAINING TRAJECTORY of the real reference (build container only; imports /root/reference read-only).

The reference's own harness pieces drive the steps:
  * ``WhisperContainer.freeze_except`` (src/models/containers.py:92-98, called on a stand-in ``self``) for the preheat phase
    (train.py:174-178), the unfreeze rule of ``CustomTrainer.training_step`` (src/utils/trainers.py:116-139: after
    ``use_fddt_only_n_steps`` optimizer steps every parameter whose name contains none of the frozen keywords trains);
  * ``get_optimizer`` (containers.py:100-114): torch AdamW, two groups, the preheat prefixes at lr x fddt_lr_multiplier and
    weight decay 0;
  * transformers' ``get_cosine_schedule_with_warmup`` and the HF Trainer's inner order: backward, clip_grad_norm_ over all
    parameters, optimizer.step, lr_scheduler.step, zero_grad.
Weights and batches are integer-hashed (tests/util.py), so the fixture holds OUTPUTS only: per-step loss, gradient norm and
learning rates, sub-samples of watched parameters after the last step -- in fp32, plus the same trajectory under bf16 autocast
(the yard-stick of the GPU tolerances).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_trajectory.py [small small_se tiny]
"""
import os
import sys
import time
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as MG                      # transformers-5 tuple shim + the reference model classes
import make_golden_realdims as RD

MG._install_placeholders()                    # peft & co. (containers.py imports them at module scope)

import numpy as np
import torch
from transformers import get_cosine_schedule_with_warmup

from models.containers import WhisperContainer, get_optimizer          # noqa: E402  (the real reference functions)
from tests.util import hashed_init_, subsample, f20_batches, F20_PREFIXES

PREFIXES = F20_PREFIXES                       # configs/base.yaml:18 prefixes_to_preheat
FROZEN = ["decoder"]
HP = {  # name: (steps, preheat steps, lr, multiplier, weight decay, warm-up steps, max grad norm)
    "small": dict(K=8, n_pre=3, lr=2e-4, mult=50.0, wd=0.01, warmup=2, max_norm=1.0),
    "small_se": dict(K=8, n_pre=3, lr=2e-4, mult=50.0, wd=0.01, warmup=2, max_norm=1.0),
    "tiny": dict(K=8, n_pre=3, lr=2e-4, mult=100.0, wd=0.01, warmup=1, max_norm=1.0),
}
WATCH = ["model.encoder.fddts.0.target_linear.weight", "model.encoder.fddts.1.non_target_linear.bias",
         "model.encoder.initial_fddt.silence_linear.weight", "model.encoder.layers.0.self_attn.q_proj.weight",
         "model.encoder.layers.1.fc1.weight", "model.encoder.layers.1.fc2.bias", "model.encoder.layers.0.final_layer_norm.weight",
         "model.encoder.layer_norm.bias", "model.encoder.conv1.weight", "model.encoder.conv2.bias",
         "model.encoder.embed_positions.weight", "model.decoder.layers.0.fc1.weight", "model.decoder.embed_tokens.weight",
         "model.encoder.ca_enrolls.0.cae.cross_attn.q_proj.weight", "model.encoder.ca_enrolls.0.cae.cross_attn.out_proj.bias",
         "model.encoder.ca_enrolls.0.cae.ffn.0.weight", "model.encoder.ca_enrolls.0.cae.cross_gate.gate"]


def build(case):
    if case.startswith("small"):
        cfg = MG.small_cfg(**(dict(use_enrollments=True, scb_layers=1) if case == "small_se" else {}))
        B, L, T, M, vocab_hi, ts0 = 2, 10, 100, 80, 400, None
    else:
        kw = dict(RD.COMMON)
        kw.update(RD.DIMS["whisper-tiny"])
        cfg = MG.DiCoWConfig(**kw)
        B, L, T, M, vocab_hi, ts0 = 1, 32, 1500, 80, 50257, RD.ts_start("whisper-tiny")
    torch.manual_seed(0)
    model = MG.DiCoWForConditionalGeneration(cfg)
    hashed_init_(model)
    with torch.no_grad():
        model.model.encoder.embed_positions.weight.copy_(MG.mw.sinusoids(T, cfg.d_model))
    return cfg, model, f20_batches(case, HP[case]["K"], ts0)


def run(case, autocast):
    hp = HP[case]
    cfg, model, batches = build(case)
    model.train()
    args = types.SimpleNamespace(use_custom_optimizer=True, learning_rate=hp["lr"], fddt_lr_multiplier=hp["mult"], weight_decay=hp["wd"])
    WhisperContainer.freeze_except(types.SimpleNamespace(model=model), PREFIXES)
    opt = get_optimizer(model, args, prefixes_with_higher_lr=PREFIXES)
    sched = get_cosine_schedule_with_warmup(opt, hp["warmup"], hp["K"])
    rec = {"loss": [], "gnorm": [], "lr0": [], "lr1": [], "ntrain": []}
    warm = True
    for k, b in enumerate(batches):
        if warm and k >= hp["n_pre"]:          # the unfreeze rule (global_step == k optimizer steps taken so far)
            for name, p in model.named_parameters():
                p.requires_grad = not any(w in name for w in FROZEN)
            warm = False
        if "enrollments" in b:                 # the collator's nested enrollment batch carries an attention mask (collators.py:216-220)
            b = dict(b, enrollments=dict(b["enrollments"], attention_mask=torch.ones(b["input_features"].shape[0], b["input_features"].shape[2])))
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            out = model(**b)
        out.loss.float().backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), hp["max_norm"])
        rec["lr0"].append(opt.param_groups[0]["lr"]); rec["lr1"].append(opt.param_groups[1]["lr"])
        opt.step()
        sched.step()
        model.zero_grad()
        rec["loss"].append(float(out.loss)); rec["gnorm"].append(float(gn))
        rec["ntrain"].append(sum(p.numel() for p in model.parameters() if p.requires_grad))
        print(f"{case} {'bf16' if autocast else 'fp32'} step {k}: loss {float(out.loss):.6f} |g| {float(gn):.4f} lr {rec['lr0'][-1]:.3e}/{rec['lr1'][-1]:.3e}", flush=True)
    named = dict(model.named_parameters())
    final = {n: named[n].detach().clone() for n in WATCH if n in named}
    return cfg, rec, final


def main(which):
    for case in which:
        t0 = time.time()
        cfg, rec, final = run(case, autocast=False)
        _, rec_b, final_b = run(case, autocast=True)
        _, model0, _ = build(case)
        start = dict(model0.named_parameters())
        arrs = {"cfg": np.array(repr(MG.cfg_dict(cfg))), "hp": np.array(repr(HP[case])), "watched": np.array("\n".join(final))}
        for k, v in rec.items():
            arrs[k] = np.array(v, dtype=np.float64)
        arrs["bf16.loss"] = np.array(rec_b["loss"], dtype=np.float64)
        arrs["bf16.gnorm"] = np.array(rec_b["gnorm"], dtype=np.float64)
        for n, t in final.items():
            d = (t - start[n].detach()).float()                 # the UPDATE each watched parameter received over the run
            arrs["upd.sub." + n] = subsample(d, 512)
            arrs["upd.norm." + n] = d.double().norm().float()
            db = (final_b[n] - start[n].detach()).float()
            arrs["bf16.upd.reldev." + n] = ((db - d).double().norm() / max(1e-30, float(d.double().norm()))).float()
        MG.save("f20_trajectory_" + case, **arrs)
        print(f"{case}: {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main(sys.argv[1:] or ["small", "small_se", "tiny"])
