"""Fixture F20 on the MI355X: trainer.TrainStep (HIP forward / backward, flat gradient store, fused clip + AdamW, device-side
schedule, staged freezing) runs the eight optimizer steps of the REAL reference's training trajectory
(tests/golden/make_golden_trajectory.py: the reference's freeze / unfreeze rules, get_optimizer, transformers' cosine schedule,
clip_grad_norm_) on the same hashed weights and batches.

Tolerances (the HIP path follows the reference's bf16 AMP recipe, the fixture is fp32; it also holds the reference's own
bf16-autocast trajectory): per-step loss within max(5e-3, 3 x |bf16 - fp32| of that step); gradient norm within max(2 %, 3 x the
reference's bf16 relative deviation); learning rates to fp32 storage precision (2e-7 relative; the schedule itself is evaluated in double on the device); the UPDATE every watched parameter received over the
run within max(5e-2, 4 x the reference's own bf16 deviation) in relative L2 (the 1 % largest element differences set aside: Adam's
sign-like first update flips elements whose gradient is ~0), frozen parameters bit-unchanged."""
import ast
import math

import pytest
import torch

import amd_pkg
from tests.util import load_golden, hashed_init_, f20_batches, subsample

pytestmark = pytest.mark.gpu
amd_pkg.load()


def _build(case, z):
    import ts_asr_whisper_amd as pkg
    from ts_asr_whisper_amd.modeling import sinusoids
    if case.startswith("small"):
        d = ast.literal_eval(str(z["cfg"]))
        d.setdefault("bos_token_id", d["pad_token_id"])
        cfg, ts0 = pkg.DiCoWConfig(**d), None
    else:
        cfg = pkg.DiCoWConfig.preset("whisper-tiny", use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True,
                                     fddt_init="suppressive", non_target_fddt_value=0.5)
        ts0 = cfg.vocab_size - 1501
    torch.manual_seed(0)
    model = pkg.DiCoWForConditionalGeneration(cfg).cuda()
    hashed_init_(model)
    with torch.no_grad():
        model.model.encoder.embed_positions.weight.copy_(sinusoids(cfg.max_source_positions, cfg.d_model))
    model.tie_weights()
    return model, cfg, ts0


@pytest.mark.parametrize("case", ["small", "small_se", "tiny"])
def test_f20_training_trajectory_vs_reference(case):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ts_asr_whisper_amd.trainer import TrainStep
    z = load_golden("f20_trajectory_" + case)
    hp = ast.literal_eval(str(z["hp"]))
    model, cfg, ts0 = _build(case, z)
    start = {n: p.detach().clone() for n, p in model.named_parameters()}
    ts = TrainStep(model, lr=hp["lr"], fddt_lr_multiplier=hp["mult"], weight_decay=hp["wd"], max_grad_norm=hp["max_norm"],
                   warmup_steps=hp["warmup"], max_steps=hp["K"], frozen_keywords=("decoder",), use_fddt_only_n_steps=hp["n_pre"])
    named = dict(model.named_parameters())
    for k, b in enumerate(f20_batches(case, hp["K"], ts0)):
        loss = ts.step({n: ({m: w.cuda() for m, w in v.items()} if isinstance(v, dict) else v.cuda()) for n, v in b.items()})
        ref, bf = float(z["loss"][k]), float(z["bf16.loss"][k])
        tol = max(5e-3, 3 * abs(bf - ref))
        assert abs(float(loss) - ref) < tol, (case, k, float(loss), ref, tol)
        gn, gref, gbf = math.sqrt(float(ts.opt.gnorm_sq)), float(z["gnorm"][k]), float(z["bf16.gnorm"][k])
        assert abs(gn - gref) < max(0.02, 3 * abs(gbf - gref) / gref) * gref, (case, k, gn, gref)
        lrs = ts.opt.applied_lr()                        # one per contiguous optimizer run; the frozen runs of the preheat phase hold their last value
        pre = [r[2] for r in ts.store.runs]
        want0, want1 = float(z["lr0"][k]), float(z["lr1"][k])
        for lr, is_pre in zip(lrs, pre):
            if is_pre:
                assert abs(lr - want1) <= 2e-7 * want1 + 1e-12, (case, k, lr, want1)
            elif k >= hp["n_pre"]:
                assert abs(lr - want0) <= 2e-7 * want0 + 1e-12, (case, k, lr, want0)
        assert sum(p.numel() for p in model.parameters() if p.requires_grad) == int(z["ntrain"][k]), (case, k)
    torch.cuda.synchronize()
    for n in str(z["watched"]).split("\n"):
        upd = (named[n].detach() - start[n]).float().cpu()
        ref, nrm = torch.from_numpy(z["upd.sub." + n]), float(z["upd.norm." + n])
        if nrm == 0.0:                                   # frozen (decoder): bit-unchanged
            assert float(upd.abs().max()) == 0.0, n
            continue
        tol = max(5e-2, 4 * float(z["bf16.upd.reldev." + n]))
        # Adam's first update of an element is lr * sign(g): an element whose gradient is ~0 lands on the other side with ANY
        # rounding difference (measured: one such element in each of two 384-element bias vectors is 6 % of their relative L2,
        # the other 383 agree to 0.7 %; which elements flip changes with any change of a rounding point upstream: one to three per 384
        # have been seen).  So the largest element differences (three, or 1 % of the elements) are set aside.
        d = (subsample(upd, 512) - ref).double()
        n_out = max(3, (d.numel() + 99) // 100)
        keep = d.abs().argsort()[: d.numel() - n_out]
        r_sub = float(d[keep].norm() / ref.double().norm())
        r_nrm = abs(float(upd.double().norm()) - nrm) / nrm
        assert r_sub < tol and r_nrm < tol, (case, n, r_sub, r_nrm, tol)
        assert float(d.abs().max()) < 2.5 * float(ref.abs().max()), (case, n)      # even a flipped element moves by ~2 lr at most
