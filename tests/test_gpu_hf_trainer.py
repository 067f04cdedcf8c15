"""The drop-in boundary against the caller the reference really uses: the INSTALLED third-party ``transformers.Seq2SeqTrainer``
(reference: ``CustomTrainer(Seq2SeqTrainer)``, src/utils/trainers.py:106-139, built in src/train.py:227-262 with ``bf16``, ``predict_with_generate``,
``max_grad_norm`` and the optimizer of src/models/containers.py:100-114), not a re-statement of it.  None of the reference's Python runs here.

The trainer drives ``DiCoWForConditionalGeneration`` through everything ``src/train.py`` makes it do:
  * ``train()``: 3 optimizer steps under its bf16 autocast, gradient clipping at 1.0, a two-group ``torch.optim.AdamW`` built the way
    ``get_optimizer`` builds it, batches with the collator's keys (``input_features, stno_mask, attention_mask, labels, upp_labels``,
    src/data/collators.py:140-187);
  * ``evaluate()`` with ``predict_with_generate=True``: ``prediction_step`` -> ``model.generate(**inputs, **gen_kwargs)`` + the eval loss;
  * ``save_model()`` (``Trainer._save`` -> ``model.save_pretrained(dir, state_dict=...)``, tied head) and ``from_pretrained()`` of the result.
What must hold: the logged losses and EVERY parameter after the three steps equal those of a plain eager loop over a second copy of
the same model (same batches, same optimizer recipe, torch's own clip) -- bit for bit: the kernels are deterministic and the trainer adds
no arithmetic of its own; the reloaded checkpoint equals the trained model; generation inside ``evaluate`` returns what a direct
``generate`` call returns.  Run with `pytest -m gpu`."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
PREFIXES = ("model.encoder.fddts", "model.encoder.initial_fddt")      # configs/base.yaml:18 prefixes_to_preheat
LR, MULT = 2e-4, 100.0                                                 # (base.yaml:61 fddt lr multiplier 100; weight decay 0: base.yaml:71)
L_LAB, GEN_LEN = 10, 12


def _cfg(pkg):
    return pkg.DiCoWConfig(vocab_size=300, d_model=128, encoder_layers=2, encoder_attention_heads=2, decoder_layers=2,
                           decoder_attention_heads=2, encoder_ffn_dim=256, decoder_ffn_dim=256, num_mel_bins=80,
                           max_source_positions=1500, max_target_positions=32, pad_token_id=1, bos_token_id=1, eos_token_id=2,
                           decoder_start_token_id=3, use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True,
                           fddt_init="suppressive", non_target_fddt_value=0.5)


def _samples(cfg, n):
    """n items as the reference's dataset + collator hand them over (CPU tensors; the trainer moves them)."""
    from ts_asr_whisper_amd.data import synthetic_batch
    out = []
    for i in range(n):
        b = synthetic_batch(cfg, 1, L_LAB, seed=500 + i, device="cpu")
        lab = b["labels"][0].clone()
        lab[0] = cfg.decoder_start_token_id
        if i % 2:
            lab[-3:] = -100                                            # padded label rows, as the collator pads them
        upp = lab.clone()
        upp[2] = (int(upp[2]) + 7) % 200 + 4                           # an "upper-cased first letter" alternative (collators.py:181-186)
        out.append({"input_features": b["input_features"][0], "stno_mask": b["stno_mask"][0].float(),
                    "attention_mask": torch.ones(2 * cfg.max_source_positions, dtype=torch.long), "labels": lab, "upp_labels": upp})
    return out


class _Stream(torch.utils.data.IterableDataset):
    """Items in a fixed order, over and over (an IterableDataset: the trainer's sampler does not shuffle it)."""

    def __init__(self, items):
        self.items = items

    def __iter__(self):
        while True:
            yield from self.items


def _collate(items):
    return {k: torch.stack([it[k] for it in items]) for k in items[0]}


def _optimizer(model):
    """reference src/models/containers.py:100-114 (`get_optimizer`): everything not under the preheat prefixes at the base rate, the
    prefixed (FDDT) parameters at rate x multiplier; weight decay 0."""
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    base = [p for n, p in named if not n.startswith(PREFIXES)]
    new = [p for n, p in named if n.startswith(PREFIXES)]
    return torch.optim.AdamW([{"params": base}, {"params": new, "lr": MULT * LR, "weight_decay": 0.0}], lr=LR, weight_decay=0.0)


def _build(pkg, cfg):
    from ts_asr_whisper_amd.trainer import freeze_by_keyword
    torch.manual_seed(11)
    model = pkg.DiCoWForConditionalGeneration(cfg)
    model.post_init()                                                  # containers.py:52
    freeze_by_keyword(model, ("decoder",))                             # dicow_v3.yaml:6-7 via containers.py:80-90
    return model


def test_installed_seq2seq_trainer_trains_evaluates_saves_and_reloads(tmp_path):
    import transformers
    from transformers import Seq2SeqTrainer, Seq2SeqTrainingArguments, PreTrainedModel, GenerationConfig
    import amd_pkg
    pkg = amd_pkg.load()
    cfg = _cfg(pkg)
    items = _samples(cfg, 4)

    # ---- the eager twin first: plain loop, torch's clip, the same optimizer recipe, the trainer's autocast
    twin = _build(pkg, cfg).cuda()
    opt2 = _optimizer(twin)
    eager_losses = []
    for step in range(3):
        batch = {k: v.cuda() for k, v in _collate([items[(2 * step) % 4], items[(2 * step + 1) % 4]]).items()}
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = twin(**batch).loss
        loss.backward()
        torch.nn.utils.clip_grad_norm_(twin.parameters(), 1.0)
        opt2.step()
        opt2.zero_grad()
        eager_losses.append(float(loss))

    # ---- the installed trainer
    model = _build(pkg, cfg)
    assert isinstance(model, PreTrainedModel) and isinstance(model.generation_config, GenerationConfig)
    assert model.config.to_json_string() and type(model).main_input_name == "input_features" and model.can_generate()
    model.generation_config.max_length = GEN_LEN                       # (src/train.py:195,219,246 edit model.generation_config in place)
    args = Seq2SeqTrainingArguments(output_dir=str(tmp_path / "out"), per_device_train_batch_size=2, per_device_eval_batch_size=2,
                                    max_steps=3, learning_rate=LR, lr_scheduler_type="constant", weight_decay=0.0, max_grad_norm=1.0,
                                    bf16=True, predict_with_generate=True, generation_max_length=GEN_LEN, logging_steps=1,
                                    save_strategy="no", eval_strategy="no", report_to="none", remove_unused_columns=False,
                                    dataloader_num_workers=0, dataloader_pin_memory=False, seed=0, disable_tqdm=True)
    seen = {}

    def compute_metrics(pred):
        seen["pred"], seen["labels"] = pred.predictions, pred.label_ids
        return {"tokens": float((pred.predictions != cfg.pad_token_id).sum())}

    trainer = Seq2SeqTrainer(model=model, args=args, train_dataset=_Stream(items), eval_dataset=items, data_collator=_collate,
                             optimizers=(_optimizer(model), None), compute_metrics=compute_metrics)
    out = trainer.train()
    assert out.global_step == 3
    logged = [h["loss"] for h in trainer.state.log_history if "loss" in h]
    assert len(logged) == 3
    for a, b in zip(logged, eager_losses):                             # (the trainer rounds what it logs to 4 decimals)
        assert abs(a - b) < 6e-5, (logged, eager_losses)
    assert trainer.state.total_flos > 0                                # floating_point_ops() of the model was asked and answered
    for (n, p), (_, q) in zip(model.named_parameters(), twin.named_parameters()):
        assert torch.equal(p.detach(), q.detach()), n                  # every parameter, bit for bit
    moved = [n for (n, p), (_, q) in zip(model.named_parameters(), _build(pkg, cfg).named_parameters()) if not torch.equal(p.detach().cpu(), q)]
    assert any(n.startswith(PREFIXES) for n in moved) and any("layers.0.fc1" in n for n in moved) and not any("decoder" in n for n in moved)

    # ---- evaluate(): prediction_step -> model.generate(**inputs, max_length=..., synced_gpus=...) + the loss of model(**inputs)
    metrics = trainer.evaluate()
    assert metrics["eval_loss"] > 0 and metrics["eval_tokens"] > 0
    assert seen["pred"].shape == (4, GEN_LEN) and seen["labels"].shape[0] == 4
    batch = {k: v.cuda() for k, v in _collate(items[:2]).items()}
    direct = model.generate(input_features=batch["input_features"], stno_mask=batch["stno_mask"], max_length=GEN_LEN).cpu()
    got = torch.as_tensor(seen["pred"][:2, :direct.shape[1]])
    assert torch.equal(got, direct)                                    # what the trainer collected is what generate returns
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        l01 = float(model(**batch).loss)
        l23 = float(model(**{k: v.cuda() for k, v in _collate(items[2:]).items()}).loss)
    assert abs(metrics["eval_loss"] - 0.5 * (l01 + l23)) < 1e-5

    # ---- save_model() -> from_pretrained(): config.json + weights (the tied head once) + generation_config.json
    trainer.save_model(str(tmp_path / "saved"))
    files = set(os.listdir(tmp_path / "saved"))
    assert {"config.json", "model.safetensors", "generation_config.json", "training_args.bin"} <= files
    again = pkg.DiCoWForConditionalGeneration.from_pretrained(str(tmp_path / "saved"))
    assert again._load_report == {"missing": [], "unexpected": []}
    assert again.proj_out.weight is again.model.decoder.embed_tokens.weight
    assert again.generation_config.max_length == GEN_LEN
    for (n, p), (_, q) in zip(model.state_dict().items(), again.state_dict().items()):
        assert torch.equal(p.cpu(), q), n
    assert again.config.hot_path_dict() == model.config.hot_path_dict()
    print("transformers", transformers.__version__, "losses", logged, "eval", {k: round(v, 4) for k, v in metrics.items() if isinstance(v, float)})
