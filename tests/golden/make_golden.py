#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REAL REFERENCE.

Runs only in the build container (needs /root/reference, which never travels to the GPU box).
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
Fixtures are data only (inputs, parameters, expected outputs/gradients); no reference source
is copied.  Versions of torch / transformers used are recorded in every file.

Version-skew shim: the reference is written for transformers 4.55
where ``WhisperEncoderLayer.forward`` returns a tuple; the installed 5.x returns a bare tensor, so
``hidden_states = layer_outputs[0]`` (reference encoder.py:223) would silently index the batch.
We wrap the HF layer forward to return ``(hidden_states,)`` again.

Import-time placeholders: ``src/data/local_datasets.py`` imports lhotse / torchaudio / omegaconf
... at module scope; those packages are absent here.  We register empty placeholder modules so
the module body executes; the only function we then call (``_create_stno_masks``) is pure numpy
and touches none of them.
"""
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
REF = "/root/reference/src"
sys.path.insert(0, REF)
sys.path.insert(1, os.path.dirname(REF))   # utils/general.py does `from src.utils...`
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np
import torch
import transformers
import transformers.models.whisper.modeling_whisper as mw

_orig_layer_fwd = mw.WhisperEncoderLayer.forward


def _tuple_fwd(self, hidden_states, attention_mask=None, layer_head_mask=None, output_attentions=False, **kw):
    return (_orig_layer_fwd(self, hidden_states, attention_mask, **kw),)


mw.WhisperEncoderLayer.forward = _tuple_fwd

from models.dicow.modeling_dicow import DiCoWForConditionalGeneration  # noqa: E402
from models.dicow.config import DiCoWConfig  # noqa: E402
from models.dicow.FDDT import FDDT  # noqa: E402
from models.dicow.layers import SpeakerCommunicationBlock  # noqa: E402

VERS = {"torch": torch.__version__, "transformers": transformers.__version__, "numpy": np.__version__}


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    out["_versions"] = np.array(repr(VERS))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  {os.path.getsize(path) / 1e6:.2f} MB")


# ----------------------------------------------------------------------------- F1: STNO builder
def f1_stno():
    import importlib.abc
    import importlib.machinery

    absent = ("lhotse", "torchaudio", "omegaconf", "wandb", "hydra", "peft", "meeteval", "jiwer")

    class _Any(types.ModuleType):
        __path__ = []

        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return type(k, (), {})

    class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, name, path=None, target=None):
            if name.split(".")[0] in absent:
                return importlib.machinery.ModuleSpec(name, self, is_package=True)
            return None

        def create_module(self, spec):
            return _Any(spec.name)

        def exec_module(self, module):
            pass

    sys.meta_path.append(_Finder())
    try:
        from data.local_datasets import TS_ASR_DatasetSuperclass
        fn = TS_ASR_DatasetSuperclass._create_stno_masks
    except Exception as ex:  # pragma: no cover
        raise SystemExit(f"cannot import reference STNO builder: {ex!r}")
    rng = np.random.default_rng(1)
    arrs = {}
    case = 0
    for S in (1, 2, 3, 4):
        for s_index in list(range(S)) + [-1]:
            T = 150
            a = rng.random((S, T)).astype(np.float32)
            a[rng.random((S, T)) < 0.3] = 0.0          # exact silences
            a[rng.random((S, T)) < 0.3] = 1.0          # exact activity
            if s_index == -1:                          # unknown speaker: caller appends a zero row
                a = np.pad(a, ((0, 1), (0, 0)))
            arrs[f"in_{case}"] = a
            arrs[f"idx_{case}"] = np.array(s_index)
            arrs[f"out_{case}"] = fn(a.copy(), s_index)
            case += 1
    arrs["n_cases"] = np.array(case)
    save("f1_stno", **arrs)


# ----------------------------------------------------------------------------- F2: log-mel
def synth_wave(seed, seconds):
    rng = np.random.default_rng(seed)
    n = int(seconds * 16000)
    t = np.arange(n) / 16000.0
    w = 0.3 * np.sin(2 * np.pi * (220.0 + 40 * seed) * t) + 0.1 * np.sin(2 * np.pi * 3100.0 * t) * (t > 0.5)
    w = w + 0.05 * rng.standard_normal(n)
    return np.round(w * 8192).astype(np.int16)         # stored as int16, used as int16/32768


def f2_logmel():
    from transformers import WhisperFeatureExtractor
    arrs = {}
    for i, (mels, secs) in enumerate([(80, 3.3), (128, 7.1)]):
        fe = WhisperFeatureExtractor(feature_size=mels)
        w16 = synth_wave(i, secs)
        wave = w16.astype(np.float32) / 32768.0
        out = fe(wave, return_tensors="np", sampling_rate=16000, return_attention_mask=True, truncation=False,
                 padding="longest", pad_to_multiple_of=fe.n_samples)
        arrs[f"wave_{i}"] = w16
        arrs[f"mels_{i}"] = np.array(mels)
        arrs[f"feat_{i}"] = out["input_features"][0].astype(np.float32)
        arrs[f"attn_sum_{i}"] = np.array(int(out["attention_mask"][0].sum()))
    arrs["n_cases"] = np.array(2)
    save("f2_logmel", **arrs)


# ----------------------------------------------------------------------------- helpers
def randomize_(module, seed):
    """Overwrite every parameter with seeded values so all of them matter."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if n.endswith("layer_norm.weight") or n.endswith("_layer_norm.weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif "fddt" in n and n.endswith(".weight") and p.dim() == 1:
                base = 0.5 if ("initial_fddt" in n and ("silence" in n or "non_target" in n)) else 1.0
                p.copy_(base + 0.1 * torch.randn(p.shape, generator=g))
            elif "fddt" in n and n.endswith(".weight") and p.dim() == 2:
                p.copy_(torch.eye(p.shape[0]) + 0.3 * p.shape[0] ** -0.5 * torch.randn(p.shape, generator=g))
            elif n.endswith("gate"):
                p.copy_(0.3 + 0.5 * torch.randn(p.shape, generator=g))
            elif "embed_positions" in n and "encoder" in n:
                pass                                    # keep sinusoids
            elif p.dim() >= 2:
                fan_in = p[0].numel()
                p.copy_(fan_in ** -0.5 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))


def soft_stno(B, T, seed, hard_frac=0.3):
    g = torch.Generator().manual_seed(seed)
    s = torch.softmax(2.0 * torch.randn(B, 4, T, generator=g), dim=1)
    hard = torch.nn.functional.one_hot(torch.randint(0, 4, (B, T), generator=g), 4).permute(0, 2, 1).float()
    pick = (torch.rand(B, 1, T, generator=g) < hard_frac).float()
    s = pick * hard + (1 - pick) * s
    s[:, :, -T // 10:] = 0.0                            # padding frames = silence 1 (collators.py:157-161)
    s[:, 0, -T // 10:] = 1.0
    return s


# ----------------------------------------------------------------------------- F3: FDDT module
def f3_fddt():
    D, B, T = 128, 2, 100
    arrs = {}
    variants = {
        "diag": dict(is_diagonal=True),
        "full": dict(is_diagonal=False),
        "bias": dict(is_diagonal=True, bias_only=True),
        "diag_no_sil_ovl": dict(is_diagonal=True, use_silence=False, use_overlap=False),
        "full_no_tgt": dict(is_diagonal=False, use_target=False),
    }
    for vi, (vn, kw) in enumerate(variants.items()):
        torch.manual_seed(10 + vi)
        m = FDDT(D, non_target_rate=0.5, fddt_init="suppressive", **kw)
        randomize_(m, 20 + vi)
        if kw.get("bias_only"):
            with torch.no_grad():
                for p in m.parameters():
                    p.copy_(0.1 * torch.randn(p.shape))
        g = torch.Generator().manual_seed(30 + vi)
        h = torch.randn(B, T, D, generator=g, requires_grad=True)
        st = soft_stno(B, T, 40 + vi)
        go = torch.randn(B, T, D, generator=g)
        out = m(h.clone() if kw.get("bias_only") else h, st)      # bias_only mutates its input in place
        out.backward(go)
        arrs[f"{vn}.h"], arrs[f"{vn}.stno"], arrs[f"{vn}.gout"] = h, st, go
        arrs[f"{vn}.out"], arrs[f"{vn}.gh"] = out, h.grad
        for n, p in m.named_parameters():
            arrs[f"{vn}.p.{n}"] = p
            arrs[f"{vn}.g.{n}"] = p.grad
    save("f3_fddt", **arrs)


# ----------------------------------------------------------------------------- F4: conv stem alone, forward + backward
def f4_conv_stem():
    """The stem of the reference encoder (encoder.py:167-170: gelu(conv1(x)), gelu(conv2(.)), permute) on the reference's own
    conv modules: case "a" = small_cfg (80 mels, D = 128, 200 frames, B = 2), case "b" = 128 mels, D = 64, the full 3000 frames.
    A forward hook on the real encoder's conv2 proves the three lines below are what the reference's forward runs."""
    arrs = {}
    for cn, over, B in (("a", {}, 2), ("b", dict(num_mel_bins=128, d_model=64, encoder_ffn_dim=128, decoder_ffn_dim=128,
                                                max_source_positions=1500, encoder_layers=1, decoder_layers=1, encoder_attention_heads=1,
                                                decoder_attention_heads=1), 1)):
        cfg = small_cfg(**over)
        torch.manual_seed(4)
        model = DiCoWForConditionalGeneration(cfg).eval()
        randomize_(model, 44)
        enc = model.model.encoder
        T = cfg.max_source_positions
        g = torch.Generator().manual_seed(45)
        x = torch.randn(B, cfg.num_mel_bins, 2 * T, generator=g).clamp(-1.5, 1.5).half().float()      # stored as fp16
        go = torch.randn(B, T, cfg.d_model, generator=g)
        seen = {}
        hk = enc.conv2.register_forward_hook(lambda m, i, o: seen.__setitem__("c2", o.detach().clone()))
        with torch.no_grad():
            enc(x, stno_mask=soft_stno(B, T, 46))
        hk.remove()
        for q in (enc.conv1.weight, enc.conv1.bias, enc.conv2.weight, enc.conv2.bias):
            q.grad = None
        h1 = torch.nn.functional.gelu(enc.conv1(x))
        c2 = enc.conv2(h1)
        assert torch.equal(c2, seen["c2"]), "the replicated stem is not what the reference encoder's forward computed"
        out = torch.nn.functional.gelu(c2).permute(0, 2, 1)
        out.backward(go)
        arrs.update({f"{cn}.cfg": np.array(repr(cfg_dict(cfg))), f"{cn}.x": x.half(), f"{cn}.gout": go, f"{cn}.out": out,
                     f"{cn}.p.conv1.weight": enc.conv1.weight, f"{cn}.p.conv1.bias": enc.conv1.bias,
                     f"{cn}.p.conv2.weight": enc.conv2.weight, f"{cn}.p.conv2.bias": enc.conv2.bias,
                     f"{cn}.g.conv1.weight": enc.conv1.weight.grad, f"{cn}.g.conv1.bias": enc.conv1.bias.grad,
                     f"{cn}.g.conv2.weight": enc.conv2.weight.grad, f"{cn}.g.conv2.bias": enc.conv2.bias.grad})
        # the reference's own bf16-autocast deviation (the yard-stick of the GPU tolerances)
        for q in (enc.conv1.weight, enc.conv1.bias, enc.conv2.weight, enc.conv2.bias):
            q.grad = None
        with torch.autocast("cpu", dtype=torch.bfloat16):
            ob = torch.nn.functional.gelu(enc.conv2(torch.nn.functional.gelu(enc.conv1(x)))).permute(0, 2, 1)
        ob.float().backward(go)
        arrs[f"{cn}.bf16.out.maxdev"] = (ob.float() - out).abs().max()
        for n_, q in (("conv1.weight", enc.conv1.weight), ("conv1.bias", enc.conv1.bias), ("conv2.weight", enc.conv2.weight),
                      ("conv2.bias", enc.conv2.bias)):
            ref = arrs[f"{cn}.g.{n_}"]
            arrs[f"{cn}.bf16.g.reldev.{n_}"] = (q.grad.float() - ref).abs().max() / ref.abs().max()
    save("f4_conv_stem", **arrs)


# ----------------------------------------------------------------------------- configs
def small_cfg(**over):
    kw = dict(vocab_size=512, num_mel_bins=80, d_model=128, encoder_layers=2, encoder_attention_heads=2,
              decoder_layers=2, decoder_attention_heads=2, encoder_ffn_dim=256, decoder_ffn_dim=256,
              max_source_positions=100, max_target_positions=32, pad_token_id=500, bos_token_id=500,
              eos_token_id=500, decoder_start_token_id=501, use_fddt=True, fddt_is_diagonal=True,
              use_pre_pos_fddt=True, fddt_init="suppressive", non_target_fddt_value=0.5)
    kw.update(over)
    return DiCoWConfig(**kw)


CFG_KEYS = ["additional_self_attention_layer", "additional_layer", "pre_ctc_sub_sample", "remove_timestamps_from_ctc", "ctc_loss_reduction",
            "eos_token_id", "vocab_size", "num_mel_bins", "d_model", "encoder_layers", "encoder_attention_heads", "decoder_layers",
            "decoder_attention_heads", "encoder_ffn_dim", "decoder_ffn_dim", "max_source_positions",
            "max_target_positions", "pad_token_id", "decoder_start_token_id", "use_fddt", "fddt_is_diagonal",
            "fddt_bias_only", "fddt_use_silence", "fddt_use_target", "fddt_use_overlap", "fddt_use_non_target",
            "apply_fddt_to_n_layers", "use_pre_pos_fddt", "use_enrollments", "scb_layers", "ctc_weight"]


def cfg_dict(cfg):
    return {k: getattr(cfg, k) for k in CFG_KEYS}


class StubTokenizer:
    """Only what SoftLabelCreator needs (modeling_dicow.py:37): a vocab with <|t.tt|> tokens."""

    def __init__(self, vocab_size, ts_start, n_ts):
        self.vocab = {f"tok{i}": i for i in range(vocab_size)}
        for j in range(n_ts):
            del self.vocab[f"tok{ts_start + j}"]
            self.vocab[f"<|{0.02 * j:.2f}|>"] = ts_start + j
        self.prefix_tokens = [501]

    def get_vocab(self):
        return self.vocab


def make_inputs(cfg, B, L, seed, ts_range=None):
    g = torch.Generator().manual_seed(seed)
    Tm = 2 * cfg.max_source_positions
    x = torch.randn(B, cfg.num_mel_bins, Tm, generator=g).clamp(-1.5, 1.5)
    st = soft_stno(B, cfg.max_source_positions, seed + 1)
    lab = torch.randint(0, 400, (B, L), generator=g)
    if ts_range is not None:
        lab[:, 0] = ts_range[0] + 3
        lab[:, L // 2] = ts_range[0] + 17
        lab[0, 3] = ts_range[0] + ts_range[1] - 1
    upp = lab.clone()
    chg = torch.rand(B, L, generator=g) < 0.3
    upp[chg] = (lab[chg] + 7) % 400
    if ts_range is not None:
        upp[:, 0], upp[:, L // 2], upp[0, 3] = lab[:, 0], lab[:, L // 2], lab[0, 3]
    lab[1, L - 3:] = -100                              # padding
    upp[1, L - 3:] = -100
    return x, st, lab, upp


def run_model(model, batch, autocast=False):
    model.zero_grad(set_to_none=True)
    if autocast:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out = model(**batch)
    else:
        out = model(**batch)
    return out


# ----------------------------------------------------------------------------- F7: end-to-end small DiCoW
def f7_e2e():
    cfg = small_cfg()
    torch.manual_seed(0)
    model = DiCoWForConditionalGeneration(cfg).eval()
    randomize_(model, 7)
    x, st, lab, upp = make_inputs(cfg, B=2, L=12, seed=70, ts_range=(400, 100))
    batch = dict(input_features=x, stno_mask=st, labels=lab, upp_labels=upp)
    arrs = {"cfg": np.array(repr(cfg_dict(cfg))), "x": x, "stno": st, "labels": lab, "upp_labels": upp}
    for n, p in model.state_dict().items():
        arrs["p." + n] = p
    # hard-label fallback loss (no tokenizer)
    out = run_model(model, batch)
    out.loss.backward()
    arrs["hard.loss"], arrs["logits"], arrs["enc"] = out.loss, out.logits, out.encoder_last_hidden_state
    for n, p in model.named_parameters():
        if p.grad is not None:
            arrs["hard.g." + n] = p.grad
    # soft-label loss (stub tokenizer: timestamps at ids 400..499)
    model.set_tokenizer(StubTokenizer(cfg.vocab_size, 400, 100))
    out = run_model(model, batch)
    out.loss.backward()
    arrs["soft.loss"] = out.loss
    for n in ["model.encoder.fddts.0.target_linear.weight", "model.encoder.fddts.0.target_linear.bias",
              "model.encoder.initial_fddt.silence_linear.weight", "model.encoder.layers.0.fc1.weight",
              "model.encoder.conv1.weight", "model.decoder.layers.1.encoder_attn.q_proj.weight",
              "model.decoder.embed_tokens.weight"]:
        arrs["soft.g." + n] = dict(model.named_parameters())[n].grad
    arrs["ts_start"], arrs["ts_n"] = np.array(400), np.array(100)
    # bf16 autocast deviation (F9)
    model.soft_label_creator = None
    out_bf = run_model(model, batch, autocast=True)
    arrs["bf16.loss"], arrs["bf16.logits"] = out_bf.loss.float(), out_bf.logits.float()
    out_bf.loss.backward()       # the reference's own bf16-autocast gradient deviation (justifies GPU gradient tolerances)
    for n, p in model.named_parameters():
        if p.grad is not None and n != "proj_out.weight":
            arrs["bf16.grel." + n] = (p.grad.float() - arrs["hard.g." + n]).abs().max() / arrs["hard.g." + n].abs().max()
    save("f7_e2e_small", **arrs)


# ----------------------------------------------------------------------------- F8: end-to-end small SE-DiCoW
def f8_se():
    cfg = small_cfg(use_enrollments=True, scb_layers=1, encoder_layers=2)
    torch.manual_seed(1)
    model = DiCoWForConditionalGeneration(cfg).eval()
    randomize_(model, 8)
    x, st, lab, upp = make_inputs(cfg, B=2, L=10, seed=80)
    xe, ste, _, _ = make_inputs(cfg, B=2, L=10, seed=81)
    enr = {"input_features": xe, "stno_mask": ste, "attention_mask": torch.ones(2, 2 * cfg.max_source_positions)}
    batch = dict(input_features=x, stno_mask=st, labels=lab, upp_labels=upp, enrollments=enr)
    out = run_model(model, batch)
    out.loss.backward()
    arrs = {"cfg": np.array(repr(cfg_dict(cfg))), "x": x, "stno": st, "labels": lab, "upp_labels": upp,
            "enr.x": xe, "enr.stno": ste, "loss": out.loss, "logits": out.logits,
            "enc": out.encoder_last_hidden_state}
    for n, p in model.state_dict().items():
        arrs["p." + n] = p
    for n, p in model.named_parameters():
        if p.grad is not None and ("ca_enrolls" in n or "fddt" in n or n.endswith("conv1.weight")
                                   or "layers.0.self_attn.q_proj" in n):
            arrs["g." + n] = p.grad
    save("f8_e2e_se", **arrs)


# ----------------------------------------------------------------------------- F5: full-length (T=1500) encoder
def f5_encoder_fulllen():
    cfg = small_cfg(max_source_positions=1500, encoder_layers=1, decoder_layers=1, vocab_size=64,
                    pad_token_id=60, bos_token_id=60, eos_token_id=60, decoder_start_token_id=61)
    torch.manual_seed(2)
    model = DiCoWForConditionalGeneration(cfg).eval()
    randomize_(model, 5)
    with torch.no_grad():      # Whisper's sinusoidal table (HF modeling_whisper.sinusoids); not stored in the fixture
        model.model.encoder.embed_positions.weight.copy_(mw.sinusoids(1500, cfg.d_model))
    g = torch.Generator().manual_seed(50)
    x = torch.randn(1, 80, 3000, generator=g).clamp(-1.5, 1.5).half().float()    # stored as fp16
    st = soft_stno(1, 1500, 51).half().float()
    enc = model.model.encoder(x, stno_mask=st).last_hidden_state
    arrs = {"cfg": np.array(repr(cfg_dict(cfg))), "x": x.half(), "stno": st.half(),
            "enc_head": enc[:, :48], "enc_tail": enc[:, -48:], "enc_mean": enc.mean(dim=-1)}
    for n, p in model.state_dict().items():
        if n.startswith("model.encoder.") and "embed_positions" not in n:
            arrs["p." + n] = p
    save("f5_encoder_T1500", **arrs)


# ----------------------------------------------------------------------------- F6: SCB block alone
def f6_scb():
    cfg = small_cfg(use_enrollments=True, scb_layers=1)
    torch.manual_seed(3)
    blk = SpeakerCommunicationBlock(cfg)
    randomize_(blk, 6)
    g = torch.Generator().manual_seed(60)
    x = torch.randn(4, 100, 128, generator=g, requires_grad=True)
    go = torch.randn(4, 100, 128, generator=g)
    out = blk(x)
    out.backward(go)
    arrs = {"x": x, "gout": go, "out": out, "gx": x.grad}
    for n, p in blk.named_parameters():
        arrs["p." + n], arrs["g." + n] = p.detach().clone(), p.grad.clone()
    # gate = 0 must be an exact identity on the mixture rows
    with torch.no_grad():
        blk.cae.cross_gate.gate.zero_()
    arrs["out_gate0"] = blk(x.detach())
    save("f6_scb", **arrs)


# ----------------------------------------------------------------------------- F10: CTC auxiliary branch (recipe default)
def f10_ctc():
    cfg = small_cfg(vocab_size=2048, pad_token_id=2000, bos_token_id=2000, eos_token_id=2000, decoder_start_token_id=2001,
                    ctc_weight=0.3, pre_ctc_sub_sample=True, additional_self_attention_layer=True,
                    remove_timestamps_from_ctc=True, encoder_layers=1, decoder_layers=1)
    torch.manual_seed(4)
    model = DiCoWForConditionalGeneration(cfg).eval()
    randomize_(model, 10)
    tok = StubTokenizer(cfg.vocab_size, 600, 100)        # timestamps at 600..699 (>= first_task_token 541: removed from CTC)
    tok.prefix_tokens = [2001]
    model.set_tokenizer(tok)
    x, st, lab, upp = make_inputs(cfg, B=3, L=12, seed=90, ts_range=(600, 100))
    lab[2, 5:] = -100
    upp[2, 5:] = -100
    lab[0, 7] = lab[0, 8]                               # a repeated token (CTC needs the blank between repeats)
    upp[0, 7] = upp[0, 8]
    lab[1, 8] = 2000                                    # an eos inside the labels -> ignored by CTC
    upp[1, 8] = 2000
    batch = dict(input_features=x, stno_mask=st, labels=lab, upp_labels=upp)
    out = run_model(model, batch)
    out.loss.backward()
    enc_logits = model.get_enc_logits(out.encoder_last_hidden_state)
    arrs = {"cfg": np.array(repr(cfg_dict(cfg))), "x": x, "stno": st, "labels": lab, "upp_labels": upp, "loss": out.loss,
            "logits": out.logits, "enc": out.encoder_last_hidden_state, "enc_logits": enc_logits,
            "ts_start": np.array(600), "ts_n": np.array(100), "prefix": np.array([2001])}
    for n, p in model.state_dict().items():
        arrs["p." + n] = p
    for n, p in model.named_parameters():
        if p.grad is not None and ("encoder" in n) and n != "proj_out.weight":
            arrs["g." + n] = p.grad
    save("f10_ctc", **arrs)


def f10b_ctc_extra_layer():
    """The other CTC-branch variant (encoder.py:16-17,88-94): ``additional_layer`` -- a full extra WhisperEncoderLayer between
    the encoder output and the subsampling convolutions.  Same recipe as F10; the CTC logits are stored every 6th frame."""
    cfg = small_cfg(vocab_size=2048, pad_token_id=2000, bos_token_id=2000, eos_token_id=2000, decoder_start_token_id=2001,
                    ctc_weight=0.3, pre_ctc_sub_sample=True, additional_layer=True, additional_self_attention_layer=False,
                    remove_timestamps_from_ctc=True, encoder_layers=1, decoder_layers=1)
    torch.manual_seed(14)
    model = DiCoWForConditionalGeneration(cfg).eval()
    randomize_(model, 11)
    tok = StubTokenizer(cfg.vocab_size, 600, 100)
    tok.prefix_tokens = [2001]
    model.set_tokenizer(tok)
    x, st, lab, upp = make_inputs(cfg, B=2, L=10, seed=91, ts_range=(600, 100))
    lab[1, 6:] = -100
    upp[1, 6:] = -100
    out = run_model(model, dict(input_features=x, stno_mask=st, labels=lab, upp_labels=upp))
    out.loss.backward()
    enc_logits = model.get_enc_logits(out.encoder_last_hidden_state)
    arrs = {"cfg": np.array(repr(cfg_dict(cfg))), "x": x, "stno": st, "labels": lab, "upp_labels": upp, "loss": out.loss,
            "enc_logits_sub": enc_logits[:, ::6], "ts_start": np.array(600), "ts_n": np.array(100), "prefix": np.array([2001])}
    for n, p in model.state_dict().items():
        arrs["p." + n] = p
    for n, p in model.named_parameters():
        if p.grad is not None and ("encoder" in n) and n != "proj_out.weight":
            arrs["g." + n] = p.grad
    save("f10b_ctc_extra_layer", **arrs)


# ----------------------------------------------------------------------------- F11: collator augmentations
def _install_placeholders():
    import importlib.abc
    import importlib.machinery
    absent = ("lhotse", "torchaudio", "omegaconf", "wandb", "hydra", "peft", "meeteval", "jiwer")

    class _Any(types.ModuleType):
        __path__ = []

        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return type(k, (), {})

    class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, name, path=None, target=None):
            if name.split(".")[0] in absent:
                return importlib.machinery.ModuleSpec(name, self, is_package=True)
            return None

        def create_module(self, spec):
            return _Any(spec.name)

        def exec_module(self, module):
            pass

    if not any(type(f).__name__ == "_Finder" for f in sys.meta_path):
        sys.meta_path.append(_Finder())


def hashed_mel(B, M, T):
    """Deterministic pseudo-random mel in [-1, 1] from integer arithmetic only (exact in fp32 on every platform)."""
    b = np.arange(B, dtype=np.int64)[:, None, None]
    m = np.arange(M, dtype=np.int64)[None, :, None]
    t = np.arange(T, dtype=np.int64)[None, None, :]
    h = (m * 7919 + t * 104729 + b * 1299709 + (m * t) % 613 * 31) % 2001 - 1000
    return torch.from_numpy((h.astype(np.float64) / 1000.0).astype(np.float32))


def f11_augment():
    """STNO Gaussian noise + renormalisation, STNO soft segment augmentation and the joint SpecAug over mel (+) STNO of
    the reference collator (src/data/collators.py:50-138, 209-214; src/data/augmentations.py), driven by the global torch
    CPU generator: every case stores the seed, the inputs and the reference outputs."""
    _install_placeholders()
    from data.collators import DataCollator
    from data.augmentations import SpecAug
    arrs = {}
    g = torch.Generator().manual_seed(123)
    B, T = 6, 1500
    stno = torch.softmax(torch.randn(B, 4, T, generator=g) * 2.0, dim=1)
    stno[1, :, 700:] = torch.tensor([1.0, 0, 0, 0])[:, None]          # padding-as-silence tail
    stno[3] = torch.nn.functional.one_hot(torch.randint(0, 4, (T,), generator=g), 4).T.float()   # hard labels
    arrs["stno"] = stno
    # (a) Gaussian noise + rescale: recipe values var 0.2 on 75 % of the rows, and an edge case (fraction -> 0 rows)
    for i, (var, frac, seed) in enumerate(((0.2, 0.75, 11), (0.05, 0.5, 12), (0.2, 0.1, 13))):
        torch.manual_seed(seed)
        out = DataCollator.add_gaussian_noise_and_rescale(stno.clone(), var, frac)
        arrs[f"noise_{i}_cfg"] = np.array([var, frac, seed], dtype=np.float64)
        arrs[f"noise_{i}_out"] = out
    arrs["n_noise"] = np.array(3)
    # (b) soft segment augmentation: recipe values change_prob 0.1, lengths 5..50, plus a dense case
    for i, (cp, lo, hi, seed) in enumerate(((0.1, 5, 50, 21), (0.6, 3, 12, 22))):
        torch.manual_seed(seed)
        out = DataCollator.soft_segment_augmentation(stno.clone(), change_prob=cp, min_seg_len=lo, max_seg_len=hi)
        arrs[f"seg_{i}_cfg"] = np.array([cp, lo, hi, seed], dtype=np.float64)
        arrs[f"seg_{i}_out"] = out
    arrs["n_seg"] = np.array(2)
    # (c) joint SpecAug exactly as the collator wires it (collators.py:209-214), M = 128 and M = 80 mel bins
    spec_params = dict(apply_time_warp=True, time_warp_window=5, time_warp_mode="bicubic", apply_freq_mask=True,
                       freq_mask_width_range=[0, 27], num_freq_mask=2, apply_time_mask=True,
                       time_mask_width_ratio_range=[0, 0.05], num_time_mask=5)
    aug = SpecAug(**spec_params)
    for i, (M, Bs, Tm, seed) in enumerate(((128, 1, 3000, 31), (80, 2, 600, 32), (128, 2, 400, 33))):
        mel = hashed_mel(Bs, M, Tm)                       # regenerated by the tests from the same integer formula
        st = stno[:Bs, :, :Tm // 2].clone()
        torch.manual_seed(seed)
        x = torch.concatenate([mel, st.repeat_interleave(2, dim=2)], dim=1).permute(0, 2, 1)
        y = aug(x)[0].permute(0, 2, 1)
        st_out = torch.stack(y[:, M:, :].split(2, dim=-1)).mean(dim=-1).permute(1, 2, 0)
        arrs[f"spec_{i}_cfg"] = np.array([M, Bs, Tm, seed])
        arrs[f"spec_{i}_mel_out"] = y[:, :M, :]
        arrs[f"spec_{i}_stno_out"] = st_out
    arrs["n_spec"] = np.array(3)
    save("f11_augment", **arrs)


# ----------------------------------------------------------------------------- F12: STNO seek windows of long-form decoding
def f12_seek():
    """DiCoWGenerationMixin.prepare_kwargs_for_generate (src/models/dicow/generation.py:73-118) called as a plain function
    on a stand-in `self` that carries only what it reads (conv strides, max_source_positions, use_enrollments)."""
    from models.dicow.generation import DiCoWGenerationMixin
    ns = types.SimpleNamespace
    arrs = {}
    g = torch.Generator().manual_seed(5)
    cases = [  # (max_source_positions, total stno frames, batch, seek (feature frames), max_frames, batch_idx_map)
        (1500, 4500, 3, [0, 3000, 6000], [9000, 9000, 7200], [0, 1, 2]),
        (1500, 4500, 2, [2000, 0, 7400], [9000, 9000, 8000], [2, 0]),
        (100, 260, 4, [0, 100, 300, 440], [520, 520, 520, 470], [3, 1, 2, 0]),
    ]
    for i, (msp, tot, cur, seek, maxf, bmap) in enumerate(cases):
        Bfull = len(seek)
        stno = torch.softmax(torch.randn(Bfull, 4, tot, generator=g), dim=1)
        me = ns(model=ns(encoder=ns(conv1=ns(stride=(1,)), conv2=ns(stride=(2,)))),
                config=ns(max_source_positions=msp, use_enrollments=False), stno_mask_seek=None)
        kwargs = {"stno_mask": stno.clone()}
        att = torch.ones(Bfull, 2 * tot, dtype=torch.long)
        out_kw, out_att = DiCoWGenerationMixin.prepare_kwargs_for_generate(
            me, torch.tensor(maxf), cur, torch.tensor(bmap), torch.tensor(seek), kwargs, att)
        arrs[f"c{i}.stno"], arrs[f"c{i}.out"] = stno, out_kw["stno_mask"]
        arrs[f"c{i}.seek"], arrs[f"c{i}.max_frames"], arrs[f"c{i}.map"] = np.array(seek), np.array(maxf), np.array(bmap)
        arrs[f"c{i}.msp"], arrs[f"c{i}.att_rows"] = np.array(msp), out_att.shape[0]
    arrs["n_cases"] = np.array(len(cases))
    save("f12_seek", **arrs)


# ----------------------------------------------------------------------------- F13: CTC prefix scoring / joint decoding
def f13_ctc_prefix():
    """CTCPrefixScore and CTCRescorerLogitsProcessor of src/models/dicow/decoding.py driven for a few greedy steps on random
    encoder logits: every call's inputs, scores and states are stored."""
    from models.dicow.decoding import CTCPrefixScore, CTCRescorerLogitsProcessor
    arrs = {}
    g = torch.Generator().manual_seed(13)
    # (a) the scorer alone: 3 chained calls, ragged prefix lengths, a finished row dropped in the last call
    B, Tn, V, C = 4, 37, 24, 7
    blank, eos = V - 1, 20
    x = torch.log_softmax(torch.randn(B, Tn, V, generator=g) * 2.0, dim=-1)
    sc = CTCPrefixScore(x, blank, eos)
    r, s0 = sc.initial_state()
    arrs["a.x"], arrs["a.blank"], arrs["a.eos"], arrs["a.r0"] = x, np.array(blank), np.array(eos), r
    y = torch.full((B, 1), blank, dtype=torch.long)
    dl = torch.zeros(B, dtype=torch.long)
    for step in range(3):
        cs = torch.stack([torch.randperm(V - 1, generator=g)[:C] for _ in range(B)])
        cs[:, -1] = eos
        cs[0, 0] = int(y[0, -1]) if step else cs[0, 0]           # a candidate equal to the last label (repeat rule)
        active = torch.ones(B, dtype=torch.bool)
        if step == 2:
            active[1] = False
        psi, rr = sc(y[active], cs[active], dl[active], active, r[active])
        arrs[f"a.{step}.y"], arrs[f"a.{step}.cs"], arrs[f"a.{step}.dl"] = y.clone(), cs, dl.clone()
        arrs[f"a.{step}.active"], arrs[f"a.{step}.r_prev"] = active, r.clone()
        arrs[f"a.{step}.psi"], arrs[f"a.{step}.r"] = psi.clone(), rr.clone()
        # extend rows 0..2 by a non-eos candidate; row 3 keeps its prefix (a timestamp step in the real decoder)
        pick = torch.tensor([1, 2, 0, 0])
        nr = r.clone()
        idx = torch.nonzero(active)[:, 0]
        for j, b in enumerate(idx.tolist()):
            if b != 3:
                nr[b] = rr[j, :, :, pick[b]]
        newtok = cs[torch.arange(B), pick]
        y = torch.cat([y, torch.where(torch.arange(B) == 3, y[:, -1], newtok)[:, None]], dim=1)
        dl = dl + (torch.arange(B) != 3).long()
        r = nr
    arrs["a.steps"] = np.array(3)

    # (b) the logits processor with a stand-in tokenizer, greedy (num_beams 1), 4 steps
    B, Tn, V = 3, 29, 41                                             # encoder logits have V + 1 entries (blank last)
    ts0, eos_id, bos_id, pad_id = 30, 28, 29, 28
    tok = types.SimpleNamespace(upper_cased_tokens={3: 13, 4: 14}, prefix_tokens=[bos_id, 27],
                                get_vocab=lambda: {"<|0.00|>": ts0}, eos_token_id=eos_id, vocab={"#": 0})
    enc_logits = torch.randn(B, Tn, V + 1, generator=g) * 2.0
    proc = CTCRescorerLogitsProcessor(enc_logits.clone(), torch.full((B,), Tn), V, pad_id, eos_id, bos_id, tok, 0, 0.3, 1, False,
                                      ctc_tokens_to_score=6)
    arrs["b.enc_logits"] = enc_logits
    arrs["b.cfg"] = np.array([V, ts0, eos_id, bos_id, pad_id, 6])
    arrs["b.upper"], arrs["b.prefix"], arrs["b.weight"] = np.array([[3, 13], [4, 14]]), np.array([bos_id, 27]), np.array(0.3)
    ids = torch.tensor([[bos_id, 27]] * B)
    forced = [None, None, torch.tensor([31, -1, -1]), None]          # step 2: row 0 emits a timestamp token
    for step in range(4):
        scores = torch.log_softmax(torch.randn(B, V, generator=g) * 1.5, dim=-1)
        out = proc(ids, scores.clone())
        nxt = out.argmax(-1)
        if forced[step] is not None:
            nxt = torch.where(forced[step] >= 0, forced[step], nxt)
        if step == 3:
            nxt[1] = eos_id
        proc.update_state(nxt, torch.arange(B))
        arrs[f"b.{step}.ids"], arrs[f"b.{step}.scores"], arrs[f"b.{step}.out"], arrs[f"b.{step}.next"] = ids.clone(), scores, out, nxt
        arrs[f"b.{step}.state"], arrs[f"b.{step}.score_prev"] = proc.ctc_state_prev.clone(), proc.ctc_score_prev.clone()
        ids = torch.cat([ids, nxt[:, None]], dim=1)
    # one more call after row 1 ended with eos (it is skipped by the scorer)
    scores = torch.log_softmax(torch.randn(B, V, generator=g) * 1.5, dim=-1)
    arrs["b.4.ids"], arrs["b.4.scores"], arrs["b.4.out"] = ids.clone(), scores, proc(ids, scores.clone())
    arrs["b.steps"] = np.array(4)
    save("f13_ctc_prefix", **arrs)


# ----------------------------------------------------------------------------- F14: timestamp rules logits processor
def f14_timestamp_rules():
    """WhisperTimeStampLogitsProcessorCustom (src/models/dicow/utils.py:5-14 on top of HF's WhisperTimeStampLogitsProcessor)
    on crafted prefixes: first step, after text, after one / two timestamps, timestamp-heavy scores, no initial cap."""
    from models.dicow.utils import WhisperTimeStampLogitsProcessorCustom
    arrs = {}
    g = torch.Generator().manual_seed(14)
    V, eos, no_ts, ts0, begin = 120, 50, 59, 60, 3
    prompt = [51, 52, 53]
    seqs = [[], [5], [5, 7], [61], [61, 63], [5, 61], [5, 7, 61, 61], [5, 61, 61, 9], [5, 61, 61, 9, 70], [5, 61, 61, 9, 70, 70]]
    n = 0
    for max_init in (25, None):
        gc = types.SimpleNamespace(eos_token_id=eos, no_timestamps_token_id=no_ts, max_initial_timestamp_index=max_init,
                                   forced_decoder_ids=None)
        proc = WhisperTimeStampLogitsProcessorCustom(gc, begin_index=begin)
        by_len = {}
        for q in seqs:
            by_len.setdefault(len(q), []).append(q)
        for L, rows in by_len.items():
            ids = torch.tensor([prompt + q for q in rows])
            for boost in (0.0, 6.0):
                sc = torch.randn(len(rows), V, generator=g) * 2.0
                sc[:, ts0:] += boost                                    # boost: the timestamp mass beats every text token
                out = proc(ids, sc.clone())
                arrs[f"c{n}.ids"], arrs[f"c{n}.scores"], arrs[f"c{n}.out"] = ids, sc, out
                arrs[f"c{n}.max_init"] = np.array(-1 if max_init is None else max_init)
                n += 1
    arrs["n_cases"] = np.array(n)
    arrs["cfg"] = np.array([V, eos, no_ts, ts0, begin])
    save("f14_timestamp_rules", **arrs)


# ----------------------------------------------------------------------------- F15: beam-search bookkeeping
def f15_beam_search():
    """The beam bookkeeping the reference's `_beam_search` (src/models/dicow/generation.py:815-1153) performs each step through
    the helpers it inherits from transformers' GenerationMixin (_get_top_k_continuations, _get_running_beams_for_next_iteration,
    _update_finished_beams, _check_early_stop_heuristic, _beam_search_has_unfinished_sequences), driven here in the reference's
    order on synthetic per-step log-probabilities.  Every state tensor after every step and the final selection are stored."""
    from transformers.generation.utils import GenerationMixin as G
    me = types.SimpleNamespace(_gather_beams=G._gather_beams)
    arrs = {}
    cases = [dict(B=2, K=3, V=17, P=2, max_length=9, eos=5, length_penalty=1.0, early_stopping=False, seed=1, eos_boost=2.0),
             dict(B=3, K=5, V=29, P=3, max_length=12, eos=7, length_penalty=1.0, early_stopping=True, seed=2, eos_boost=3.5),
             dict(B=1, K=4, V=11, P=1, max_length=7, eos=2, length_penalty=0.6, early_stopping="never", seed=3, eos_boost=1.0)]
    for ci, c in enumerate(cases):
        B, K, V, P, max_length, eos = c["B"], c["K"], c["V"], c["P"], c["max_length"], c["eos"]
        lp_, es = c["length_penalty"], c["early_stopping"]
        g = torch.Generator().manual_seed(c["seed"])
        beams_to_keep = 2 * K
        top_mask = torch.cat((torch.ones(K, dtype=torch.bool), torch.zeros(beams_to_keep - K, dtype=torch.bool)))
        cur_len = P
        running_sequences = torch.full((B, K, max_length), eos, dtype=torch.int64)
        running_sequences[:, :, :P] = torch.randint(8, V, (B, 1, P), generator=g).expand(B, K, P)
        sequences = running_sequences.clone()
        running_beam_scores = torch.zeros(B, K)
        running_beam_scores[:, 1:] = -1e9
        beam_scores = torch.full((B, K), -1e9)
        is_sent_finished = torch.zeros(B, K, dtype=torch.bool)
        unsat = torch.ones(B, 1, dtype=torch.bool)
        running_beam_indices = torch.full((B, K, max_length - P), -1, dtype=torch.int32)
        beam_indices = running_beam_indices.clone()
        arrs[f"c{ci}.cfg"] = np.array([B, K, V, P, max_length, eos])
        arrs[f"c{ci}.length_penalty"] = np.array(lp_)
        arrs[f"c{ci}.early_stopping"] = np.array({False: 0, True: 1, "never": 2}[es])
        arrs[f"c{ci}.prompt"] = running_sequences[:, 0, :P].clone()
        step = 0
        finished = False
        while not finished:
            logits = torch.randn(B * K, V, generator=g) * 2.0
            logits[:, eos] += c["eos_boost"] * (step - 1)               # eos becomes attractive as decoding goes on
            log_probs = torch.log_softmax(logits, dim=-1)
            arrs[f"c{ci}.s{step}.log_probs"] = log_probs.clone()
            lp = log_probs.view(B, K, V) + running_beam_scores[:, :, None]
            lp = lp.reshape(B, K * V)
            tk_lp, tk_seq, tk_idx = G._get_top_k_continuations(me, accumulated_log_probs=lp, running_sequences=running_sequences,
                                                               running_beam_indices=running_beam_indices, cur_len=cur_len,
                                                               decoder_prompt_len=P, do_sample=False, beams_to_keep=beams_to_keep,
                                                               num_beams=K, vocab_size=V, batch_size=B)
            hits = (tk_seq[:, :, cur_len] == eos) | (cur_len + 1 >= max_length)       # EosTokenCriteria | MaxLengthCriteria
            running_sequences, running_beam_scores, running_beam_indices = G._get_running_beams_for_next_iteration(
                me, topk_log_probs=tk_lp, topk_running_sequences=tk_seq, topk_running_beam_indices=tk_idx,
                next_token_hits_stopping_criteria=hits, num_beams=K)
            sequences, beam_scores, beam_indices, is_sent_finished = G._update_finished_beams(
                me, sequences=sequences, topk_running_sequences=tk_seq, beam_scores=beam_scores, topk_log_probs=tk_lp,
                beam_indices=beam_indices, topk_running_beam_indices=tk_idx, is_early_stop_heuristic_unsatisfied=unsat,
                is_sent_finished=is_sent_finished, next_token_hits_stopping_criteria=hits, top_num_beam_mask=top_mask,
                num_beams=K, cur_len=cur_len, decoder_prompt_len=P, length_penalty=lp_, early_stopping=es)
            beam_idx = running_beam_indices[..., cur_len - P].flatten()
            cur_len += 1
            unsat = G._check_early_stop_heuristic(is_early_stop_heuristic_unsatisfied=unsat, running_beam_scores=running_beam_scores,
                                                  beam_scores=beam_scores, is_sent_finished=is_sent_finished, cur_len=cur_len,
                                                  max_length=max_length, decoder_prompt_len=P, early_stopping=es, length_penalty=lp_)
            finished = not bool(G._beam_search_has_unfinished_sequences(unsat, is_sent_finished, hits, es))
            for nm, t in (("running_sequences", running_sequences), ("running_beam_scores", running_beam_scores),
                          ("sequences", sequences), ("beam_scores", beam_scores), ("is_sent_finished", is_sent_finished),
                          ("unsat", unsat), ("beam_idx", beam_idx)):
                arrs[f"c{ci}.s{step}.{nm}"] = t.clone()
            step += 1
        arrs[f"c{ci}.steps"] = np.array(step)
        max_gen = int(((beam_indices[:, :1] + 1).bool()).sum(dim=2).max())
        arrs[f"c{ci}.final"] = sequences[:, 0, :P + max_gen].clone()
        arrs[f"c{ci}.final_scores"] = beam_scores[:, 0].clone()
    arrs["n_cases"] = np.array(len(cases))
    save("f15_beam_search", **arrs)


# ----------------------------------------------------------------------------- F16: long-form segment retrieval
def f16_retrieve_segment():
    """DiCoWGenerationMixin._retrieve_segment (src/models/dicow/generation.py:416-534, a static method): how one window's
    decoded tokens become segments and how far the seek pointer moves.  Crafted token sequences cover every branch."""
    from models.dicow.generation import DiCoWGenerationMixin as G
    arrs = {}
    tb = 100                                              # timestamp_begin: tokens >= 100 are timestamps (0.02 s each)
    T = lambda x: tb + x
    cases = [
        [T(0), 5, 6, T(50), T(50), 7, 8, T(120), T(120), 9, T(400)],        # two closed segments + an open one -> seek to 120
        [T(0), 5, 6, T(50), T(50), 7, 8, T(120)],                           # single timestamp ending -> whole window consumed
        [T(10), 5, 6, 7, T(90), T(90), 8, 9, T(200), T(200)],               # ends on a timestamp pair
        [5, 6, 7, 8],                                                       # no timestamps at all
        [T(30), 5, 6, 7],                                                   # one timestamp early
        [T(260), 5, 6, 7],                                                  # one timestamp late -> rollback, nothing emitted
        [T(20), 5, 6, T(80), 7, 8, T(140)],                                 # separated timestamps, no consecutive pair
        [T(0)],                                                             # a lone <|0.00|>
    ]
    time_offset = torch.tensor([12.5, 40.0], dtype=torch.float64)
    seek_num_frames = torch.tensor([3000, 1800])
    dec_ids = torch.zeros(1, 4, dtype=torch.long)
    n = 0
    for prev_idx in (0, 1):
        for seq in cases:
            try:
                segs, off = G._retrieve_segment(seek_sequence=torch.tensor(seq), seek_outputs=[None, None], time_offset=time_offset,
                                                timestamp_begin=tb, seek_num_frames=seek_num_frames, time_precision=0.02,
                                                time_precision_features=0.01, input_stride=2, prev_idx=prev_idx, idx=prev_idx,
                                                return_token_timestamps=False, decoder_input_ids=dec_ids)
            except ValueError:
                segs, off = None, -1
            arrs[f"c{n}.seq"], arrs[f"c{n}.prev"] = np.array(seq), np.array(prev_idx)
            arrs[f"c{n}.offset"] = np.array(int(off))
            arrs[f"c{n}.nseg"] = np.array(-1 if segs is None else len(segs))
            for j, sg in enumerate(segs or []):
                arrs[f"c{n}.s{j}.start"], arrs[f"c{n}.s{j}.end"] = np.array(float(sg["start"])), np.array(float(sg["end"]))
                arrs[f"c{n}.s{j}.tokens"] = np.array(sg["tokens"].tolist())
            n += 1
    arrs["n_cases"] = np.array(n)
    arrs["time_offset"], arrs["seek_num_frames"], arrs["timestamp_begin"] = time_offset, seek_num_frames, np.array(tb)
    save("f16_retrieve_segment", **arrs)


def f17_fix_timestamps():
    """DiCoWGenerationMixin._fix_timestamps_from_segmentation (src/models/dicow/generation.py:313-415): recording-time segments
    -> window-time (0..30 s) timestamp token sequences.  The method only touches ``self.tokenizer``; a stand-in tokenizer whose
    decode/encode round trip is the identity on text ids makes the result a pure function of the segment times."""
    import re
    import types
    from models.dicow.generation import DiCoWGenerationMixin as G
    TS0, FILL, PAD = 1000, 7, 0

    class Tok:
        pad_token_id = PAD

        def get_vocab(self):
            return {"<|0.00|>": TS0, "\u0120": FILL}

        def decode(self, ids):
            return "".join(f"[{int(t)}]" for t in ids if int(t) < TS0)            # Whisper's decode() drops timestamp ids

        def __call__(self, text):
            ids = []
            for m in re.finditer(r"<\|(\d+\.\d\d)\|>|\[(\d+)\]", text):
                ids.append(TS0 + int(round(float(m.group(1)) / 0.02)) if m.group(1) is not None else int(m.group(2)))
            return {"input_ids": ids}

    me = types.SimpleNamespace(tokenizer=Tok(), round_to_nearest_0_02=G.round_to_nearest_0_02)
    rng = np.random.default_rng(17)
    arrs, n = {}, 0

    def emit(segs):
        nonlocal n
        seqs = {"segments": [[{"start": torch.tensor(a, dtype=torch.float64), "end": torch.tensor(b, dtype=torch.float64),
                               "tokens": torch.tensor(t, dtype=torch.long)} for a, b, t in segs]],
                "sequences": torch.zeros(1, 1, dtype=torch.long)}
        out = G._fix_timestamps_from_segmentation(me, seqs)
        arrs[f"c{n}.start"], arrs[f"c{n}.end"] = np.array([a for a, _, _ in segs]), np.array([b for _, b, _ in segs])
        arrs[f"c{n}.ntok"] = np.array([len(t) for _, _, t in segs])
        arrs[f"c{n}.tokens"] = np.array([x for _, _, t in segs for x in t], dtype=np.int64)
        arrs[f"c{n}.ids"] = out[0].numpy()
        n += 1

    def toks():
        k = int(rng.integers(1, 5))
        return [TS0 + 3] + [int(x) for x in rng.integers(10, 900, k)] + [TS0 + 40]

    # crafted: boundary hits, exact 30 s segments, skipped blocks, late first segment, dropped segments, half-tick rounding
    emit([(0.0, 30.0, toks()), (30.0, 60.0, toks()), (60.0, 61.5, toks())])
    emit([(12.34, 42.34, toks()), (42.34, 50.0, toks()), (59.98, 60.02, toks())])
    emit([(65.0, 70.0, toks()), (200.0, 201.0, toks())])
    emit([(1.0, 2.0, []), (2.0, 3.0, [TS0]), (3.0, 29.5, toks()), (29.5, 30.5, toks()), (95.0, 125.0, toks())])
    emit([(0.01, 0.03, toks()), (0.05, 29.99, toks()), (29.99, 30.01, toks())])
    emit([(29.98, 59.98, toks()), (59.98, 89.98, toks()), (90.0, 90.5, toks())])
    emit([(10.0, 40.0, toks()), (40.0, 70.0, toks()), (70.0, 70.02, toks()), (89.98, 119.98, toks())])
    for _ in range(240):
        t, segs = float(rng.integers(0, 2500)) * 0.02 * float(rng.random() < 0.7), []
        for _ in range(int(rng.integers(1, 9))):
            r = rng.random()
            if r < 0.25:
                t += float(rng.integers(0, 4000)) * 0.02                       # silence, possibly several blocks long
            if rng.random() < 0.15:
                t = max(float(np.ceil(t / 30.0)) * 30.0 - (0.02 if rng.random() < 0.4 else 0.0), 0.0)      # sit on / just before a boundary
            r = rng.random()
            d = 30.0 if r < 0.2 else (float(np.ceil((t + 0.02) / 30.0)) * 30.0 - t if r < 0.35 else float(rng.integers(1, 1400)) * 0.02)
            a, b = round(t, 2), round(t + d, 2)
            if rng.random() < 0.1:
                a, b = a + 0.01, b + 0.01                                       # odd hundredths: the half-up rounding
            segs.append((a, b, toks() if rng.random() > 0.08 else ([TS0] if rng.random() < 0.5 else [])))
            t = b
        emit(segs)
    arrs["n_cases"], arrs["first_timestamp"], arrs["filler"], arrs["pad"] = np.array(n), np.array(TS0), np.array(FILL), np.array(PAD)
    save("f17_fix_timestamps", **arrs)


# ----------------------------------------------------------------------------- F18: temperature fallback control flow
def f18_fallback():
    """transformers' WhisperGenerationMixin.generate_with_fallback / _need_fallback / _retrieve_* (what the reference's
    generate_with_fallback, generation.py:567-611, delegates to) driven with SCRIPTED decoder outputs: fixed logits => fixed
    fallback decisions.  The stub generate() looks the active windows up by a row id planted in the prompt."""
    import copy
    from transformers import GenerationConfig
    from transformers.generation import GenerationMixin
    from transformers.generation.utils import GenerateEncoderDecoderOutput
    from transformers.generation.logits_process import WhisperNoSpeechDetection
    from transformers.models.whisper.generation_whisper import WhisperGenerationMixin

    V, eos, P, NMAX = 64, 60, 3, 48
    temps = (0.0, 0.2, 0.4, 1.0)
    g = torch.Generator().manual_seed(18)
    R = 5

    def script(row, k):
        """(tokens incl. eos, scores [n, V]) of window `row` at temperature index k."""
        gg = torch.Generator().manual_seed(1000 * row + k)
        if row == 0 or (row == 1 and k >= 2) or (row == 4 and k >= 1):      # varied tokens, confident scores
            n = 12 + row
            toks = torch.randint(0, 50, (n,), generator=gg).tolist() + [eos]
            peak = 9.0
        elif row == 1 or row == 4:                                         # repetitive (high compression ratio), confident
            toks = ([7, 8] * 20) + [eos]
            peak = 9.0
        else:                                                              # rows 2, 3: varied but unsure (low average log-probability)
            n = 10 + k
            toks = torch.randint(0, 50, (n,), generator=gg).tolist() + [eos]
            peak = 0.5
        sc = torch.randn(len(toks), V, generator=gg)
        for i, t in enumerate(toks):
            sc[i, t] += peak
        t_ = temps[k]
        if t_ > 0:
            sc = sc / t_                                                    # what the temperature warper leaves in `scores`
        return toks, sc

    class Stub(GenerationMixin):
        def generate(self, segment_input, generation_config=None, decoder_input_ids=None, **kw):
            k = self.calls
            self.calls += 1
            rows = (decoder_input_ids[:, 1] - 10).tolist()
            self.log.append((k, rows, bool(generation_config.do_sample), float(generation_config.temperature)))
            outs = [script(r, k) for r in rows]
            n = max(len(t) for t, _ in outs)
            seqs = torch.full((len(rows), P + n), eos, dtype=torch.long)
            seqs[:, :P] = decoder_input_ids
            scores = torch.zeros(n, len(rows), V)
            for i, (t, sc) in enumerate(outs):
                seqs[i, P:P + len(t)] = torch.tensor(t)
                scores[:len(t), i] = sc
            return GenerateEncoderDecoderOutput(sequences=seqs, scores=tuple(scores[i] for i in range(n)))

    class Dummy(WhisperGenerationMixin, Stub):
        pass

    d = Dummy.__new__(Dummy)
    d.config = types.SimpleNamespace(vocab_size=V, decoder_layers=1)
    d.calls, d.log = 0, []
    nsd = WhisperNoSpeechDetection.__new__(WhisperNoSpeechDetection)
    no_speech_prob = torch.tensor([0.1, 0.2, 0.3, 0.9, 0.05])
    nsd.no_speech_prob_holder = no_speech_prob
    type(nsd).no_speech_prob = property(lambda self_: self_.no_speech_prob_holder)
    nsd.set_inputs = lambda inputs: None
    gc = GenerationConfig(pad_token_id=eos, eos_token_id=eos)
    gc.no_speech_threshold, gc.compression_ratio_threshold, gc.logprob_threshold = 0.6, 2.4, -1.0
    gc.condition_on_prev_tokens, gc.cache_implementation, gc.num_beams = False, None, 1
    prompt = torch.tensor([[61, 10 + r, 62] for r in range(R)])
    seek_sequences, seek_outputs, should_skip, do_cond, _ = d.generate_with_fallback(
        segment_input=torch.zeros(R, 4, 8), decoder_input_ids=prompt, cur_bsz=R, seek=torch.zeros(R, dtype=torch.long),
        batch_idx_map=list(range(R)), temperatures=temps, generation_config=gc, logits_processor=[nsd], stopping_criteria=None,
        prefix_allowed_tokens_fn=None, synced_gpus=False, return_token_timestamps=False, do_condition_on_prev_tokens=[False] * R,
        is_shortform=False, batch_size=R, attention_mask=None, kwargs={})
    arrs = {"V": np.array(V), "eos": np.array(eos), "P": np.array(P), "temps": np.array(temps), "n_rows": np.array(R),
            "thr": np.array([gc.compression_ratio_threshold, gc.logprob_threshold, gc.no_speech_threshold]),
            "no_speech_prob": no_speech_prob, "should_skip": np.array(should_skip),
            "calls": np.array(repr([(k, rows, ds, t) for k, rows, ds, t in d.log]))}
    for r in range(R):
        arrs[f"final_{r}"] = np.array(seek_sequences[r].tolist(), dtype=np.int64)
        for k in range(len(temps)):
            t, sc = script(r, k)
            arrs[f"tok_{r}_{k}"], arrs[f"sc_{r}_{k}"] = np.array(t, dtype=np.int64), sc
            arrs[f"cr_{r}_{k}"] = np.array(WhisperGenerationMixin._retrieve_compression_ratio(torch.tensor(t), V))
            arrs[f"lp_{r}_{k}"] = np.array(float(WhisperGenerationMixin._retrieve_avg_logprobs([s_ for s_ in sc], torch.tensor(t), temps[k])))
    save("f18_fallback", **arrs)


if __name__ == "__main__":
    which = sys.argv[1:] or ["f1", "f2", "f3", "f4", "f5", "f6", "f7", "f8", "f10", "f10b", "f11", "f12", "f13", "f14", "f15", "f16", "f17", "f18"]
    fns = {"f1": f1_stno, "f2": f2_logmel, "f3": f3_fddt, "f4": f4_conv_stem, "f5": f5_encoder_fulllen, "f6": f6_scb, "f7": f7_e2e,
           "f8": f8_se, "f10": f10_ctc, "f10b": f10b_ctc_extra_layer, "f11": f11_augment, "f12": f12_seek, "f13": f13_ctc_prefix, "f14": f14_timestamp_rules, "f15": f15_beam_search, "f16": f16_retrieve_segment, "f17": f17_fix_timestamps, "f18": f18_fallback}
    for w in which:
        fns[w]()
