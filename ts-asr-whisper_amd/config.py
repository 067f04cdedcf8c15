"""DiCoWConfig: the hot-path switches of the reference's ``DiCoWConfig(WhisperConfig)``
(reference src/models/dicow/config.py:6-59) plus the Whisper dimensions they ride on.

Like the reference's class it IS a ``transformers.PretrainedConfig`` when transformers can be imported (the reference's caller --
``transformers.Seq2SeqTrainer`` via src/utils/trainers.py:106-139, ``save_model`` / ``from_pretrained`` -- needs
``to_json_string`` / ``to_dict`` / ``save_pretrained`` / ``from_pretrained`` of its model's config); without transformers it is a plain
attribute container.  ``from_hf`` accepts a HuggingFace ``WhisperConfig`` / the reference's ``DiCoWConfig`` / a plain dict so that
checkpoints' ``config.json`` load.
"""
import inspect
from typing import Optional

try:                                        # the reference's DiCoWConfig(WhisperConfig) is a PretrainedConfig; so is this one
    from transformers import PretrainedConfig as _ConfigBase
    _HF = True
except Exception:                           # pragma: no cover -- transformers is part of the reference's own requirements
    _ConfigBase, _HF = object, False

_WHISPER_DEFAULTS = dict(
    vocab_size=51865, num_mel_bins=80, d_model=384, encoder_layers=4, encoder_attention_heads=6, decoder_layers=4,
    decoder_attention_heads=6, encoder_ffn_dim=1536, decoder_ffn_dim=1536, max_source_positions=1500,
    max_target_positions=448, pad_token_id=50257, bos_token_id=50257, eos_token_id=50257,
    decoder_start_token_id=50258, activation_function="gelu", dropout=0.0, attention_dropout=0.0,
    activation_dropout=0.0, encoder_layerdrop=0.0, decoder_layerdrop=0.0, scale_embedding=False,
    layer_norm_eps=1e-5,
)
_DICOW_DEFAULTS = dict(
    ctc_loss_reduction="mean", final_dropout=0.0, ctc_zero_infinity=False, ctc_weight=0.0, blank_token_id=None,
    additional_layer=False, additional_self_attention_layer=False, pre_ctc_sub_sample=False, use_fddt=True,
    fddt_is_diagonal=True, fddt_bias_only=False, fddt_use_silence=True, fddt_use_target=True, fddt_use_overlap=True,
    fddt_use_non_target=True, remove_timestamps_from_ctc=False, apply_fddt_to_n_layers=-1, fddt_init="suppressive",
    non_target_fddt_value=0.0, use_enrollments=False, scb_layers=None, use_pre_pos_fddt=False,
)

PRESETS = {
    "whisper-tiny": dict(d_model=384, encoder_layers=4, encoder_attention_heads=6, decoder_layers=4,
                         decoder_attention_heads=6, encoder_ffn_dim=1536, decoder_ffn_dim=1536, num_mel_bins=80,
                         vocab_size=51865),
    "whisper-base": dict(d_model=512, encoder_layers=6, encoder_attention_heads=8, decoder_layers=6,
                         decoder_attention_heads=8, encoder_ffn_dim=2048, decoder_ffn_dim=2048, num_mel_bins=80,
                         vocab_size=51865),
    "whisper-large-v3-turbo": dict(d_model=1280, encoder_layers=32, encoder_attention_heads=20, decoder_layers=4,
                                   decoder_attention_heads=20, encoder_ffn_dim=5120, decoder_ffn_dim=5120,
                                   num_mel_bins=128, vocab_size=51866),
}


class DiCoWConfig(_ConfigBase):
    model_type = "DiCoW"
    # (HF-generic attribute names used by Trainer / generation utilities, as in WhisperConfig)
    attribute_map = {"num_attention_heads": "encoder_attention_heads", "hidden_size": "d_model", "num_hidden_layers": "encoder_layers"}
    keys_to_ignore_at_inference = ["past_key_values"]

    def __init__(self, **kwargs):
        vals = dict(_WHISPER_DEFAULTS)
        vals.update(_DICOW_DEFAULTS)
        known = set(vals)
        vals.update({k: v for k, v in kwargs.items() if k in known})
        extra = {k: v for k, v in kwargs.items() if k not in known}
        extra.pop("use_return_dict", None)                   # (pre-round-6 config.json files carry it; HF has `return_dict`)
        extra.pop("model_type", None)
        for k, v in vals.items():
            setattr(self, k, v)
        if _HF:
            base_args = set(inspect.signature(_ConfigBase.__init__).parameters) - {"self"}
            base_kw = {k: extra.pop(k) for k in list(extra) if k in base_args}
            base_kw.setdefault("is_encoder_decoder", True)
            super().__init__(**base_kw)
            self.tie_word_embeddings = True                  # proj_out.weight IS model.decoder.embed_tokens.weight
            self.use_cache = True
        else:
            self.use_return_dict = True
        for k, v in extra.items():                           # whatever else a Whisper config.json holds (suppress_tokens, max_length, ...)
            try:
                setattr(self, k, v)
            except AttributeError:
                pass
        if self.d_model % 64 != 0 or self.d_model // self.encoder_attention_heads != 64 \
                or self.d_model // self.decoder_attention_heads != 64:
            raise ValueError("the HIP attention kernels support head_dim == 64 only (every Whisper size)")
        if self.activation_function != "gelu":
            raise ValueError("only the exact-erf 'gelu' activation (Whisper default) is implemented")
        for k in ("dropout", "attention_dropout", "activation_dropout", "encoder_layerdrop", "decoder_layerdrop"):
            if getattr(self, k) != 0.0:
                raise ValueError(f"{k} != 0 is not supported by the fused training path (Whisper default is 0)")

    @classmethod
    def preset(cls, name: str, **overrides):
        kw = dict(PRESETS[name.replace("openai/", "")])
        kw.update(overrides)
        return cls(**kw)

    @classmethod
    def from_hf(cls, cfg, **overrides):
        d = cfg if isinstance(cfg, dict) else cfg.to_dict()
        known = set(_WHISPER_DEFAULTS) | set(_DICOW_DEFAULTS)
        kw = {k: v for k, v in d.items() if k in known}
        kw.update(overrides)
        return cls(**kw)

    def hot_path_dict(self):
        """The Whisper dimensions + the DiCoW switches (what this implementation reads), without HF's bookkeeping keys."""
        return {k: getattr(self, k) for k in list(_WHISPER_DEFAULTS) + list(_DICOW_DEFAULTS)}

    if not _HF:
        to_dict = hot_path_dict

    @property
    def num_fddts(self) -> int:
        return self.encoder_layers if self.apply_fddt_to_n_layers == -1 else self.apply_fddt_to_n_layers
