"""MI355X-native (gfx950) DiCoW / SE-DiCoW training-step hot path.

Host side (Python, mirrors the reference's module surface: ``FDDT``, ``DiCoWEncoder``,
``DiCoWForConditionalGeneration``, ``DiCoWConfig``) over the C-ABI library ``libdicow_hip.so``
(hand-written HIP kernels, see ``csrc/`` and ``include/dicow_hip.h``).

There is NO CPU fallback: every op raises if the HIP library is missing or the tensors are not
on a GPU.  The CPU oracle lives in ``oracle/`` and is test infrastructure only.
"""
from . import _lib  # noqa: F401
from .config import DiCoWConfig, PRESETS  # noqa: F401
from .modeling import (FDDT, DiCoWEncoder, DiCoW, DiCoWForConditionalGeneration, SpeakerCommunicationBlock,  # noqa: F401
                       shift_tokens_right, build_ts_tables)

__all__ = ["DiCoWConfig", "FDDT", "DiCoWEncoder", "DiCoW", "DiCoWForConditionalGeneration", "SpeakerCommunicationBlock"]
