"""roctx ranges around the phases of a training step (rocprofv3 counters + roctx ranges).

``with tracing.range("forward"): ...`` pushes / pops a range on the calling thread through ``libroctx64.so`` (ROCm's
marker library: ``roctxRangePushA`` / ``roctxRangePop``); ``rocprofv3 --marker-trace`` shows the ranges beside the
kernel trace.  The ranges are host-side annotations of where the launches of a phase are ENQUEUED -- they cost two
library calls (~0.1 us each, no device work) and are therefore always on; ``DICOW_ROCTX=0`` turns them into no-ops.
A missing marker library also yields no-ops: tracing is never a reason for a step to fail.

The reference has no counterpart (it relies on the HF Trainer's wall-clock logging); the phase names follow its
training loop: ``forward`` and ``backward`` (src/utils/trainers.py:116-139 -> ``model(**batch)``, ``loss.backward()``),
``exchange`` (DDP's gradient all-reduce, scripts/submit_slurm.sh:34) and ``optimizer`` (clip + AdamW,
src/models/containers.py:100-114).
"""
import contextlib
import ctypes
import os

PHASES = ("forward", "backward", "exchange", "optimizer")

_lib = None
_enabled = os.environ.get("DICOW_ROCTX", "1") != "0"
counts = {}                      # range name -> number of times it was opened (tests; cheap)


def _load():
    global _lib, _enabled
    if _lib is not None or not _enabled:
        return _lib
    for name in ("libroctx64.so", "/opt/rocm/lib/libroctx64.so", "librocprofiler-sdk-roctx.so"):
        try:
            lib = ctypes.CDLL(name)
            lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
            lib.roctxRangePushA.restype = ctypes.c_int
            lib.roctxRangePop.argtypes = []
            lib.roctxRangePop.restype = ctypes.c_int
            _lib = lib
            return _lib
        except (OSError, AttributeError):
            continue
    _enabled = False
    return None


def available():
    """True when ranges reach the marker library (False: they are no-ops)."""
    return _load() is not None


def push(name):
    counts[name] = counts.get(name, 0) + 1
    lib = _load()
    if lib is not None:
        lib.roctxRangePushA(name.encode())


def pop():
    lib = _load()
    if lib is not None:
        lib.roctxRangePop()


@contextlib.contextmanager
def range(name):                 # noqa: A001  (the roctx vocabulary)
    push(name)
    try:
        yield
    finally:
        pop()
