"""Training-time batch augmentation on the GPU: host planner + launches of csrc/augment.hip.

Mirrors the augmentation block of the reference collator (src/data/collators.py:189-214) for batches whose features
already live in HBM (features.log_mel):

  * ``soft_segment_augmentation``          reference DataCollator.soft_segment_augmentation   (collators.py:79-138)
  * ``add_gaussian_noise_and_rescale``     reference DataCollator.add_gaussian_noise_and_rescale (collators.py:50-77)
  * ``spec_aug_joint``                     reference collators.py:209-214 + SpecAug (augmentations.py:295-379)
  * ``BatchAugmenter``                     the gating / ordering of the three with the collator's field names

Split of work.  Every random number is drawn here, on the host, from the global torch CPU generator with the same
calls in the same order as the reference (randperm/randn, randint/rand per segment, the SpecAug randints), so that
``torch.manual_seed(s)`` gives the batch the reference's collator would have produced; none of those draws depends on
tensor data, which is what makes planning ahead of the device possible.  The plans are a few KB (the Gaussian noise is
the draw itself, <= B*4*T floats) and go up with one non-blocking copy each; the arithmetic on the batch runs in the
kernels.  No CPU fallback: tensors must be on the GPU.
"""
import math
from dataclasses import dataclass, field
from typing import Optional, Sequence, Tuple

import torch

from . import _lib as L

N_CLASSES = 4
MASKABLE_FEATURES = 128      # SpecAug.forward masks x[:, :, :128] only (augmentations.py:369-375)


def _need_gpu(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise L.DicowError(f"{name}: the batch must be on the GPU (no CPU fallback)")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise L.DicowError(f"{name}: expected a contiguous fp32 tensor, got {t.dtype} contiguous={t.is_contiguous()}")


class _Staging:
    """Pinned host buffers for the plan uploads, reused across steps (allocating pinned memory per call costs more than
    the whole augmentation).  A slot is recycled only after the copy that last read it has completed."""
    SLOTS = 4

    def __init__(self):
        self.buf = [None] * self.SLOTS
        self.done = [None] * self.SLOTS
        self.i = 0

    def host(self, shape, dtype) -> torch.Tensor:
        """A pinned host tensor of the given shape from the next slot (valid until SLOTS further requests)."""
        k, self.i = self.i, (self.i + 1) % self.SLOTS
        nbytes = math.prod(shape) * torch.empty(0, dtype=dtype).element_size()
        if self.done[k] is not None:
            self.done[k].synchronize()
        if self.buf[k] is None or self.buf[k].numel() < nbytes:
            self.buf[k] = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8).pin_memory()
        h = self.buf[k][:nbytes].view(dtype).view(shape)
        h._dicow_slot = k
        return h

    def send(self, h: torch.Tensor, device) -> torch.Tensor:
        out = h.to(device, non_blocking=True)
        self.done[h._dicow_slot] = torch.cuda.Event()
        self.done[h._dicow_slot].record()
        return out

    def upload(self, t: torch.Tensor, device) -> torch.Tensor:
        if t.is_cuda:
            return t
        if t.numel() == 0:
            return t.to(device)
        h = self.host(tuple(t.shape), t.dtype)
        h.copy_(t)
        return self.send(h, device)


_staging = _Staging()


def _up(t: torch.Tensor, device) -> torch.Tensor:
    return _staging.upload(t.contiguous(), device)


# ------------------------------------------------------------------------------------------------- Gaussian noise
def plan_gaussian_noise(B: int, C: int, T: int, variance: float, fraction: float, pinned: bool = False):
    """Draws of collators.py:54-65: which rows, and their N(0,1) noise.  Returns None when no row is selected.
    pinned: draw the noise straight into a pinned staging slot (the upload then needs no extra host copy)."""
    n = int(B * fraction)
    if n == 0:
        return None
    rows = torch.randperm(B)[:n].to(torch.int32)
    noise = torch.randn((n, C, T), out=_staging.host((n, C, T), torch.float32)) if pinned else torch.randn((n, C, T))
    return rows, noise, float(torch.tensor(variance ** 0.5, dtype=torch.float32))


def add_gaussian_noise_and_rescale(stno: torch.Tensor, variance: float = 0.05, fraction: float = 0.5) -> torch.Tensor:
    """stno fp32 [B, 4, T] on the GPU; modified in place and returned."""
    _need_gpu(stno, "add_gaussian_noise_and_rescale")
    B, C, T = stno.shape
    plan = plan_gaussian_noise(B, C, T, variance, fraction, pinned=True)
    if plan is None:
        return stno
    rows, noise, sd = plan
    rows_d, noise_d = _up(rows, stno.device), _staging.send(noise, stno.device)
    L.call("dicow_stno_noise_rescale", stno.data_ptr(), rows_d.data_ptr(), noise_d.data_ptr(), rows.numel(), C, T, sd,
           L.stream())
    return stno


# ------------------------------------------------------------------------------------------------- soft segments
def plan_soft_segments(B: int, C: int, T: int, change_prob: float, min_seg_len: int, max_seg_len: int):
    """Draws of collators.py:96-126 for every row: only the segments that change are kept.
    Returns segs int32 [n, 4] = (row, start, end, pick) and coef fp32 [n, 2] = (1 - softness, softness)."""
    segs, coef = [], []
    for b in range(B):
        pos = 0
        while pos < T:
            end = min(pos + int(torch.randint(min_seg_len, max_seg_len + 1, (1,)).item()), T)
            if torch.rand(1).item() < change_prob and C > 1:
                pick = int(torch.randint(0, C - 1, (1,)).item())
                soft = torch.rand(1).item()
                segs.append((b, pos, end, pick))
                coef.append((1 - soft, soft))
            pos = end
    return (torch.tensor(segs, dtype=torch.int32).reshape(-1, 4), torch.tensor(coef, dtype=torch.float32).reshape(-1, 2))


def soft_segment_augmentation(stno: torch.Tensor, change_prob: float = 0.2, min_seg_len: int = 5,
                              max_seg_len: int = 20) -> torch.Tensor:
    """stno fp32 [B, 4, T] on the GPU; modified in place and returned."""
    _need_gpu(stno, "soft_segment_augmentation")
    B, C, T = stno.shape
    segs, coef = plan_soft_segments(B, C, T, change_prob, min_seg_len, max_seg_len)
    if segs.shape[0]:
        segs_d, coef_d = _up(segs, stno.device), _up(coef, stno.device)
        L.call("dicow_stno_segment_augment", stno.data_ptr(), segs_d.data_ptr(), coef_d.data_ptr(), segs.shape[0], C, T,
               L.stream())
    return stno


# ------------------------------------------------------------------------------------------------- SpecAug
@dataclass
class SpecAugConfig:
    """The parameters the reference collator hard-wires (collators.py:31-48)."""
    apply_time_warp: bool = True
    time_warp_window: int = 5
    apply_freq_mask: bool = True
    freq_mask_width_range: Tuple[int, int] = (0, 27)
    num_freq_mask: int = 2
    apply_time_mask: bool = True
    time_mask_width_ratio_range: Tuple[float, float] = (0.0, 0.05)
    num_time_mask: int = 5


@dataclass
class SpecAugPlan:
    center: int = 0
    warped: int = -1                                       # < 0: no warp
    fmask: torch.Tensor = field(default_factory=lambda: torch.zeros(0, 0, 2, dtype=torch.int32))
    tmask: torch.Tensor = field(default_factory=lambda: torch.zeros(0, 0, 2, dtype=torch.int32))


def _plan_masks(B: int, D: int, width_range: Sequence[int], num_mask: int) -> torch.Tensor:
    """Draws of mask_along_axis (augmentations.py:50-60) -> int32 [B, num_mask, 2] = (pos, len)."""
    length = torch.randint(width_range[0], width_range[1], (B, num_mask))
    pos = torch.randint(0, max(1, D - int(length.max())), (B, num_mask))
    return torch.stack([pos, length], dim=-1).to(torch.int32)


def plan_spec_aug(B: int, T: int, n_features: int, cfg: SpecAugConfig = SpecAugConfig()) -> SpecAugPlan:
    """Draws of SpecAug.forward for an input [B, T, n_features] (augmentations.py:363-379), in its order."""
    plan = SpecAugPlan()
    w = cfg.time_warp_window
    if cfg.apply_time_warp and T - w > w:                                      # augmentations.py:101-106
        plan.center = int(torch.randint(w, T - w, (1,))[0])
        plan.warped = int(torch.randint(plan.center - w, plan.center + w, (1,))[0]) + 1
    if cfg.apply_freq_mask:
        plan.fmask = _plan_masks(B, min(MASKABLE_FEATURES, n_features), cfg.freq_mask_width_range, cfg.num_freq_mask)
    if cfg.apply_time_mask:                                                    # augmentations.py:267-281
        lo = max(0, math.floor(T * cfg.time_mask_width_ratio_range[0]))
        hi = min(T, math.floor(T * cfg.time_mask_width_ratio_range[1]))
        if hi > lo:
            plan.tmask = _plan_masks(B, T, (lo, hi), cfg.num_time_mask)
    return plan


def spec_aug_joint(mel: torch.Tensor, stno: torch.Tensor, sub: int = 2, cfg: SpecAugConfig = SpecAugConfig(),
                   plan: Optional[SpecAugPlan] = None):
    """mel fp32 [B, M, T], stno fp32 [B, 4, T/sub] on the GPU -> (mel_out, stno_out), new tensors."""
    _need_gpu(mel, "spec_aug_joint")
    _need_gpu(stno, "spec_aug_joint")
    B, M, T = mel.shape
    if stno.shape != (B, N_CLASSES, T // sub) or T % sub:
        raise L.DicowError(f"spec_aug_joint: stno {tuple(stno.shape)} does not match mel {tuple(mel.shape)} / {sub}")
    if plan is None:
        plan = plan_spec_aug(B, T, M + N_CLASSES, cfg)
    fm, tm = _up(plan.fmask, mel.device), _up(plan.tmask, mel.device)
    n_f = plan.fmask.shape[1] if plan.fmask.numel() else 0
    n_t = plan.tmask.shape[1] if plan.tmask.numel() else 0
    mel_out, stno_out = torch.empty_like(mel), torch.empty_like(stno)
    L.call("dicow_specaug_joint", mel.data_ptr(), stno.data_ptr(), mel_out.data_ptr(), stno_out.data_ptr(), B, M, T, sub,
           plan.center, plan.warped, fm.data_ptr() if n_f else None, n_f, tm.data_ptr() if n_t else None, n_t,
           min(MASKABLE_FEATURES, M + N_CLASSES), L.stream())
    return mel_out, stno_out


# ------------------------------------------------------------------------------------------------- collator block
@dataclass
class BatchAugmenter:
    """Field names and defaults of the reference DataCollator (collators.py:19-27); ``__call__`` is its augmentation
    block (collators.py:189-214) for a batch dict holding GPU tensors."""
    conv_subsample_factor: int = 2
    stno_gaussian_noise_var: Optional[float] = None
    stno_gaussian_noise_prob: Optional[float] = None
    stno_segment_augment_prob: Optional[float] = 0.3
    stno_segment_change_prob: float = 0.1
    stno_min_segment_length: int = 5
    stno_max_segment_length: int = 50
    spec_aug_prob: float = 0.3
    spec_aug: SpecAugConfig = field(default_factory=SpecAugConfig)

    def __call__(self, batch: dict) -> dict:
        stno = batch["stno_mask"]
        if (self.stno_segment_augment_prob is not None and self.stno_segment_augment_prob > 0
                and torch.rand(1).item() < self.stno_segment_augment_prob):
            stno = soft_segment_augmentation(stno.clone(), self.stno_segment_change_prob, self.stno_min_segment_length,
                                             self.stno_max_segment_length)
        if self.stno_gaussian_noise_var is not None and self.stno_gaussian_noise_var > 0:
            stno = add_gaussian_noise_and_rescale(stno.clone() if stno is batch["stno_mask"] else stno,
                                                  self.stno_gaussian_noise_var, self.stno_gaussian_noise_prob)
        batch["stno_mask"] = stno
        if torch.rand(1).item() < self.spec_aug_prob:
            batch["input_features"], batch["stno_mask"] = spec_aug_joint(batch["input_features"].contiguous(), stno,
                                                                         self.conv_subsample_factor, self.spec_aug)
        return batch
