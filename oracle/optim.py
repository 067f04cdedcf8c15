"""Oracle (test infrastructure, torch CPU): the optimizer side of the reference's training harness.

Restates what happens to the parameters between two forward passes when the reference trains with its recipe:
  * keyword freezing                       reference src/models/containers.py:80-90 (``"decoder"`` in dicow_v3.yaml:6-7)
  * staged freezing / preheat phase        reference src/train.py:174-178 (freeze_except) and
                                           src/utils/trainers.py:122-137 (unfreeze once global_step >= n; the optimizer
                                           and scheduler objects are kept: HF's create_optimizer_and_scheduler does
                                           nothing when they already exist)
  * two AdamW parameter groups             reference src/models/containers.py:100-114
  * gradient clipping                      HF Trainer: torch.nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)
  * cosine schedule with warm-up           HF get_cosine_schedule_with_warmup (lr_scheduler_type cosine, dicow_v3.yaml:66-67)
    applied as a LambdaLR that steps after the optimizer

The arithmetic is torch's own ``torch.optim.AdamW`` -- the third-party code the reference calls -- so this file is the
procedure, not a re-derivation.  Only tests/ may import it.

Pinned: tests/test_trajectory.py drives oracle/dicow_oracle.py with this harness over the eight optimizer steps of fixture F20
(tests/golden/make_golden_trajectory.py: the REAL reference's freeze_except / get_optimizer, transformers' cosine schedule and
clip_grad_norm_ on the real model) and reproduces its per-step loss, gradient norm, learning rates, trainable-parameter counts
and the updates of the watched parameters.
"""
import math

import torch


def cosine_with_warmup_lambda(step, warmup, total, num_cycles=0.5):
    """transformers.optimization._get_cosine_schedule_with_warmup_lr_lambda."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    progress = float(step - warmup) / float(max(1, total - warmup))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


class ReferenceHarness:
    """params: dict name -> fp32 tensor (cloned).  ``step(grads)`` takes dict name -> gradient for the parameters that
    currently require grad and applies one optimizer step exactly like the reference's Trainer would."""

    def __init__(self, params, lr, fddt_lr_multiplier, weight_decay, max_grad_norm, warmup_steps, max_steps,
                 frozen_keywords, preheat_prefixes, use_fddt_only_n_steps):
        self.p = {n: torch.nn.Parameter(t.detach().clone().float()) for n, t in params.items()}
        self.frozen_keywords, self.preheat = tuple(frozen_keywords), tuple(preheat_prefixes)
        self.n_steps, self.max_norm = use_fddt_only_n_steps, max_grad_norm
        for n, p in self.p.items():                                    # containers.py:80-90
            p.requires_grad_(not any(k in n for k in self.frozen_keywords))
        if self.n_steps > 0:                                           # train.py:176-178 -> containers.py:92-97
            for n, p in self.p.items():
                p.requires_grad_(any(n.startswith(pp) for pp in self.preheat))
        base = [p for n, p in self.p.items() if not any(n.startswith(pp) for pp in self.preheat)]
        new = [p for n, p in self.p.items() if any(n.startswith(pp) for pp in self.preheat)]
        self.opt = torch.optim.AdamW([{"params": base}, {"params": new, "lr": fddt_lr_multiplier * lr, "weight_decay": 0.0}],
                                     lr=lr, weight_decay=weight_decay)
        lam = (lambda s: cosine_with_warmup_lambda(s, warmup_steps, max_steps)) if max_steps > 0 else \
              (lambda s: min(1.0, s / max(1, warmup_steps)) if warmup_steps else 1.0)
        self.sched = torch.optim.lr_scheduler.LambdaLR(self.opt, lam)
        self.global_step, self.warmup_phase = 0, self.n_steps > 0

    def trainable(self):
        return [n for n, p in self.p.items() if p.requires_grad]

    def begin_step(self):
        if self.warmup_phase and self.global_step >= self.n_steps:     # trainers.py:122-137
            for n, p in self.p.items():
                p.requires_grad_(not any(k in n for k in self.frozen_keywords))
            self.warmup_phase = False

    def step(self, grads):
        for n, p in self.p.items():
            p.grad = grads[n].detach().clone().float() if (p.requires_grad and n in grads) else None
        torch.nn.utils.clip_grad_norm_(list(self.p.values()), self.max_norm)
        self.opt.step()
        self.sched.step()
        self.global_step += 1
