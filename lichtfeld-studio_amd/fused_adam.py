"""Mirror of gs::training::FusedAdam (/root/reference/src/training/optimizers/fused_adam.cpp:22-95,
fused_adam.hpp) and ExponentialLR (scheduler.hpp:12) over the HIP Adam kernel.

Per-group lr/eps/betas, lazily created state, per-parameter step_count, bias-correction
reciprocals computed in double on the host and cast to float (fused_adam.cpp:78-92), and the
"skip the higher-degree SH group while iteration <= 1000" shortcut (:68-70; step_count still
advances, :66).  `fused=True` issues ONE multi-tensor launch for all groups instead of the
reference's one launch per tensor; the per-element arithmetic is identical.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch

from . import ops


class FusedAdam:
    def __init__(self, param_groups: List[dict], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-15, fused: bool = True):
        self.param_groups = []
        for g in param_groups:
            g = dict(g)
            g.setdefault("lr", lr); g.setdefault("betas", betas); g.setdefault("eps", eps)
            g["params"] = list(g["params"])
            self.param_groups.append(g)
        self.state = {}
        self.fused = fused
        self._inline_done = set()

    def _state(self, p: torch.Tensor) -> dict:
        st = self.state.get(id(p))
        if st is None:
            st = {"step_count": 0, "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
            self.state[id(p)] = st
        return st

    def prepare_inline(self, p: torch.Tensor) -> dict:
        """The scalars and moment tensors of this step's update of `p`, for a producer kernel that applies Adam itself
        (lfs_sh_model_bwd_adam). Advances step_count; the next step() skips `p` (once)."""
        for group in self.param_groups:
            if any(q is p for q in group["params"]):
                st = self._state(p)
                st["step_count"] += 1
                beta1, beta2 = group["betas"]
                self._inline_done.add(id(p))
                return {"exp_avg": st["exp_avg"], "exp_avg_sq": st["exp_avg_sq"], "lr": float(group["lr"]), "beta1": float(beta1), "beta2": float(beta2),
                        "eps": float(group["eps"]), "bc1_rcp": float(1.0 / (1.0 - math.pow(beta1, st["step_count"]))),
                        "bc2_sqrt_rcp": float(1.0 / math.sqrt(1.0 - math.pow(beta2, st["step_count"])))}
        raise ValueError("prepare_inline: not a parameter of this optimizer")

    @torch.no_grad()
    def step(self, iteration: int) -> None:
        entries = []
        for i, group in enumerate(self.param_groups, start=1):
            lr, eps = group["lr"], group["eps"]
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if id(p) in self._inline_done:   # updated by its producer kernel in this step
                    self._inline_done.discard(id(p))
                    continue
                if p.grad is None:
                    continue
                st = self._state(p)
                st["step_count"] += 1
                if i == 3 and iteration <= 1000:  # shN is unused so far: free speed-up (fused_adam.cpp:68-70)
                    continue
                bc1_rcp = 1.0 / (1.0 - math.pow(beta1, st["step_count"]))
                bc2_sqrt_rcp = 1.0 / math.sqrt(1.0 - math.pow(beta2, st["step_count"]))
                entries.append((p, st["exp_avg"], st["exp_avg_sq"], p.grad.contiguous(), float(lr), float(beta1), float(beta2),
                                float(eps), float(bc1_rcp), float(bc2_sqrt_rcp)))
        self._inline_done.clear()   # (ids of tensors replaced meanwhile must not linger: Python reuses ids)
        if self.fused:
            ops.adam_step_multi(entries)
        else:
            for e in entries:
                ops.adam_step_wrapper(*e)

    def replace_param(self, group_index: int, old: torch.Tensor, new: torch.Tensor, state_fn) -> None:
        """Swap a parameter tensor for a new one (densification: cat / index_select), carrying its Adam state over:
        exp_avg / exp_avg_sq -> state_fn(tensor), step_count kept (strategy_utils.cpp:57-129)."""
        group = self.param_groups[group_index]
        group["params"] = [new if p is old else p for p in group["params"]]
        st = self.state.pop(id(old), None)
        if st is not None:
            self.state[id(new)] = {"step_count": st["step_count"], "exp_avg": state_fn(st["exp_avg"]).contiguous(),
                                   "exp_avg_sq": state_fn(st["exp_avg_sq"]).contiguous()}

    def zero_grad(self, set_to_none: bool = True) -> None:
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.detach_().zero_()


class ExponentialLR:
    """scheduler.hpp:12 — decays only param_group_index (the means group), gamma = 0.01^(1/iters)."""

    def __init__(self, optimizer: FusedAdam, gamma: float, param_group_index: int = 0):
        self.opt, self.gamma, self.idx = optimizer, gamma, param_group_index

    def step(self) -> None:
        self.opt.param_groups[self.idx]["lr"] *= self.gamma


class WarmupExponentialLR:
    """scheduler.hpp / scheduler.cpp:27-63 — linear warm-up from warmup_start_factor to 1 over warmup_steps, exponential decay after;
    param_group_index -1 = all groups. Works on FusedAdam and torch.optim optimizers (both expose param_groups[i]["lr"])."""

    def __init__(self, optimizer, gamma: float, warmup_steps: int = 0, warmup_start_factor: float = 1.0, param_group_index: int = -1):
        self.opt, self.gamma, self.warmup_steps, self.start, self.idx = optimizer, gamma, warmup_steps, warmup_start_factor, param_group_index
        self.current_step = 0
        self.initial_lrs = [g["lr"] for g in optimizer.param_groups]

    def step(self) -> None:
        self.current_step += 1
        groups = range(len(self.opt.param_groups)) if self.idx < 0 else [self.idx]
        for i in groups:
            if self.current_step <= self.warmup_steps:
                factor = self.start + (1.0 - self.start) * (self.current_step / self.warmup_steps)
            else:
                factor = self.gamma ** (self.current_step - self.warmup_steps)
            self.opt.param_groups[i]["lr"] = self.initial_lrs[i] * factor


def default_param_groups(model, scene_scale: float = 1.0, means_lr=1.6e-4, shs_lr=2.5e-3, scaling_lr=5e-3, rotation_lr=1e-3, opacity_lr=5e-2):
    """strategy_utils.cpp:20-45 with the learning rates of eval/default_optimization_params.json. The reference's OptimizationParameters fields and the scene
    scale are `float`: `means_lr * scene_scale` and `shs_lr / 20.f` are fp32 expressions widened to the optimizer's double afterwards, and so they are here (the
    scheduler then decays the double; the kernel receives static_cast<float>(lr) - one ulp of the start value can decide that cast)."""
    import numpy as np
    f32 = np.float32
    means, sh0, shN, scales, quats, opac = model.parameters()
    return [
        {"params": [means], "lr": float(f32(means_lr) * f32(scene_scale))},
        {"params": [sh0], "lr": float(f32(shs_lr))},
        {"params": [shN], "lr": float(f32(shs_lr) / f32(20.0))},
        {"params": [scales], "lr": float(f32(scaling_lr))},
        {"params": [quats], "lr": float(f32(rotation_lr))},
        {"params": [opac], "lr": float(f32(opacity_lr))},
    ]
