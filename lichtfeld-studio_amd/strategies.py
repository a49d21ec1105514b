"""Mirror of the reference's strategy layer, the caller of the ★ MCMC kernels:
`gs::training::MCMC` (src/training/strategies/mcmc.cpp:104-506) and `DefaultStrategy` (default_strategy.cpp:18-335) with the
optimizer-state surgery of strategy_utils.cpp:57-129, over this package's ops (`relocation`, `add_noise`,
`quats_to_rotmats`) and `FusedAdam`.

Differences from the reference, all about data parallelism (the reference is single-GPU):
  * every random draw (`multinomial`, `randn`) takes an explicit `torch.Generator`; ranks that seed it identically and
    hold bit-identical parameters (they do: same all-reduced gradients, same Adam) take identical densification decisions,
    so no index broadcast is needed;
  * parameters are replaced through `SplatModel` + `FusedAdam.replace_param`, and the trainer is told (`on_resize`) to
    rebuild its flat gradient bucket.
`DefaultStrategy` needs `densification_info` [2,N] (visibility count, accumulated screen-space gradient norm): in the reference
it is produced by the fastgs rasterizer's backward (kernels_backward.cuh:233-236, SURVEY.md §8f row 1, not built yet); here it
is an input of `grow_gs` so the state surgery is testable on its own.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional

import torch

from . import ops
from .fused_adam import ExponentialLR, FusedAdam, default_param_groups
from .rasterizer import SplatModel


@dataclass
class OptimizationParameters:
    """The fields of gs::param::OptimizationParameters the strategies read (defaults: eval/mcmc_optimization_params.json /
    eval/default_optimization_params.json)."""
    iterations: int = 30000
    sh_degree_interval: int = 1000
    means_lr: float = 1.6e-4
    shs_lr: float = 2.5e-3
    opacity_lr: float = 5e-2
    scaling_lr: float = 5e-3
    rotation_lr: float = 1e-3
    lambda_dssim: float = 0.2
    min_opacity: float = 0.005
    refine_every: int = 100
    start_refine: int = 500
    stop_refine: int = 25000
    grad_threshold: float = 2e-4
    opacity_reg: float = 0.01
    scale_reg: float = 0.01
    max_cap: int = 1_000_000
    prune_opacity: float = 0.005
    grow_scale3d: float = 0.01
    prune_scale3d: float = 0.1
    reset_every: int = 3000
    pause_refine_after_reset: int = 0
    revised_opacity: bool = False

    @staticmethod
    def for_strategy(strategy: str, **kw) -> "OptimizationParameters":
        """The reference keeps one JSON per strategy; the class defaults are eval/mcmc_optimization_params.json, and ADC differs in
        stop_refine (15000) and the regularisers (0) - eval/default_optimization_params.json."""
        if strategy == "default":
            kw = dict(dict(stop_refine=15000, opacity_reg=0.0, scale_reg=0.0), **kw)
        return OptimizationParameters(**kw)


_PARAM_NAMES = ["means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opacities"]  # param-group order, strategy_utils.cpp:35-40


def _binoms(n_max: int, device) -> torch.Tensor:
    """mcmc.cpp:449-463 (float32 running product, same arithmetic)."""
    b = torch.zeros((n_max, n_max), dtype=torch.float32)
    for n in range(n_max):
        for k in range(n + 1):
            v = torch.tensor(1.0, dtype=torch.float32)
            for i in range(k):
                v = v * (torch.tensor(float(n - i), dtype=torch.float32) / torch.tensor(float(i + 1), dtype=torch.float32))
            b[n, k] = v
    return b.to(device)


def _split_samples(rotmats: torch.Tensor, scales: torch.Tensor, rnd: torch.Tensor) -> torch.Tensor:
    """R diag(s) z for rotmats [n,3,3], scales [n,3], rnd [b,n,3] -> [b,n,3] (default_strategy.cpp:100-104), written out term by term so that a row's
    result does not depend on how many rows are processed with it (an einsum may pick a different contraction kernel)."""
    t = scales.unsqueeze(0) * rnd
    return (rotmats[:, :, 0].unsqueeze(0) * t[..., 0:1] + rotmats[:, :, 1].unsqueeze(0) * t[..., 1:2]) + rotmats[:, :, 2].unsqueeze(0) * t[..., 2:3]


class _StrategyBase:
    def __init__(self, model: SplatModel, params: OptimizationParameters, scene_scale: float = 1.0,
                 generator: Optional[torch.Generator] = None, on_resize: Optional[Callable[[], None]] = None):
        self.model, self.params, self.scene_scale = model, params, scene_scale
        self.device = model.means.device
        self.generator = generator if generator is not None else torch.Generator(device=self.device).manual_seed(0)
        self.on_resize = on_resize
        self.max_sh_degree = int(round((1 + model.shN.shape[1]) ** 0.5)) - 1
        self.optimizer = FusedAdam(default_param_groups(model, scene_scale, params.means_lr, params.shs_lr, params.scaling_lr,
                                                        params.rotation_lr, params.opacity_lr))
        self.scheduler = ExponentialLR(self.optimizer, gamma=0.01 ** (1.0 / params.iterations), param_group_index=0)

    # -- strategy_utils.cpp:57-129 ------------------------------------------------------------------------------------
    def _update_params(self, param_fn, state_fn, idxs=range(6)) -> None:
        """param_fn(i, param) -> new leaf tensor; state_fn(state_tensor, new_param) -> new state tensor (exp_avg / exp_avg_sq)."""
        for i in idxs:
            old = getattr(self.model, _PARAM_NAMES[i])
            new = param_fn(i, old.detach()).contiguous().requires_grad_(old.requires_grad)
            self.optimizer.replace_param(i, old, new, lambda t: state_fn(t, new))
            setattr(self.model, _PARAM_NAMES[i], new)
        if self.on_resize is not None:
            self.on_resize()

    def increment_sh_degree(self) -> None:  # splat_data.cpp:387-391
        if self.model.active_sh_degree < self.max_sh_degree:
            self.model.active_sh_degree += 1

    def _remove(self, is_prune: torch.Tensor) -> None:  # default_strategy.cpp:197-227 / mcmc.cpp:395-433
        keep = is_prune.logical_not().nonzero().squeeze(-1)
        self._update_params(lambda i, p: p.index_select(0, keep), lambda s, new: s.index_select(0, keep))

    @torch.no_grad()
    def remove_gaussians(self, mask: torch.Tensor) -> None:
        if int(mask.sum()) == 0:
            return
        self._remove(mask)

    def step(self, it: int) -> None:  # mcmc.cpp:386-393 / default_strategy.cpp:327-334
        if it < self.params.iterations:
            self.optimizer.step(it)
            self.optimizer.zero_grad(set_to_none=True)
            self.scheduler.step()


class MCMC(_StrategyBase):
    """mcmc.cpp. `noise_lr` = 5e5 (mcmc.hpp)."""

    def __init__(self, *a, noise_lr: float = 5e5, **kw):
        super().__init__(*a, **kw)
        self.noise_lr = noise_lr
        self.n_max = 51
        self.binoms = _binoms(self.n_max, self.device)

    def multinomial_sample(self, weights: torch.Tensor, n: int, replacement: bool = True) -> torch.Tensor:
        """mcmc.cpp:43-85. torch.multinomial is limited to 2^24 categories; above that: inverse-CDF sampling (on the device,
        with this strategy's generator, instead of the reference's host loop over std::mt19937)."""
        if weights.shape[0] <= (1 << 24):
            return torch.multinomial(weights, n, replacement, generator=self.generator)
        cdf = torch.cumsum(weights.double() / weights.double().sum(), 0)
        u = torch.rand(n, device=weights.device, dtype=torch.float64, generator=self.generator)
        return torch.searchsorted(cdf, u).clamp_max_(weights.shape[0] - 1)

    def is_refining(self, it: int) -> bool:  # mcmc.cpp:502-506
        p = self.params
        return it < p.stop_refine and it > p.start_refine and it % p.refine_every == 0

    def _zero_state(self, idxs: torch.Tensor) -> None:  # mcmc.cpp:87-111
        for i in range(6):
            st = self.optimizer.state.get(id(getattr(self.model, _PARAM_NAMES[i])))
            if st is not None:
                st["exp_avg"].index_fill_(0, idxs, 0)
                st["exp_avg_sq"].index_fill_(0, idxs, 0)

    def _relocation(self, sampled_idxs: torch.Tensor, ratios: torch.Tensor):
        m = self.model
        opac = m.get_opacity().detach()
        new_o, new_s = ops.relocation(opac.index_select(0, sampled_idxs).contiguous(), m.get_scaling().detach().index_select(0, sampled_idxs).contiguous(),
                                      ratios.contiguous(), self.binoms, self.n_max)
        new_o = new_o.clamp_(self.params.min_opacity, 1.0 - 1e-7)
        m.raw_opacities.data.index_put_((sampled_idxs,), torch.logit(new_o))
        m.raw_scales.data.index_put_((sampled_idxs,), torch.log(new_s))

    @torch.no_grad()
    def relocate_gs(self) -> torch.Tensor:  # mcmc.cpp:113-194
        """Dead Gaussians (opacity <= min_opacity or a degenerate quaternion) become copies of alive ones drawn in proportion to opacity; the
        drawn sources get the relocated opacity / scale (gsplat::relocation) and zeroed Adam moments. ONE enqueue, no host round trip
        (lfs_mcmc_relocate: the reference's nonzero() / multinomial() pipeline as fixed-size device passes). The random
        numbers come from this strategy's generator, so identically seeded replicas relocate identically. -> n_dead as a DEVICE tensor."""
        import ctypes as C

        from .capi import ParamRows, check, load_library, ptr, stream, workspace
        m, lib = self.model, load_library()
        N = int(m.means.shape[0])
        n_dead = torch.zeros(1, dtype=torch.int32, device=self.device)
        if N == 0:
            return n_dead
        u = torch.rand(N, dtype=torch.float64, device=self.device, generator=self.generator)
        rows = (ParamRows * 6)()
        for i, name in enumerate(_PARAM_NAMES):
            p = getattr(m, name)
            st = self.optimizer.state.get(id(p)) or {}
            rows[i].param = p.data.data_ptr()
            rows[i].exp_avg = st["exp_avg"].data_ptr() if "exp_avg" in st else None
            rows[i].exp_avg_sq = st["exp_avg_sq"].data_ptr() if "exp_avg_sq" in st else None
            rows[i].width = p[0].numel() if p.dim() > 1 else 1
        ws = workspace(lib.lfs_mcmc_relocate_workspace_bytes(C.c_uint32(N)), self.device, "mcmc_relocate")
        check(lib.lfs_mcmc_relocate(C.c_uint32(N), rows, ptr(u), ptr(self.binoms), C.c_int32(self.n_max), C.c_float(self.params.min_opacity), ptr(n_dead),
                                    ptr(ws), C.c_size_t(ws.numel()), stream()), "mcmc_relocate")
        return n_dead

    @torch.no_grad()
    def add_new_gs(self) -> int:  # mcmc.cpp:196-340
        m = self.model
        cur = m.means.shape[0]
        n_new = max(0, min(self.params.max_cap, int(1.05 * cur)) - cur)
        if n_new == 0:
            return 0
        opac = m.get_opacity().detach()
        sampled = self.multinomial_sample(opac.flatten(), n_new, True)
        ratios = torch.zeros(cur, dtype=torch.float32, device=self.device)
        ratios.index_add_(0, sampled, torch.ones_like(sampled, dtype=torch.float32))
        ratios = (ratios.index_select(0, sampled) + 1).clamp_(1, self.n_max).to(torch.int32)
        self._relocation(sampled, ratios)  # existing Gaussians first, then the copies inherit the updated values
        self._update_params(lambda i, p: torch.cat([p, p.index_select(0, sampled)], 0),
                            lambda s, new: torch.cat([s, torch.zeros((n_new,) + tuple(s.shape[1:]), dtype=s.dtype, device=s.device)], 0))
        return n_new

    @torch.no_grad()
    def inject_noise(self) -> None:  # mcmc.cpp:342-360
        m = self.model
        lr = float(self.optimizer.param_groups[0]["lr"]) * self.noise_lr
        noise = torch.randn(m.means.shape, device=self.device, dtype=m.means.dtype, generator=self.generator)
        ops.add_noise(m.raw_opacities.detach(), m.raw_scales.detach(), m.raw_quats.detach(), noise, m.means.data, lr)

    @torch.no_grad()
    def post_backward(self, it: int) -> None:  # mcmc.cpp:362-384
        if it % self.params.sh_degree_interval == 0:
            self.increment_sh_degree()
        if self.is_refining(it):
            self.relocate_gs()
            self.add_new_gs()
        self.inject_noise()


class DefaultStrategy(_StrategyBase):
    """default_strategy.cpp (ADC: duplicate / split / prune / opacity reset)."""
    fused_refine = True   # post_backward: grow_and_prune_fused (one host read per refinement) instead of grow_gs + prune_gs (the reference's sequence)

    def is_refining(self, it: int) -> bool:  # :30-34
        p = self.params
        return it > p.start_refine and it % p.refine_every == 0 and it % p.reset_every >= p.pause_refine_after_reset

    @torch.no_grad()
    def duplicate(self, is_duplicated: torch.Tensor) -> None:  # :47-82
        idx = is_duplicated.nonzero().squeeze(-1)
        n = int(idx.numel())
        self._update_params(lambda i, p: torch.cat([p, p.index_select(0, idx)], 0),
                            lambda s, new: torch.cat([s, torch.zeros((n,) + tuple(s.shape[1:]), dtype=s.dtype, device=s.device)], 0))

    @torch.no_grad()
    def split(self, is_split: torch.Tensor, rnd_full: Optional[torch.Tensor] = None) -> None:  # :84-160
        """rnd_full [2, N0, 3] (tests): the normal deviates per ORIGINAL row instead of a fresh draw for the split rows only."""
        m = self.model
        idx = is_split.nonzero().squeeze(-1)
        rest = is_split.logical_not().nonzero().squeeze(-1)
        scales = m.get_scaling().detach().index_select(0, idx)
        rotmats = ops.quats_to_rotmats(m.get_rotation().detach().index_select(0, idx).contiguous())
        n, split_size = int(idx.numel()), 2
        if rnd_full is not None:
            rnd = rnd_full.index_select(1, idx)
        else:
            rnd = torch.randn((split_size, n, 3), device=self.device, dtype=scales.dtype, generator=self.generator)
        samples = _split_samples(rotmats, scales, rnd)

        def param_fn(i, p):
            sp = p.index_select(0, idx)
            if i == 0:
                new = (sp.unsqueeze(0) + samples).reshape(-1, 3)
            elif i == 3:
                new = torch.log(scales / 1.6).repeat(split_size, 1)
            elif i == 5 and self.params.revised_opacity:
                new = torch.logit(1.0 - torch.sqrt(1.0 - torch.sigmoid(sp))).repeat(split_size)
            else:
                new = sp.repeat((split_size,) + (1,) * (p.dim() - 1))
            return torch.cat([p.index_select(0, rest), new], 0)

        self._update_params(param_fn, lambda s, new: torch.cat([s.index_select(0, rest), torch.zeros((n * split_size,) + tuple(s.shape[1:]),
                                                                                                     dtype=s.dtype, device=s.device)], 0))

    @torch.no_grad()
    def grow_gs(self, it: int, densification_info: torch.Tensor, rnd_full: Optional[torch.Tensor] = None) -> None:  # :162-195
        m, p = self.model, self.params
        grads = densification_info[1] / densification_info[0].clamp_min(1.0)
        is_grad_high = grads > p.grad_threshold
        is_small = m.get_scaling().detach().max(-1).values <= p.grow_scale3d * self.scene_scale
        is_dup = is_grad_high & is_small
        n_dup = int(is_dup.sum())
        is_split = is_grad_high & ~is_small
        n_split = int(is_split.sum())
        if n_dup > 0:
            self.duplicate(is_dup)
        is_split = torch.cat([is_split, torch.zeros(n_dup, dtype=torch.bool, device=self.device)])
        if n_split > 0:
            self.split(is_split, rnd_full)

    @torch.no_grad()
    def grow_and_prune_fused(self, it: int, densification_info: torch.Tensor, rnd_full: Optional[torch.Tensor] = None) -> None:
        """grow_gs + prune_gs (default_strategy.cpp:162-249) as device-side index arithmetic with ONE host read - the new Gaussian count, which the
        host needs to allocate the tensors (the step-by-step form above follows the reference line by line with ~7 reads and
        rebuilds every parameter and both Adam moments three times). The outcome of duplicate -> split -> prune is known per ORIGINAL row up front:
        the final order is [rows that are not split | their duplicates | first children | second children], each filtered by the pruning test of the
        row it would hold (a duplicate has its source's parameters; both children of a row share opacity, rotation and scale, hence the test). One
        inclusive scan over the 4 x N keep-flags gives every output slot, searchsorted inverts it, and every tensor is rebuilt once with a gather.
        Same values, bit for bit, as grow_gs + prune_gs given the same deviates (tests/test_gpu_strategies.py)."""
        m, p = self.model, self.params
        N = m.means.shape[0]
        means, raw_scales, raw_quats, raw_opac = m.means.detach(), m.raw_scales.detach(), m.raw_quats.detach(), m.raw_opacities.detach()
        grads = densification_info[1] / densification_info[0].clamp_min(1.0)
        is_grad_high = grads > p.grad_threshold
        scales = torch.exp(raw_scales)
        is_small = scales.max(-1).values <= p.grow_scale3d * self.scene_scale
        is_dup, is_split = is_grad_high & is_small, is_grad_high & ~is_small
        child_raw_scales = torch.log(scales / 1.6)
        child_raw_opac = torch.logit(1.0 - torch.sqrt(1.0 - torch.sigmoid(raw_opac))) if p.revised_opacity else raw_opac

        def pruned(ro, rs):  # prune_gs on a row with these raw opacity / scales (and this row's rotation)
            out = (torch.sigmoid(ro) < p.prune_opacity) | ((raw_quats ** 2).sum(-1) < 1e-8)
            if it > p.reset_every:
                out = out | (torch.exp(rs).max(-1).values > p.prune_scale3d * self.scene_scale)
            return out
        if N == 0:
            return
        keep_parent, keep_child = ~pruned(raw_opac, raw_scales), ~pruned(child_raw_opac, child_raw_scales)
        flags = torch.stack([~is_split & keep_parent, is_dup & keep_parent, is_split & keep_child, is_split & keep_child]).reshape(-1)
        inc = torch.cumsum(flags.to(torch.int64), 0)
        split_inc = torch.cumsum(is_split.to(torch.int64), 0)
        total, n_split = (int(x) for x in torch.stack([inc[-1], split_inc[-1]]).tolist())   # the ONE host read of the refinement step: new count + split count
        if total == 0:
            self._remove(torch.ones(N, dtype=torch.bool, device=self.device))
            return
        # the children of the split rows only: the normal deviates are drawn for exactly those rows, [2, n_split, 3] in row order - the draw of the
        # reference's split() (default_strategy.cpp:100-104) and of grow_gs above, so the generator stream is the same on both paths for the same seed.
        # rnd_full (tests): deviates per ORIGINAL row. nonzero_static: the size is known from the read above, no second synchronisation.
        sidx = torch.nonzero_static(is_split, size=n_split).reshape(-1)
        if rnd_full is not None:
            rnd = rnd_full.index_select(1, sidx)
        else:
            rnd = torch.randn((2, n_split, 3), device=self.device, dtype=scales.dtype, generator=self.generator)
        rotmats = ops.quats_to_rotmats(torch.nn.functional.normalize(raw_quats.index_select(0, sidx), dim=-1).contiguous()) if n_split else means.new_zeros((0, 3, 3))
        child_means = means.index_select(0, sidx).unsqueeze(0) + _split_samples(rotmats, scales.index_select(0, sidx), rnd)      # [2, n_split, 3]
        k = torch.searchsorted(inc, torch.arange(total, device=self.device), right=True)   # output slot -> flat (category, row)
        cat, row = torch.div(k, N, rounding_mode="floor"), k % N
        is_child, is_old = cat >= 2, cat == 0
        child_slot = ((cat - 2).clamp_min(0) * n_split + (split_inc.index_select(0, row) - 1).clamp_min(0)).clamp_max(max(2 * n_split - 1, 0))

        def param_fn(i, t):
            new = t.index_select(0, row)
            if i == 0 and n_split:
                new = torch.where(is_child.unsqueeze(-1), child_means.reshape(2 * n_split, 3).index_select(0, child_slot), new)
            elif i == 3:
                new = torch.where(is_child.unsqueeze(-1), child_raw_scales.index_select(0, row), new)
            elif i == 5:
                new = torch.where(is_child, child_raw_opac.index_select(0, row), new)
            return new

        def state_fn(st, new):
            g = st.index_select(0, row)
            return torch.where(is_old.reshape((-1,) + (1,) * (g.dim() - 1)), g, torch.zeros_like(g))
        self._update_params(param_fn, state_fn)

    @torch.no_grad()
    def prune_gs(self, it: int) -> None:  # :229-249
        m, p = self.model, self.params
        is_prune = m.get_opacity().detach() < p.prune_opacity
        is_prune |= (m.raw_quats.detach() ** 2).sum(-1) < 1e-8
        if it > p.reset_every:
            is_prune |= m.get_scaling().detach().max(-1).values > p.prune_scale3d * self.scene_scale
        if int(is_prune.sum()) > 0:
            self._remove(is_prune)

    @torch.no_grad()
    def reset_opacity(self) -> None:  # :251-293
        thr = torch.logit(torch.tensor(2.0 * self.params.prune_opacity)).item()
        self._update_params(lambda i, p: p.clamp_max(thr), lambda s, new: torch.zeros_like(s), idxs=[5])

    @torch.no_grad()
    def post_backward(self, it: int, densification_info: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:  # :295-325
        """Returns the (re-)initialised densification_info after a refinement step, else the tensor it was given."""
        p = self.params
        if it % p.sh_degree_interval == 0:
            self.increment_sh_degree()
        if it >= p.stop_refine:
            return None if it == p.stop_refine else densification_info
        if self.is_refining(it) and densification_info is not None:
            if self.fused_refine:
                self.grow_and_prune_fused(it, densification_info)
            else:
                self.grow_gs(it, densification_info)
                self.prune_gs(it)
            densification_info = torch.zeros((2, self.model.means.shape[0]), dtype=self.model.means.dtype, device=self.device)
        if it % p.reset_every == 0 and it > 0:
            self.reset_opacity()
        return densification_info
