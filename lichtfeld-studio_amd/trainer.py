"""Minimal stand-in for Trainer::train_step (/root/reference/src/training/trainer.cpp:579-760) on
the --gut path: render -> loss -> backward -> (all-reduce) -> FusedAdam::step -> zero_grad ->
scheduler.step. The reference's app shell (dataset IO, strategies, viewer, checkpoints) is out of
scope; the loss here is the rasterizer-only MSE of SURVEY.md §8d.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import torch

from . import dist as lfs_dist
from .fused_adam import ExponentialLR, FusedAdam, default_param_groups
from .rasterizer import Camera, RenderMode, SplatModel, rasterize
from .scenes import Scene


@dataclass(frozen=True)
class StepPlan:
    """Which form of the training step runs, and with which fusions (GutTrainer._train_step dispatches on it)."""
    path: str            # "fastgs" | "autograd" | "cxx_all" | "cxx_views" | "batch_views" | "py_views"
    inline_shN: bool     # shN's Adam update happens inside the SH backward (no shN gradient tensor)
    inline_all: bool     # every parameter is updated by the backward kernels (no gradient tensors at all)
    inline_shard: bool   # SH-sharded: the owners' multi-view SH backward applies the shard's Adam update
    multi: bool          # the multi-rank code path (world > 1, or LFS_DIST_FORCE_COLLECTIVES on one GPU)
    skip_deferred: bool  # the shN segment stays out of the flat all-reduce


def plan_step(*, rasterizer: str, fused_l2: bool, world: int, force_collectives: bool, sh_sharded: bool, shard_rows: int, n_views: int, loss: str,
              strategy: Optional[str], refining: bool, iteration: int, has_shN: bool, optimizer_fused: bool, bilateral: bool, inline_shN_adam: bool = True,
              inline_all_adam: bool = True, cxx_step: bool = True, batch_views: bool = True, factored_sh: bool = False) -> StepPlan:
    """Pure function of the configuration -> the step form. The rules, in the order they are applied:
      * fastgs rasterizer -> its own step; fused_l2 off -> torch autograd over the op-by-op mirror.
      * shN's Adam update moves into the SH backward (inline_shN) when Adam reads shN anyway (iteration > 1000, fused_adam.cpp:68-70), there IS an shN, the
        optimizer is the fused one, and the strategy does not touch shN before the optimizer step: no strategy, or MCMC between refinements (post_backward then
        only adds noise to the means, mcmc.cpp:362-384) - never on refining iterations (relocation rewrites shN rows and moments first).
        - replicated layout: additionally one view on one rank (gradient of a single view, no all-reduce: nothing else needs the tensor)
        - several views on one rank (batch_views): inside the ONE multi-view SH backward
        - SH-sharded, one view per rank: the owners' multi-view backward updates the shard (inline_shard)
      * inline_all (no gradient tensor at all) = inline_shN on one view / one rank + MSE + no strategy + no bilateral grid.
      * C++ step driver (cxx_step): inline_all -> ONE call (cxx_all); otherwise its per-view form whenever the layout is replicated and the rank has one view
        or runs the multi-rank path (cxx_views). Several views on one rank without collectives -> batch_views. Everything else - SH-sharded ranks, cxx_step
        off - -> the kernels enqueued from Python view by view (py_views)."""
    multi = world > 1 or force_collectives
    if rasterizer == "fastgs":
        return StepPlan("fastgs", False, False, False, multi, iteration <= 1000)
    if not fused_l2:
        return StepPlan("autograd", False, False, False, multi, iteration <= 1000)
    strat_ok = strategy is None or (strategy == "mcmc" and not refining)
    adam_reads_shN = iteration > 1000 and has_shN and optimizer_fused
    inline_one = inline_shN_adam and not multi and not sh_sharded and n_views == 1 and strat_ok and adam_reads_shN
    inline_all = inline_one and inline_all_adam and strategy is None and loss == "mse" and not bilateral
    inline_shard = inline_shN_adam and sh_sharded and n_views == 1 and adam_reads_shN and shard_rows > 0 and not refining
    skip_deferred = iteration <= 1000 or sh_sharded
    if factored_sh and multi and not sh_sharded:
        # replicated layout with the factored SH exchange (dist.ColorGradExchange): per view rasterizer backward + finish, rows gathered, ONE multi-view SH backward
        # over every rank's views with shN's Adam update inside whenever Adam reads shN and no refinement rewrites it first; sh0 / shN never enter the all-reduce
        if not cxx_step:
            # (a shape the speculative C++ step does not take - GutTrainer.__init__ rejects factored_sh for those up front; densification growing N past the index-bit
            #  limit mid-run lands here: an error on every rank at the same iteration, N is replicated)
            raise ValueError("the factored SH exchange runs through the C++ step driver (cxx_step): unsupported problem shape or cxx_step switched off")
        return StepPlan("cxx_factored", inline_shN_adam and strat_ok and adam_reads_shN, False, False, multi, True)
    if inline_all and cxx_step:
        return StepPlan("cxx_all", True, True, False, multi, skip_deferred)
    if cxx_step and not sh_sharded and (multi or n_views == 1):
        return StepPlan("cxx_views", inline_one, False, False, multi, skip_deferred)
    if batch_views and not multi and not sh_sharded and n_views > 1:
        return StepPlan("batch_views", inline_shN_adam and strat_ok and adam_reads_shN, False, False, multi, skip_deferred)
    return StepPlan("py_views", inline_one, inline_all, inline_shard, multi, skip_deferred)


class GutTrainer:
    def __init__(self, scene: Scene, device, iterations: int = 7000, world: int = 1, rank: int = 0,
                 views_per_rank: int = 1, fused_adam: bool = True, fused_l2: bool = True, loss: str = "mse", lambda_dssim: float = 0.2,
                 strategy: Optional[str] = None, opt_params=None, scene_scale: float = 1.0, seed: int = 0, rasterizer: str = "gut",
                 use_bilateral_grid: bool = False, bilateral_grid_dims=(16, 16, 8), bilateral_grid_lr: float = 2e-3, tv_loss_weight: float = 10.0,
                 sh_sharded: Optional[bool] = None, factored_sh: bool = False):
        """strategy: None (fixed set of Gaussians: the benchmark), "mcmc" (strategies.MCMC: relocation + growth + SGLD noise, with
        the scale / opacity regularisers of trainer.cpp:132-158) or "default" (ADC; needs densification_info, see strategies.py).
        `seed` seeds the strategy's generator: the same on every rank, so replicas densify identically."""
        self.device, self.world, self.rank, self.views_per_rank = device, world, rank, views_per_rank
        sc = scene.to(device)
        self.scene = sc
        mk = lambda t: t.clone().contiguous().requires_grad_(True)
        # Data-parallel layout. Default (round 3): the north-star one - Gaussians REPLICATED, per-rank forward / backward, one all-reduce of the flat gradient
        # bucket before the fused Adam step. sh_sharded=True opts into dist.ShExchange (shN and its Adam state owned by one rank each: 14 instead of 59 floats per
        # Gaussian in the all-reduce; fused 3DGUT step, no strategy or MCMC). bench.py --gpus N times both.
        # factored_sh=True (round 4) keeps the replicated layout and shrinks its collective: the SH gradients are exchanged as dL/dcolour rows (all-gather, 12 B per
        # Gaussian and view) and assembled by every rank itself (dist.ColorGradExchange); only 11 of 59 floats per Gaussian remain in the all-reduce.
        if sh_sharded is None:
            sh_sharded = False
        if factored_sh and (sh_sharded or not (fused_l2 and rasterizer == "gut")):
            raise ValueError("factored_sh is a variant of the replicated layout of the fused 3DGUT step")
        self.factored_sh = bool(factored_sh)
        if self.factored_sh:   # fail at construction, on every rank alike, not in the middle of a run (plan_step has no fallback form for this layout)
            from .capi import load_library
            if not load_library().lfs_gut_step_supported(int(sc.means.shape[0]), int(sc.width), int(sc.height), 16):
                raise ValueError("factored_sh needs a problem shape the C++ step driver takes (lfs_gut_step_supported: <= 512 tile rows, Gaussian index + tile column in 32 bits)")
        self.color_exchange = None
        if sh_sharded and not (fused_l2 and rasterizer == "gut" and strategy in (None, "mcmc")):
            raise ValueError("sh_sharded needs the fused 3DGUT step (no strategy, or MCMC: its refinement steps run on the gathered tensors)")
        self.sh_exchange = lfs_dist.ShExchange(sc.means.shape[0], world, rank) if sh_sharded else None
        shN0 = self.sh_exchange.shard(sc.shN) if sh_sharded else sc.shN
        # scenes built from a point cloud (loader.colmap_scene) carry active_sh_degree = 0: the SH schedule of the reference starts there
        self.model = SplatModel(mk(sc.means), mk(sc.sh0), mk(shN0), mk(sc.raw_scales), mk(sc.raw_quats), mk(sc.raw_opacities), sc.sh_degree,
                                active_sh_degree=sc.extra.get("active_sh_degree"))
        self.rasterizer = rasterizer  # "gut" (3DGUT, the north-star path) | "fastgs" (the reference's default EWA rasterizer)
        self._fg_settings = {}
        self.strategy = None
        self.strategy_kind = strategy
        self._resize_suspended, self._resize_pending = False, False
        self.batch_views = True          # one rank, several views per step: SH forward / backward once over all views (False: view by view, A/B and tests)
        self.densification_info = None   # [2,N]: fastgs backward's (visibility count, screen-space gradient norm) for ADC
        self.scale_reg = self.opacity_reg = 0.0
        if strategy == "default" and rasterizer != "fastgs":
            # ADC reads densification_info (per-Gaussian visibility counts and screen-space gradient norms), which only the fastgs backward
            # produces (kernels_backward.cuh:233-236); on the 3DGUT path grow / prune would silently never run. The reference pairs --gut with MCMC.
            raise ValueError("strategy='default' (ADC) needs rasterizer='fastgs': the 3DGUT backward has no densification_info")
        if strategy is not None:
            from . import strategies
            op = opt_params or strategies.OptimizationParameters.for_strategy(strategy, iterations=iterations)
            gen = torch.Generator(device=device).manual_seed(seed)
            cls = {"mcmc": strategies.MCMC, "default": strategies.DefaultStrategy}[strategy]
            self.strategy = cls(self.model, op, scene_scale=scene_scale, generator=gen, on_resize=self._on_resize)
            self.optimizer, self.scheduler = self.strategy.optimizer, self.strategy.scheduler
            if strategy == "mcmc":
                self.scale_reg, self.opacity_reg = op.scale_reg, op.opacity_reg
            lambda_dssim = op.lambda_dssim
        else:
            self.optimizer = FusedAdam(default_param_groups(self.model), fused=fused_adam)
            self.scheduler = ExponentialLR(self.optimizer, gamma=0.01 ** (1.0 / iterations), param_group_index=0)
        self.sh_degree_interval = 1000     # without a strategy the trainer keeps the SH schedule itself (strategies do it in post_backward)
        self.bg = torch.zeros(3, device=device)
        # appearance model of BASELINE config 5 (trainer.cpp:66-99: Adam(lr, eps 1e-15) + warm-up exponential schedule), fastgs path only
        self.bilateral, self.tv_loss_weight = None, tv_loss_weight
        if use_bilateral_grid:
            if rasterizer == "gut" and not fused_l2:
                raise ValueError("the bilateral grid is wired into the fused training steps (fastgs, or 3DGUT with fused_l2)")
            from .bilateral_grid import BilateralGrid
            from .fused_adam import WarmupExponentialLR
            gx, gy, gl = bilateral_grid_dims
            self.bilateral = BilateralGrid(sc.viewmats.shape[0], gx, gy, gl, device=device)
            self.bilateral.grids.grad = torch.zeros_like(self.bilateral.grids)
            self.bilateral_optimizer = torch.optim.Adam([self.bilateral.grids], lr=bilateral_grid_lr, eps=1e-15)
            self.bilateral_scheduler = WarmupExponentialLR(self.bilateral_optimizer, gamma=0.01 ** (1.0 / iterations), warmup_steps=1000, warmup_start_factor=0.01)
        # fused_l2: explicit forward/backward through the fused kernels (fused.py) instead of torch autograd over
        # the op-by-op mirror (rasterizer.py); gradients land directly in the flat bucket the all-reduce works on.
        self.fused_l2 = fused_l2
        self.loss_kind, self.lambda_dssim = loss, lambda_dssim  # "mse" | "l1_ssim" (trainer.cpp:115-128)
        self.bucket = lfs_dist.GradBucket(self.model.parameters(), deferred=self._deferred()) if (world > 1 or fused_l2) else None  # 2 = shN
        self.loss_acc = torch.zeros(1, device=device)
        self.inline_all_adam = True   # one view / one rank / MSE: all six parameters are updated inside the backward kernels (fused.backward_adam_all)
        self.fused_tail = True        # ... whose three per-Gaussian tail passes (SH backward + Adam, finish + Adam, next step's SH colours) are ONE launch (lfs_gut_train_step_ex)
        self._next_view = None        #     the view the NEXT step renders, when the trainer knows it (round-robin schedule, or train_step(next_views=...))
        self.pipelined = False        # True: with its SH Adam pass / SH colours on the library's side stream, under the next step's front end (lfs_gut_train_step_pipelined)
        self.cxx_step = True          # ... and that step is ONE C++ call without a host read on the critical path (gut_step.GutStep -> csrc/gut_step.hip);
        self._gut_step = None         #     False: the same kernels enqueued call by call from Python (fused.py; tests compare the two)
        self.inline_shN_adam = True   # see train_step; False keeps the SH backward and the optimizer separate (tests compare the two)
        self.iteration = 0
        self.last_plan: Optional[StepPlan] = None   # the form the last step took (plan_step): tests and tools read it
        self.last_n_isects = 0
        self._last_radii = None
        self._last_visible = None

    def _deferred(self):
        """bucket segments that stay out of the flat all-reduce: shN (2) while Adam does not read it; with the factored exchange sh0 (1) and shN always"""
        return [1, 2] if getattr(self, "factored_sh", False) else [2]

    def _on_resize(self) -> None:
        """The strategy replaced parameter tensors (densification): the flat gradient bucket has to follow."""
        if self._resize_suspended:      # (_refine_with_full_shN: shN is gathered right now - one rebuild when the shard is back)
            self._resize_pending = True
            return
        if self.bucket is not None:
            self.bucket = lfs_dist.GradBucket(self.model.parameters(), deferred=self._deferred())

    @property
    def last_visible(self):
        """bool [N]: Gaussians with radii > 0 in the last rendered view"""
        if self._last_radii is not None:
            return (self._last_radii[0] > 0).all(-1)
        return self._last_visible

    def _fastgs_settings(self, view: int):
        from .fastgs import FastGSSettings
        st = self._fg_settings.get(view)
        if st is None:  # intrinsics / camera centre read back once per view, not per step
            sc = self.scene
            K = sc.Ks[view].cpu()
            w2c = sc.viewmats[view]
            cam_pos = (-(w2c[:3, :3].T @ w2c[:3, 3])).contiguous()
            deg = self.model.get_active_sh_degree()
            st = FastGSSettings(cam_pos, (deg + 1) ** 2, sc.width, sc.height, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), 0.01, 1e10)
            self._fg_settings[view] = st
        st.active_sh_bases = (self.model.get_active_sh_degree() + 1) ** 2
        return st

    def _train_step_fastgs(self, targets, views, total_views):
        """The same step through the fastgs (EWA) rasterizer - the reference's default training path (trainer.cpp:656-760): fused
        preprocess -> blend -> [bilateral grid] -> loss ("mse" | "l1_ssim") -> blend backward -> preprocess backward (raw-parameter
        gradients straight into the flat bucket) -> all-reduce -> strategy.post_backward (ADC fed by the rasterizer's
        densification_info, or MCMC) -> fused Adam. Black background."""
        from .fastgs import render_and_backward as fg_step
        if self.bucket is None:
            self.bucket = lfs_dist.GradBucket(self.model.parameters(), deferred=self._deferred())
        params = self.model.parameters()
        N = params[0].shape[0]
        self.loss_acc.zero_()
        dens = None
        if self.strategy is not None and self.strategy_kind == "default" and self.iteration < self.strategy.params.stop_refine:
            if self.densification_info is None or self.densification_info.shape[1] != N:   # default_strategy.cpp:312-314
                self.densification_info = torch.zeros((2, N), device=self.device)
            dens = self.densification_info
        if len(views) > 1 and (not hasattr(self, "_fg_tmp") or self._fg_tmp[0].shape[0] != N):
            self._fg_tmp = [torch.empty_like(v) for v in self.bucket.views]
        inline = None   # one view on one rank, Adam reading shN: the SH backward inside lfs_fastgs_backward_adam updates shN itself
        if (self.inline_shN_adam and self.world == 1 and len(views) == 1 and self.strategy is None and self.iteration > 1000 and self.model.shN.shape[1] > 0
                and getattr(self.optimizer, "fused", False)):
            inline = self.optimizer.prepare_inline(self.model.shN)
        for k, v in enumerate(views):
            dst = self.bucket.views if k == 0 else self._fg_tmp
            _, _, self.last_n_isects = fg_step(self._fastgs_settings(v), self.scene.viewmats[v:v + 1].contiguous(), self.model, targets[k % len(targets)],
                                               1.0 / total_views, dst, self.loss_acc, densification_info=dens, loss=self.loss_kind,
                                               lambda_dssim=self.lambda_dssim, bilateral=self.bilateral, image_idx=v, adam_shN=inline)
            if k > 0:
                for a, b in zip(self.bucket.views, self._fg_tmp):
                    a.add_(b)
        if self.scale_reg > 0 or self.opacity_reg > 0:   # trainer.cpp:132-158, once per step and 1/world of it per rank
            with torch.no_grad():
                raw_scales, raw_opac = params[3].detach(), params[5].detach()
                if self.scale_reg > 0:
                    self.bucket.views[3].add_(torch.exp(raw_scales), alpha=self.scale_reg / self.world / raw_scales.numel())
                    self.loss_acc += self.scale_reg / self.world * torch.exp(raw_scales).mean()
                if self.opacity_reg > 0:
                    sg = torch.sigmoid(raw_opac)
                    self.bucket.views[5].add_((sg * (1 - sg)).view_as(self.bucket.views[5]), alpha=self.opacity_reg / self.world / raw_opac.numel())
                    self.loss_acc += self.opacity_reg / self.world * sg.mean()
        self._last_radii = None
        self.bucket.all_reduce(skip_deferred=self.iteration <= 1000)
        for p, gv in zip(params, self.bucket.views):
            p.grad = gv
        self._bilateral_step()
        if self.strategy is not None:
            if self.strategy_kind == "default":
                if dens is not None and self.world > 1 and self.strategy.is_refining(self.iteration):
                    lfs_dist.all_reduce_sum(dens)   # replicas must take the same densification decisions
                self.densification_info = self.strategy.post_backward(self.iteration, dens)
            else:
                self.strategy.post_backward(self.iteration)
            self.strategy.step(self.iteration)
        else:
            self.optimizer.step(self.iteration)
            self.scheduler.step()
        return self.loss_acc

    def _bilateral_step(self) -> None:
        """trainer.cpp:699-705, :758-761: TV regulariser (1/world of it per rank), sum of the ranks' grid gradients, Adam + warm-up schedule."""
        if self.bilateral is None:
            return
        if self.tv_loss_weight > 0:
            self.bilateral.tv_loss_fused(self.tv_loss_weight / self.world, self.loss_acc)
        if self.world > 1:
            lfs_dist.all_reduce_sum(self.bilateral.grids.grad)
        self.bilateral_optimizer.step()
        self.bilateral_optimizer.zero_grad(set_to_none=False)
        self.bilateral_scheduler.step()

    def _refine_with_full_shN(self, fn) -> None:
        """SH-sharded + a densification strategy: the strategy's index surgery (relocation, growth, pruning) addresses Gaussians globally, so for a
        refinement step shN and its Adam moments are all-gathered (3 x 180 B / Gaussian, every `refine_every` iterations: < 1 % of the traffic of
        the steps in between), the unchanged replicated strategy code runs - identically on every rank - and the result is re-sharded for the new N."""
        ex, opt = self.sh_exchange, self.optimizer
        old = self.model.shN
        n_before, grad_before = self.model.means.shape[0], old.grad
        self._resize_suspended, self._resize_pending = True, False
        st = opt._state(old)
        m_full, v_full = ex.gather_rows(st["exp_avg"]), ex.gather_rows(st["exp_avg_sq"])
        full = ex.gather_rows(old.detach()).contiguous().requires_grad_(True)
        opt.replace_param(2, old, full, lambda t: m_full if t is st["exp_avg"] else v_full)
        self.model.shN = full
        self.sh_exchange = None
        try:
            fn()
        finally:
            cur = self.model.shN
            st2 = opt._state(cur)
            ex2 = lfs_dist.ShExchange(self.model.means.shape[0], self.world, self.rank)
            ms, vs = ex2.shard(st2["exp_avg"]).clone(), ex2.shard(st2["exp_avg_sq"]).clone()
            shard = ex2.shard(cur.detach()).clone().requires_grad_(True)
            opt.replace_param(2, cur, shard, lambda t: ms if t is st2["exp_avg"] else vs)
            # nothing was added or removed and the strategy kept the tensor (relocation works in place): the shard keeps its gradient, so this
            # step's Adam update of shN happens as it does in the replicated layout and in the reference
            if cur is full and self.model.means.shape[0] == n_before and grad_before is not None:
                shard.grad = grad_before
            self.model.shN = shard
            self.sh_exchange = ex2
            self._resize_suspended = False
            if self._resize_pending or self.model.means.shape[0] != n_before:
                self._on_resize()

    def full_shN(self) -> torch.Tensor:
        """[N,K-1,3] on every rank (all-gathers the owners' rows when SH-sharded): export, evaluation."""
        self.join_pipeline()
        return self.model.shN.detach() if self.sh_exchange is None else self.sh_exchange.gather_rows(self.model.shN.detach())

    def export_model(self) -> SplatModel:
        """The complete model on this rank (SH-sharded: shN all-gathered; every rank must call it): what loader.save_ply / evaluate.evaluate take."""
        self.join_pipeline()
        m = self.model
        out = SplatModel(m.means.detach(), m.sh0.detach(), self.full_shN(), m.raw_scales.detach(), m.raw_quats.detach(), m.raw_opacities.detach(), m.max_sh_degree,
                         active_sh_degree=m.active_sh_degree)
        return out

    def camera(self, view: int) -> Camera:
        sc = self.scene
        return Camera(sc.viewmats[view:view + 1].contiguous(), sc.Ks[view:view + 1].contiguous(), sc.width, sc.height)

    def train_step(self, targets: List[torch.Tensor], views: Optional[List[int]] = None, views_all: Optional[List[List[int]]] = None,
                   next_views: Optional[List[int]] = None) -> float:
        """next_views: with an explicit `views`, the views the NEXT call will pass (a data loader knows them: src/training/dataloader.cpp prefetches) - the one-call step's
        fused tail then evaluates their SH colours on the side. Without `views` the round-robin schedule names them."""
        self._next_view = None
        if self.world == 1 and self.views_per_rank == 1:
            if views is None and views_all is None:
                self._next_view = lfs_dist.views_for_step(self.iteration + 1, self.rank, self.world, self.scene.viewmats.shape[0], self.views_per_rank)[0]
            elif next_views:
                self._next_view = int(next_views[0])
        out = self._train_step(targets, views, views_all)
        # the SH schedule, AFTER the backward / optimizer step of the iteration, where the strategies keep it (post_backward: mcmc.cpp:366-368,
        # default_strategy.cpp) - iteration 1000, 2000, ... still renders with the old degree, as the reference does; without a strategy the trainer does it
        if self.strategy is None and self.iteration % self.sh_degree_interval == 0 and self.model.active_sh_degree < self.model.max_sh_degree:
            self.model.active_sh_degree += 1
        return out

    def _train_step(self, targets: List[torch.Tensor], views: Optional[List[int]] = None, views_all: Optional[List[List[int]]] = None) -> float:
        """One optimisation step on this rank's share of the global view batch. `views` overrides this rank's views of the round-robin
        schedule; SH-sharded AND with the factored SH exchange every rank must know every rank's views: pass `views_all` (one list per rank) with an explicit
        schedule (a bare `views` raises there when world > 1).
        WHICH of the step forms runs is decided by plan_step (a pure function of the configuration, table-tested in tests/test_host_logic.py);
        each form is one method below."""
        self.iteration += 1
        if views_all is not None:
            views = views_all[self.rank]
        elif views is not None and self.sh_exchange is not None and self.world > 1:
            raise ValueError("SH-sharded: pass views_all (every rank's views), the SH owners evaluate them")
        elif views is not None and self.factored_sh and self.world > 1:
            # the factored exchange evaluates the SH backward over the views of ALL ranks: with only this rank's views overridden the cameras of the round-robin
            # schedule would be paired with the gathered dL/dcolour rows of other views - sh0 / shN / the direction term of dL/dmeans silently wrong
            raise ValueError("factored SH exchange: pass views_all (every rank's views), each rank evaluates the SH backward over all of them")
        if views is None:
            views = lfs_dist.views_for_step(self.iteration - 1, self.rank, self.world, self.scene.viewmats.shape[0], self.views_per_rank)
        self._views_all = views_all
        total_views = self.world * len(views)
        plan = self.last_plan = self._plan(len(views))   # (kept for tests and tools: which form the step took)
        if plan.path != "cxx_all":
            self.join_pipeline()
            if self._gut_step is not None:
                self._gut_step.colors_for = None   # (another step form is about to change the parameters: colours a fused tail left for this step are void)
        if plan.path == "fastgs":
            return self._train_step_fastgs(targets, views, total_views)
        if plan.path == "autograd":
            return self._step_autograd(targets, views, total_views)
        if not plan.inline_all:
            self.loss_acc.zero_()   # (the all-inline step stores the loss: no fill launch)
        if plan.path == "cxx_all":
            self._step_cxx_all(plan, targets, views, total_views)
        elif plan.path == "cxx_views":
            self._step_cxx_views(plan, targets, views, total_views)
        elif plan.path == "cxx_factored":
            self._step_cxx_factored(plan, targets, views, total_views)
        elif plan.path == "batch_views":
            self._step_batch_views(plan, targets, views, total_views)
        else:
            self._step_py_views(plan, targets, views, total_views)
        return self._finish_fused_step(plan)

    def _cxx_supported(self) -> bool:
        """Shapes the speculative C++ step does not take (more than 512 tile rows, the index-bit limit, debug bit 5: csrc/intersect.hip would return
        LFS_E_UNSUPPORTED) run the same kernels enqueued call by call from Python (the py_views form) instead of failing."""
        from .capi import load_library
        sc = self.scene
        return bool(load_library().lfs_gut_step_supported(int(self.model.means.shape[0]), int(sc.width), int(sc.height), 16))

    def _plan(self, n_views: int) -> "StepPlan":
        st = self.strategy
        return plan_step(rasterizer=self.rasterizer, fused_l2=self.fused_l2, world=self.world, force_collectives=bool(lfs_dist._FORCE),
                         sh_sharded=self.sh_exchange is not None, shard_rows=(self.sh_exchange.n if self.sh_exchange is not None else 0), n_views=n_views,
                         loss=self.loss_kind, strategy=(None if st is None else self.strategy_kind), refining=bool(st is not None and st.is_refining(self.iteration)),
                         iteration=self.iteration, has_shN=self.model.shN.shape[1] > 0, optimizer_fused=bool(getattr(self.optimizer, "fused", False)),
                         bilateral=self.bilateral is not None, inline_shN_adam=self.inline_shN_adam, inline_all_adam=self.inline_all_adam,
                         cxx_step=self.cxx_step and self.rasterizer != "fastgs" and self.fused_l2 and self._cxx_supported(), batch_views=self.batch_views,
                         factored_sh=self.factored_sh)

    def _gut(self):
        from .gut_step import GutStep
        if self._gut_step is None:
            self._gut_step = GutStep(self.device)
        return self._gut_step

    def _step_cxx_all(self, plan, targets, views, total_views) -> None:
        """One view, one rank, MSE, iteration > 1000: forward + backward + Adam on all six tensors as ONE C++ call (csrc/gut_step.hip), no gradient tensors."""
        inline_all = {name: self.optimizer.prepare_inline(getattr(self.model, name)) for name in ("shN", "means", "sh0", "raw_scales", "raw_quats", "raw_opacities")}
        gs, sc, v = self._gut(), self.scene, views[0]
        N = self.model.means.shape[0]
        self.last_n_isects = gs.train_step([p.detach() for p in self.model.parameters()], inline_all, self.model.get_active_sh_degree(), sc.width, sc.height,
                                           sc.viewmats[v], sc.Ks[v], self.bg, targets[0], 1.0 / total_views, self.loss_acc, self.scale_reg, self.opacity_reg,
                                           pipelined=self.pipelined, fused_tail=self.fused_tail,
                                           next_viewmat=None if self._next_view is None else sc.viewmats[self._next_view])
        self._last_radii = gs.view("radii", torch.int32, (1, N, 2))

    def join_pipeline(self) -> None:
        """After pipelined steps sh0 / shN and their moments belong to the library's side stream: the current stream waits for its last update (device-side, no host
        wait). Called before everything that touches them outside the pipelined step - other step forms, strategies, export, evaluation."""
        if self._gut_step is not None:
            self._gut_step.join()

    def _step_cxx_views(self, plan, targets, views, total_views) -> None:
        """Gradient-tensor form of the C++ step (data-parallel ranks with the north-star layout - replicated Gaussians, one all-reduce of the flat bucket -,
        single-rank steps while iteration <= 1000, and every step whose loss is not the folded MSE: L1 + D-SSIM, bilateral grid, MCMC): per view one speculative
        forward + two backward calls; the host looks at the forward's (pinned) counts once per view, after it has enqueued the forward. The loss kernels run
        between forward and backward on the image the forward left in the step workspace. The SH backward runs BEFORE the finish pass, so on the last view of a
        rank the shN segment (45 of 59 floats per Gaussian at degree 3) is on the wire, in 4 chunks, while the finish kernel still runs; the remaining 14 floats
        follow in all_reduce(). One view on one rank with Adam reading shN (plan.inline_shN): the SH backward applies shN's update itself."""
        inline = self.optimizer.prepare_inline(self.model.shN) if plan.inline_shN else None
        gs, sc = self._gut(), self.scene
        ps = [p.detach() for p in self.model.parameters()]
        deg, N = self.model.get_active_sh_degree(), self.model.means.shape[0]
        weight = 1.0 / total_views
        for k, v in enumerate(views):
            vm, Km, tgt = sc.viewmats[v], sc.Ks[v], targets[k % len(targets)]
            self.last_n_isects = gs.view_forward(ps, deg, sc.width, sc.height, vm, Km, self.bg)
            v_render, fold = None, None
            if self.bilateral is not None:   # clamp -> slice -> loss on the un-clamped result -> slice backward (fused.render_and_backward does the same)
                from .losses import loss_fwd_bwd
                render = gs.view("render", torch.float32, (sc.height, sc.width, 3))
                shown = self.bilateral.apply_fused(render, v, chw=False)
                v_shown = loss_fwd_bwd(self.loss_kind, shown, tgt, weight, self.loss_acc, chw=False, clamp=False, lambda_dssim=self.lambda_dssim)
                v_render = self.bilateral.apply_fused_backward(render, v, v_shown, chw=False)
            elif self.loss_kind == "l1_ssim":
                from .losses import photometric_loss_fwd_bwd
                v_render = photometric_loss_fwd_bwd(gs.view("render", torch.float32, (1, sc.height, sc.width, 3)), tgt, self.lambda_dssim, weight, self.loss_acc)
            else:
                fold = tgt                   # the clamped MSE is derived inside the rasterizer backward
            gs.view_backward_sh(ps, deg, sc.width, sc.height, vm, Km, self.bg, self.bucket.views, k > 0, target_chw=fold, weight=weight,
                                loss_acc=self.loss_acc, v_render=v_render, adam_shN=inline)
            if plan.multi and k == len(views) - 1 and self.iteration > 1000 and ps[2].numel():
                self.bucket.all_reduce_early([2], chunks=4)
            gs.view_backward_finish(ps, deg, sc.width, sc.height, vm, Km, self.bg, self.bucket.views, k > 0, target_chw=fold, weight=weight,
                                    loss_acc=self.loss_acc, scale_reg=self.scale_reg / self.world if k == 0 else 0.0,
                                    opacity_reg=self.opacity_reg / self.world if k == 0 else 0.0)
        self._last_radii = gs.view("radii", torch.int32, (1, N, 2))

    def _step_cxx_factored(self, plan, targets, views, total_views) -> None:
        """Replicated layout, factored SH exchange. Per view of this rank: speculative forward, loss, rasterizer backward + finish (means / scales / quaternions /
        opacities gradients into the bucket, dL/dcolour rows - clamp-masked - into the send buffer). Then: the rows of all ranks are all-gathered, the all-reduce of
        the 11 remaining floats per Gaussian starts behind them (asynchronously, on RCCL's stream), and the multi-view SH backward runs over the views of ALL ranks in
        rank-major order: sh0's gradient -> bucket, shN's Adam update inside (or its gradient -> bucket on refining iterations, or nothing while iteration <= 1000),
        the direction term of the means gradient -> added after the all-reduce has landed. Every rank executes the same sums in the same order."""
        from . import fused
        gs, sc = self._gut(), self.scene
        ps = [p.detach() for p in self.model.parameters()]
        deg, N = self.model.get_active_sh_degree(), self.model.means.shape[0]
        ex = self.color_exchange
        if ex is None or ex.n != N or ex.vpr != len(views):
            ex = self.color_exchange = lfs_dist.ColorGradExchange(N, self.world, self.rank, len(views), self.device)
        weight = 1.0 / total_views
        for k, v in enumerate(views):
            vm, Km, tgt = sc.viewmats[v], sc.Ks[v], targets[k % len(targets)]
            self.last_n_isects = gs.view_forward(ps, deg, sc.width, sc.height, vm, Km, self.bg)
            v_render, fold = None, None
            if self.bilateral is not None:
                from .losses import loss_fwd_bwd
                render = gs.view("render", torch.float32, (sc.height, sc.width, 3))
                shown = self.bilateral.apply_fused(render, v, chw=False)
                v_shown = loss_fwd_bwd(self.loss_kind, shown, tgt, weight, self.loss_acc, chw=False, clamp=False, lambda_dssim=self.lambda_dssim)
                v_render = self.bilateral.apply_fused_backward(render, v, v_shown, chw=False)
            elif self.loss_kind == "l1_ssim":
                from .losses import photometric_loss_fwd_bwd
                v_render = photometric_loss_fwd_bwd(gs.view("render", torch.float32, (1, sc.height, sc.width, 3)), tgt, self.lambda_dssim, weight, self.loss_acc)
            else:
                fold = tgt
            gs.view_backward_rows(ps, deg, sc.width, sc.height, vm, Km, self.bg, self.bucket.views, k > 0, ex.send[k], target_chw=fold, weight=weight,
                                  loss_acc=self.loss_acc, v_render=v_render, scale_reg=self.scale_reg / self.world if k == 0 else 0.0,
                                  opacity_reg=self.opacity_reg / self.world if k == 0 else 0.0)
        self._last_radii = gs.view("radii", torch.int32, (1, N, 2))
        # Round 6: the all-gather goes FIRST. Collectives of one communicator execute in issue order on RCCL's stream; the multi-view SH backward below waits for the
        # gathered rows and for nothing else, the 11-float all-reduce is only needed by the optimizer afterwards - issued second it travels UNDER the SH backward
        # (rounds 4 - 5 issued it first: the rows, and with them the SH backward, started one all-reduce later).
        rows = ex.gather()                                            # [world * views, N, 3], rank-major
        self.bucket.all_reduce_early([0, 3, 4, 5], chunks=1)          # 11 floats per Gaussian, on RCCL's stream while the SH backward runs
        every = self._views_all or [lfs_dist.views_for_step(self.iteration - 1, j, self.world, sc.viewmats.shape[0], len(views)) for j in range(self.world)]
        if rows.shape[0] != len(every) * len(views):                  # (one GPU forced through the collectives: the gathered rows are this rank's own)
            every = [list(views)]
        vms = torch.stack([sc.viewmats[v] for e in every for v in e]).contiguous()
        adam = self.optimizer.prepare_inline(self.model.shN) if plan.inline_shN else None
        want_shN_grad = adam is None and self.iteration > 1000 and self.model.shN.shape[1] > 0     # a refining iteration: relocation first, then the optimizer
        ex.v_dirs.zero_()
        fused.sh_model_bwd_views(deg, ps[0], vms, ps[1], ps[2], None, None, rows, self.bucket.views[1], self.bucket.views[2] if want_shN_grad else None, ex.v_dirs,
                                 False, adam=adam)
        self._pending_v_dirs = ex.v_dirs

    def _step_batch_views(self, plan, targets, views, total_views) -> None:
        """Several views per step on one rank: the SH stages run ONCE over all views (fused.render_views_and_backward), and shN's Adam update moves into
        that one SH backward when the optimizer would read the gradient anyway."""
        from .fused import render_views_and_backward
        inline_v = self.optimizer.prepare_inline(self.model.shN) if plan.inline_shN else None
        outs = render_views_and_backward([self.camera(v) for v in views], self.model, self.bg, [targets[k % len(targets)] for k in range(len(views))],
                                         1.0 / total_views, self.bucket.views, self.loss_acc, loss=self.loss_kind, lambda_dssim=self.lambda_dssim,
                                         scale_reg=self.scale_reg, opacity_reg=self.opacity_reg, adam_shN=inline_v, bilateral=self.bilateral,
                                         image_idxs=list(views))
        self.last_n_isects, self._last_radii = outs[-1].n_isects, outs[-1].radii

    def _step_py_views(self, plan, targets, views, total_views) -> None:
        """The fused kernels enqueued call by call from Python (fused.render_and_backward), view by view: the SH-sharded layout, cxx_step = False (the tests hold
        the C++ step to this path bit for bit), and the fallback of the C++ step."""
        from .fused import render_and_backward
        inline = self.optimizer.prepare_inline(self.model.shN) if (plan.inline_shN and not plan.inline_all) else None
        inline_shard = self.optimizer.prepare_inline(self.model.shN) if plan.inline_shard else None
        inline_all = None
        if plan.inline_all:
            inline_all = {name: self.optimizer.prepare_inline(getattr(self.model, name)) for name in ("shN", "means", "sh0", "raw_scales", "raw_quats", "raw_opacities")}
        every = None
        if self.sh_exchange is not None:  # what every rank renders at sub-step k (the owners evaluate SH for all of them)
            every = self._views_all or [lfs_dist.views_for_step(self.iteration - 1, j, self.world, self.scene.viewmats.shape[0], len(views)) for j in range(self.world)]
        for k, v in enumerate(views):
            vm_all = None if every is None else [self.scene.viewmats[e[k]:e[k] + 1].contiguous() for e in every]
            out = render_and_backward(self.camera(v), self.model, self.bg, targets[k % len(targets)], 1.0 / total_views,
                                      self.bucket.views, self.loss_acc, accumulate=k > 0, loss=self.loss_kind, lambda_dssim=self.lambda_dssim,
                                      # regularisers: once per step, and 1/world of them per rank (the all-reduce sums the ranks)
                                      scale_reg=self.scale_reg / self.world if k == 0 else 0.0,
                                      opacity_reg=self.opacity_reg / self.world if k == 0 else 0.0,
                                      sh_exchange=self.sh_exchange, viewmats_all=vm_all, adam_shN=inline, adam_shard=inline_shard, adam_all=inline_all,
                                      bilateral=self.bilateral, image_idx=v,
                                      # last view: scales / quats / opacities gradients are final before the SH backward starts - their all-reduce overlaps with it
                                      on_geometry_grads=(lambda: self.bucket.all_reduce_early([3, 4, 5])) if (self.world > 1 and k == len(views) - 1) else None)
            self.last_n_isects, self._last_radii = out.n_isects, out.radii

    def _finish_fused_step(self, plan):
        """What follows the backward of every fused step form: all-reduce of the flat bucket, the bilateral grid's own optimizer, strategy / optimizer step."""
        params = self.model.parameters()
        # the deferred segment (shN) stays out of the all-reduce while Adam does not read it (iteration <= 1000) and, SH-sharded, always
        self.bucket.all_reduce(skip_deferred=plan.skip_deferred)
        if plan.path == "cxx_factored":   # the SH direction term of dL/dmeans, summed over every rank's views by the multi-view SH backward: identical on all ranks
            self.bucket.views[0].add_(self._pending_v_dirs)
        self._bilateral_step()
        for p, gv in zip(params, self.bucket.views):
            p.grad = gv
        if self.strategy is not None:  # trainer.cpp:741-760: post_backward (may replace the parameter tensors) then step
            if self.sh_exchange is not None and self.strategy.is_refining(self.iteration):
                self._refine_with_full_shN(lambda: self.strategy.post_backward(self.iteration))
            else:
                self.strategy.post_backward(self.iteration)
            self.strategy.step(self.iteration)  # FusedAdam skips tensors without a gradient, as the reference's does after add_new_gs
        else:
            self.optimizer.step(self.iteration)
            self.scheduler.step()
        return self.loss_acc  # this rank's share of the loss (a 1-element tensor, read it after a sync)

    def _step_autograd(self, targets, views, total_views):
        """torch autograd over the op-by-op mirror of gs::training::rasterize (rasterizer.py): the reference's own structure, kept as the comparison path."""
        loss_value = None
        for k, v in enumerate(views):
            out = rasterize(self.camera(v), self.model, self.bg, 1.0, False, False, RenderMode.RGB)
            if self.loss_kind == "l1_ssim":
                from .losses import photometric_loss
                loss = photometric_loss(out.image, targets[k % len(targets)], self.lambda_dssim) / total_views
            else:
                loss = torch.nn.functional.mse_loss(out.image, targets[k % len(targets)]) / total_views
            if k == 0 and self.scale_reg > 0:
                loss = loss + self.scale_reg / self.world * self.model.get_scaling().mean()
            if k == 0 and self.opacity_reg > 0:
                loss = loss + self.opacity_reg / self.world * self.model.get_opacity().mean()
            loss.backward()
            loss_value = loss.detach()
            self.last_n_isects, self._last_visible = out.n_isects, out.visibility
        if self.bucket is not None:
            params = self.model.parameters()
            self.bucket.gather([p.grad for p in params])
            self.bucket.all_reduce(skip_deferred=self.iteration <= 1000)
            for p, gv in zip(params, self.bucket.views):
                p.grad = gv
        if self.strategy is not None:
            self.strategy.post_backward(self.iteration)
            self.strategy.step(self.iteration)
            return loss_value
        self.optimizer.step(self.iteration)
        self.optimizer.zero_grad(set_to_none=True)
        self.scheduler.step()
        return loss_value
