"""Host side of csrc/gut_step.hip: the --gut training step as ONE C call (lfs_gut_train_step) and its split form for gradient tensors
(lfs_gut_view_forward / lfs_gut_view_backward). Mirrors what Trainer::train_step does around rasterize() on the --gut path
(/root/reference/src/training/trainer.cpp:579-770, rasterization/rasterizer.cpp:200-344) - minus the host synchronisation of
gsplat/Intersect.cpp:75-76: the intersection lists live in a workspace sized for a CAPACITY, the count stays on the device, and this class
looks at the (pinned) counts only after the whole step has been enqueued. An attempt that did not fit (count above capacity, or a tile list longer
than the sort classes launched) updated nothing - the kernels check a device flag - and is simply run again with a larger workspace.

torch is the allocator here (one uint8 tensor per workspace, one pinned int64[3] for the counts) and owns the stream; nothing else.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch

from .capi import LfsError, check, load_library, stream

GROUPS = ("means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opacities")   # FusedAdam group order (strategy_utils.cpp:35-40)


class StepArgs(C.Structure):  # lfs_gut_step_args
    _fields_ = [("N", C.c_uint32), ("K", C.c_uint32), ("sh_degree", C.c_uint32), ("image_width", C.c_uint32), ("image_height", C.c_uint32), ("tile_size", C.c_uint32),
                ("means", C.c_void_p), ("sh0", C.c_void_p), ("shN", C.c_void_p), ("raw_scales", C.c_void_p), ("raw_quats", C.c_void_p), ("raw_opacities", C.c_void_p),
                ("exp_avg", C.c_void_p * 6), ("exp_avg_sq", C.c_void_p * 6), ("adam", (C.c_float * 6) * 6),
                ("viewmat", C.c_void_p), ("Kmat", C.c_void_p), ("background", C.c_void_p), ("target_chw", C.c_void_p),
                ("loss_weight", C.c_float), ("scale_reg", C.c_float), ("opacity_reg", C.c_float), ("loss", C.c_void_p)]


class StepLayout(C.Structure):  # lfs_gut_step_layout
    _fields_ = [(k, C.c_size_t) for k in ("bytes", "render", "alpha", "last_ids", "radii", "means2d", "depths", "colors", "quats", "scales", "opacities",
                                          "tile_offsets", "flatten_ids", "isect_ids", "counts", "abort_flag")]


class GutStep:
    """One workspace + the capacity bookkeeping for steps of one (N, W, H, tile) shape on one device."""

    def __init__(self, device, tile: int = 16, initial_capacity: Optional[int] = None, margin: float = 1.25):
        self.device = torch.device(device)
        self.tile = tile
        self.margin = margin
        self.capacity = int(initial_capacity) if initial_capacity else 0
        self.assumed_longest = 1024
        self.ws: Optional[torch.Tensor] = None
        self.layout: Optional[StepLayout] = None
        self.shape = None
        self.counts = torch.zeros(3, dtype=torch.int64).pin_memory()
        self._stamp = 0
        self.n_isects = 0
        self.longest = 0
        self.wait_poll_s = 5.0    # how long _wait spins on the pinned counts before it falls back to a stream synchronisation
        self.retries = 0          # attempts that did not fit (each one is re-run): a few right after start-up or a densification, none in steady state
        self.colors_for = None    # fused tail (lfs_gut_train_step_ex): what the workspace's SH colours were evaluated for by the previous step - (viewmat pointer, N, K,
                                  # degree, workspace pointer) - or None; a step for exactly that skips its SH colour kernel
        self.colour_launches_saved = 0

    # ---- workspace --------------------------------------------------------------------------------------------------------------------------
    def _ensure(self, N: int, W: int, H: int) -> None:
        lib = load_library()
        if self.capacity <= 0:
            self.capacity = max(4 * N, 1 << 16)       # first guess: ~4 tile entries per Gaussian (SYN-B: 4.4); the first step corrects it
        shape = (N, W, H, self.capacity, int(lib.lfs_get_debug_flags()))   # (debug bit 4, the deterministic backward, adds a 64-bit accumulator to the workspace)
        if self.shape == shape and self.ws is not None:
            return
        lay = StepLayout()
        check(lib.lfs_gut_step_layout_for(C.c_uint32(N), C.c_uint32(W), C.c_uint32(H), C.c_uint32(self.tile), C.c_int64(self.capacity), C.byref(lay)), "gut_step_layout_for")
        if self.ws is None or self.ws.numel() < lay.bytes:
            self.ws = None                            # (release the old block before the larger one is requested)
            self.ws = torch.empty(int(lay.bytes), dtype=torch.uint8, device=self.device)
        self.layout, self.shape = lay, shape

    def _grow(self, n_isects: int, longest: int) -> None:
        self.capacity = max(int(n_isects * self.margin) + 1024, self.capacity)
        self.assumed_longest = max(int(longest * self.margin), self.assumed_longest)
        self.shape = None

    def view(self, name: str, dtype, shape: Sequence[int]) -> torch.Tensor:
        """A tensor view of one workspace region (render [H,W,3] f32, alpha [H,W] f32, radii [N,2] i32, ...): valid until the next step."""
        off = getattr(self.layout, name)
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        return self.ws[off:off + nbytes].view(dtype).view(*shape)

    # ---- argument block -----------------------------------------------------------------------------------------------------------------------
    def _args(self, params: Sequence[torch.Tensor], sh_degree: int, W: int, H: int, viewmat: torch.Tensor, Kmat: torch.Tensor, bg: Optional[torch.Tensor],
              target_chw: Optional[torch.Tensor], weight: float, scale_reg: float, opacity_reg: float, loss_acc: Optional[torch.Tensor],
              adam: Optional[Dict[str, dict]]) -> StepArgs:
        means, sh0, shN, raw_scales, raw_quats, raw_opac = params
        for t in (*params, viewmat, Kmat, bg, target_chw, loss_acc):
            if t is not None and (not t.is_cuda or not t.is_contiguous() or t.dtype != torch.float32):
                raise LfsError("gut_step: every tensor must be a contiguous float32 CUDA (HIP) tensor")
        a = StepArgs()
        a.N, a.K, a.sh_degree = means.shape[0], 1 + shN.shape[1], sh_degree
        a.image_width, a.image_height, a.tile_size = W, H, self.tile
        a.means, a.sh0, a.shN = means.data_ptr(), sh0.data_ptr(), (shN.data_ptr() if shN.numel() else None)
        a.raw_scales, a.raw_quats, a.raw_opacities = raw_scales.data_ptr(), raw_quats.data_ptr(), raw_opac.data_ptr()
        if adam is not None:
            for k, name in enumerate(GROUPS):
                d = adam.get(name)
                if d is None:   # (view_backward_sh takes shN's state alone)
                    continue
                a.exp_avg[k], a.exp_avg_sq[k] = d["exp_avg"].data_ptr(), d["exp_avg_sq"].data_ptr()
                for j, key in enumerate(("lr", "beta1", "beta2", "eps", "bc1_rcp", "bc2_sqrt_rcp")):
                    a.adam[k][j] = d[key]
        a.viewmat, a.Kmat = viewmat.data_ptr(), Kmat.data_ptr()
        a.background = bg.data_ptr() if bg is not None else None
        a.target_chw = target_chw.data_ptr() if target_chw is not None else None
        a.loss_weight, a.scale_reg, a.opacity_reg = weight, scale_reg, opacity_reg
        a.loss = loss_acc.data_ptr() if loss_acc is not None else None
        return a

    def _wait(self) -> bool:
        """-> did the attempt fit? (the counts were written by the scan kernel long before the host got here)"""
        lib = load_library()
        n, lg = C.c_int64(0), C.c_int64(0)
        # the stamp usually is there already; when the step sits behind other work on the stream (the previous step's all-reduce, a first-collective RCCL
        # initialisation that can take longer than any fixed timeout) the host simply waits for the stream and looks again - it only gives up when the stream
        # has DRAINED and the stamp of this step is still missing
        rc = lib.lfs_gut_step_wait(C.c_void_p(self.counts.data_ptr()), C.c_int64(self._stamp), C.c_double(self.wait_poll_s), C.byref(n), C.byref(lg))
        if rc != 0:
            torch.cuda.current_stream(self.device).synchronize()
            rc = lib.lfs_gut_step_wait(C.c_void_p(self.counts.data_ptr()), C.c_int64(self._stamp), C.c_double(1.0), C.byref(n), C.byref(lg))
        if rc != 0:
            raise LfsError("gut_step: the stream has drained and the intersection counts of this step never arrived in pinned host memory")
        self.n_isects, self.longest = int(n.value), int(lg.value)
        return bool(lib.lfs_gut_step_fits(n, lg, C.c_int64(self.capacity), C.c_int64(self.assumed_longest)))

    def _after_fit(self) -> None:
        # stay ahead of a slowly growing scene: enlarge BEFORE the next step would overflow (a re-run costs a whole step, a reallocation nothing)
        if self.n_isects > 0.92 * self.capacity or self.longest > 0.92 * self._class_limit():
            self._grow(self.n_isects, self.longest)

    def _class_limit(self) -> int:
        a = self.assumed_longest
        return 1024 if a <= 1024 else 4096 if a <= 4096 else 16384 if a <= 16384 else 1 << 62

    # ---- the step -------------------------------------------------------------------------------------------------------------------------------
    def train_step(self, params: Sequence[torch.Tensor], adam: Dict[str, dict], sh_degree: int, W: int, H: int, viewmat: torch.Tensor, Kmat: torch.Tensor,
                   bg: Optional[torch.Tensor], target_chw: torch.Tensor, weight: float, loss_acc: torch.Tensor, scale_reg: float = 0.0,
                   opacity_reg: float = 0.0, pipelined: bool = False, fused_tail: bool = False, next_viewmat: Optional[torch.Tensor] = None) -> int:
        """Forward + backward + Adam on all six parameter tensors, in place; *loss_acc = weight * mse. `adam[name]` = FusedAdam.prepare_inline(param) for
        the six names of GROUPS. Returns n_isects.
        pipelined: lfs_gut_train_step_pipelined - the SH Adam pass of this step and the SH colours of the next run on the library's side stream, under the next step's
        front end. Same results; sh0 / shN and their moments then belong to that stream until join() (every other method of this class joins by itself).
        fused_tail: lfs_gut_train_step_ex - SH backward, the six Adam updates and (next_viewmat: the view the NEXT step renders, a tensor that stays untouched until then)
        the next step's SH colours in one launch; a following step for exactly that view (same tensor, same N / K / degree / workspace) skips its SH colour kernel."""
        lib = load_library()
        N = params[0].shape[0]
        K = 1 + params[2].shape[1]
        fn, what = (lib.lfs_gut_train_step_pipelined, "gut_train_step_pipelined") if pipelined else (lib.lfs_gut_train_step, "gut_train_step")
        for attempt in range(4):
            self._ensure(N, W, H)
            a = self._args(params, sh_degree, W, H, viewmat, Kmat, bg, target_chw, weight, scale_reg, opacity_reg, loss_acc, adam)
            self._stamp += 1
            if fused_tail and not pipelined:
                key = lambda vm: (vm.data_ptr(), N, K, sh_degree, self.ws.data_ptr(), params[0].data_ptr(), params[1].data_ptr(), params[2].data_ptr())   # (replaced parameter tensors void the colours too)
                ready = self.colors_for is not None and self.colors_for == key(viewmat)
                nxt = next_viewmat if (next_viewmat is not None and K <= 16) else None
                self.colors_for = None   # (whatever happens below, the colours of THIS view are consumed / overwritten)
                check(lib.lfs_gut_train_step_ex(C.byref(a), C.c_void_p(nxt.data_ptr()) if nxt is not None else None, C.c_int(int(ready)), C.c_int64(self.capacity),
                                                C.c_int64(self.assumed_longest), C.c_void_p(self.ws.data_ptr()), C.c_size_t(self.ws.numel()),
                                                C.c_void_p(self.counts.data_ptr()), C.c_int64(self._stamp), stream()), "gut_train_step_ex")
                if self._wait():
                    if nxt is not None:
                        self.colors_for = key(nxt)
                    self.colour_launches_saved += int(ready)
                    self._after_fit()
                    return self.n_isects
                # the attempt did not fit: its tail returned without writing anything, so colours that were ready still are - unless _grow replaces the workspace below
                if ready:
                    self.colors_for = key(viewmat)
                self.retries += 1
                self._grow(self.n_isects, self.longest)
                continue
            self.colors_for = None
            check(fn(C.byref(a), C.c_int64(self.capacity), C.c_int64(self.assumed_longest), C.c_void_p(self.ws.data_ptr()),
                     C.c_size_t(self.ws.numel()), C.c_void_p(self.counts.data_ptr()), C.c_int64(self._stamp), stream()), what)
            if self._wait():
                self._after_fit()
                return self.n_isects
            self.retries += 1
            self._grow(self.n_isects, self.longest)
        raise LfsError("gut_step: the step did not fit its workspace after 4 attempts")

    @staticmethod
    def join() -> bool:
        """The current stream waits (on the device) for the side stream's last SH update of a pipelined step. -> was there one?"""
        rc = load_library().lfs_gut_pipeline_join(stream())
        if rc < 0:
            raise LfsError("gut_pipeline_join failed")
        return rc == 1

    def view_forward(self, params: Sequence[torch.Tensor], sh_degree: int, W: int, H: int, viewmat, Kmat, bg) -> int:
        """Forward of one view into the workspace (render / alpha / radii via .view()); re-run on overflow. Returns n_isects."""
        lib = load_library()
        N = params[0].shape[0]
        self.colors_for = None   # (this forward evaluates its own colours into the workspace)
        for attempt in range(4):
            self._ensure(N, W, H)
            a = self._args(params, sh_degree, W, H, viewmat, Kmat, bg, None, 0.0, 0.0, 0.0, None, None)
            self._stamp += 1
            check(lib.lfs_gut_view_forward(C.byref(a), C.c_int64(self.capacity), C.c_int64(self.assumed_longest), C.c_void_p(self.ws.data_ptr()),
                                           C.c_size_t(self.ws.numel()), C.c_void_p(self.counts.data_ptr()), C.c_int64(self._stamp), stream()), "gut_view_forward")
            if self._wait():
                return self.n_isects
            self.retries += 1
            self._grow(self.n_isects, self.longest)
        raise LfsError("gut_step: the view did not fit its workspace after 4 attempts")

    def _backward_call(self, fn_name: str, params, sh_degree, W, H, viewmat, Kmat, bg, grads, accumulate, target_chw, weight, loss_acc, v_render, scale_reg, opacity_reg,
                       adam=None):
        lib = load_library()
        a = self._args(params, sh_degree, W, H, viewmat, Kmat, bg, target_chw, weight, scale_reg, opacity_reg, loss_acc, adam)
        for g in grads:
            if not g.is_cuda or not g.is_contiguous():
                raise LfsError("gut_step: gradient tensors must be contiguous CUDA (HIP) tensors")
        gp = (C.c_void_p * 6)(*[g.data_ptr() if g.numel() else None for g in grads])
        ws, nb = C.c_void_p(self.ws.data_ptr()), C.c_size_t(self.ws.numel())
        if fn_name == "lfs_gut_view_backward_finish":
            check(lib.lfs_gut_view_backward_finish(C.byref(a), C.c_int64(self.capacity), gp, C.c_int(int(accumulate)), ws, nb, stream()), fn_name)
            return
        if v_render is not None:
            v_render = v_render.contiguous()
        check(getattr(lib, fn_name)(C.byref(a), C.c_int64(self.capacity), C.c_void_p(v_render.data_ptr()) if v_render is not None else None, gp,
                                    C.c_int(int(accumulate)), ws, nb, stream()), fn_name)

    def view_backward(self, params: Sequence[torch.Tensor], sh_degree: int, W: int, H: int, viewmat, Kmat, bg, grads: List[torch.Tensor], accumulate: bool, *,
                      target_chw: Optional[torch.Tensor] = None, weight: float = 0.0, loss_acc: Optional[torch.Tensor] = None,
                      v_render: Optional[torch.Tensor] = None, scale_reg: float = 0.0, opacity_reg: float = 0.0) -> None:
        """Backward of the view view_forward() left in the workspace, into the six gradient tensors (group order): written, or added to with `accumulate`.
        With target_chw the clamped MSE (weight) is folded into the backward and loss_acc += it; otherwise v_render [H,W,3] is dL/d(render)."""
        self._backward_call("lfs_gut_view_backward", params, sh_degree, W, H, viewmat, Kmat, bg, grads, accumulate, target_chw, weight, loss_acc, v_render,
                            scale_reg, opacity_reg)

    def view_backward_sh(self, params, sh_degree, W, H, viewmat, Kmat, bg, grads, accumulate, *, target_chw=None, weight=0.0, loss_acc=None, v_render=None,
                         adam_shN: Optional[dict] = None) -> None:
        """First half of view_backward: rasterizer backward + SH backward. grads[1] (sh0) and grads[2] (shN) are final for this view when it has run.
        adam_shN (FusedAdam.prepare_inline(shN); one view per step): shN is updated in place by the SH backward instead and grads[2] is left alone."""
        self._backward_call("lfs_gut_view_backward_sh", params, sh_degree, W, H, viewmat, Kmat, bg, grads, accumulate, target_chw, weight, loss_acc, v_render, 0.0, 0.0,
                            adam=None if adam_shN is None else {"shN": adam_shN})

    def view_backward_rows(self, params, sh_degree, W, H, viewmat, Kmat, bg, grads, accumulate, rows_out: torch.Tensor, *, target_chw=None, weight=0.0, loss_acc=None,
                           v_render=None, scale_reg: float = 0.0, opacity_reg: float = 0.0) -> None:
        """The view's backward for the factored gradient exchange (dist.ColorGradExchange): grads[0] (means, without the SH direction term), grads[3..5] written / added
        to, dL/dcolour -> rows_out [N,3], masked with the clamp of the SH colours (colour > 0) - what the multi-view SH backward of every rank takes as it is."""
        lib = load_library()
        a = self._args(params, sh_degree, W, H, viewmat, Kmat, bg, target_chw, weight, scale_reg, opacity_reg, loss_acc, None)
        if not rows_out.is_cuda or not rows_out.is_contiguous() or rows_out.numel() != 3 * params[0].shape[0]:
            raise LfsError("gut_step: rows_out must be a contiguous CUDA (HIP) tensor [N,3]")
        gp = (C.c_void_p * 6)(*[g.data_ptr() if g.numel() else None for g in grads])
        if v_render is not None:
            v_render = v_render.contiguous()
        check(lib.lfs_gut_view_backward_rows(C.byref(a), C.c_int64(self.capacity), C.c_void_p(v_render.data_ptr()) if v_render is not None else None, gp,
                                             C.c_int(int(accumulate)), C.c_void_p(rows_out.data_ptr()), C.c_void_p(self.ws.data_ptr()), C.c_size_t(self.ws.numel()),
                                             stream()), "gut_view_backward_rows")
        N = params[0].shape[0]
        rows_out.view(N, 3).mul_(self.view("colors", torch.float32, (N, 3)) > 0)   # clamp_min backward of rasterizer.cpp:262 (colours of invisible Gaussians: rows are 0 anyway)

    def view_backward_finish(self, params, sh_degree, W, H, viewmat, Kmat, bg, grads, accumulate, *, target_chw=None, weight=0.0, loss_acc=None,
                             scale_reg: float = 0.0, opacity_reg: float = 0.0) -> None:
        """Second half: accumulator rows + dL/d(dirs) -> grads[0], grads[3..5]; loss_acc += the fused MSE of the first half (when target_chw was given)."""
        self._backward_call("lfs_gut_view_backward_finish", params, sh_degree, W, H, viewmat, Kmat, bg, grads, accumulate, target_chw, weight, loss_acc, None,
                            scale_reg, opacity_reg)
