"""Data-parallel sharding of training views (new capability: the reference is single-process,
single-GPU). One process per GPU; Gaussians are replicated; rank r renders the
views  step*world + r ; ONE all-reduce (RCCL over xGMI, backend "nccl" on ROCm) sums the flat
gradient bucket before the identical Adam step on every rank, so the replicated parameters stay
bit-identical across ranks. Two layouts: fully replicated (59 floats / Gaussian in
the all-reduce at SH degree 3) and, by default for the fused 3DGUT step, SH-sharded (class ShExchange:
14 floats / Gaussian in the all-reduce, shN and its Adam state owned by one rank each).
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; world == 1 -> no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or _FORCE) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:  # LFS_DIST_BACKEND=gloo: several ranks on one GPU (tests / smoke runs; collectives staged through the host)
            backend = os.environ.get("LFS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        # a bounded timeout: a rank that dies before it enters a collective must not leave its peers blocked for the backend's default 10 - 30 minutes
        import datetime
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=int(os.environ.get("LFS_DIST_TIMEOUT_S", "300"))))
    return rank, world, local_rank


_AGREE_GROUP = None


def agree_all(ok_here: bool) -> bool:
    """True when EVERY rank passes True. Runs over a gloo side group on CPU tensors (created on first use, by all ranks together): the agreement does not travel over
    the communicator whose collectives are being tried - after a caught RCCL error that communicator may be aborted or poisoned."""
    global _AGREE_GROUP
    if not _active():
        return bool(ok_here)
    if _AGREE_GROUP is None:
        import datetime
        _AGREE_GROUP = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=int(os.environ.get("LFS_DIST_TIMEOUT_S", "300"))))
    f = torch.tensor([1.0 if ok_here else 0.0])
    dist.all_reduce(f, op=dist.ReduceOp.MIN, group=_AGREE_GROUP)
    return float(f) == 1.0


# LFS_DIST_FORCE_COLLECTIVES=1: issue every collective even at world size 1 (tests/test_gpu_rccl_world1.py: the RCCL code path - device
# tensors handed to backend "nccl" - executes on a single GPU; RCCL refuses two ranks per device, so this is the only way to run it there).
_FORCE = bool(os.environ.get("LFS_DIST_FORCE_COLLECTIVES"))


def _active() -> bool:
    return dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


# ---- per-collective accounting (bench.py --gpus N prints it): calls, payload bytes and, when timing is on, device milliseconds -------------
STATS: dict = {}
_TIMING = False
_PENDING: list = []


def stats_enable(timing: bool) -> None:
    global _TIMING
    STATS.clear(); _PENDING.clear()
    _TIMING = timing


def _account(kind: str, t: torch.Tensor):
    st = STATS.setdefault(kind, {"calls": 0, "bytes": 0, "ms": 0.0})
    st["calls"] += 1
    st["bytes"] += t.numel() * t.element_size()
    if _TIMING and t.is_cuda:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _PENDING.append((kind, a, b))
        return b
    return None


def stats_collect() -> dict:
    """resolves the recorded event pairs (synchronises) and returns {kind: {calls, bytes, ms}}"""
    if _PENDING:
        torch.cuda.synchronize()
        for kind, a, b in _PENDING:
            STATS[kind]["ms"] += a.elapsed_time(b)
        _PENDING.clear()
    return {k: dict(v, ms=round(v["ms"], 4)) for k, v in STATS.items()}


def ranks_seen(device) -> int:
    """how many ranks a sum all-reduce of ones reaches (bench line: proves the process group spans what --gpus claims)"""
    if not dist.is_initialized():
        return 1
    t = torch.ones(1, dtype=torch.float32, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(round(float(t.item())))


def views_for_step(step: int, rank: int, world: int, n_views: int, views_per_rank: int = 1) -> List[int]:
    """Round-robin assignment: global batch of world*views_per_rank views per step, disjoint across ranks."""
    base = step * world * views_per_rank
    return [(base + rank * views_per_rank + k) % n_views for k in range(views_per_rank)]


class GradBucket:
    """Flat fp32 bucket [sum(numel)] with per-parameter views, all-reduced in one collective.

    `deferred` lists parameter indices whose gradient the optimizer may ignore for a while (the higher-degree
    SH group: FusedAdam skips it while iteration <= 1000, fused_adam.cpp:68-70). They are laid out at the END of
    the flat buffer so that `all_reduce(skip_deferred=True)` reduces only the prefix the optimizer will read:
    56 MB instead of 236 MB per step at 1M Gaussians / SH degree 3 - on 8 GPUs over xGMI that is the difference
    between a collective that costs ~10 % of a step and one that costs most of it."""

    def __init__(self, params: List[torch.Tensor], deferred: Optional[List[int]] = None):
        self.sizes = [p.numel() for p in params]
        self.shapes = [p.shape for p in params]
        deferred = sorted(set(deferred or []))
        order = [i for i in range(len(params)) if i not in deferred] + deferred
        total = sum(self.sizes)
        self.flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
        self.views: List[Optional[torch.Tensor]] = [None] * len(params)
        self.offsets = [0] * len(params)
        self._early = None
        o = 0
        self.active_numel = total
        for i in order:
            if deferred and i == deferred[0]:
                self.active_numel = o
            n, s = self.sizes[i], self.shapes[i]
            self.offsets[i] = o
            self.views[i] = self.flat[o:o + n].view(s)
            o += n

    def gather(self, grads: List[Optional[torch.Tensor]]) -> None:
        for v, g in zip(self.views, grads):
            if g is None:
                v.zero_()
            else:
                v.copy_(g)

    def _reduce(self, buf: torch.Tensor, kind: str, async_op: bool = False):
        end = _account(kind, buf)
        work = None
        if _staged(buf):
            host = buf.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            buf.copy_(host)
        else:
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=async_op)
        if end is not None and not async_op:
            end.record()
        return work, end

    def all_reduce(self, average: bool = False, skip_deferred: bool = False) -> None:
        """Sum over ranks of the whole bucket (or its non-deferred prefix), minus whatever `all_reduce_early` already has in flight."""
        if _active():
            hi = self.active_numel if skip_deferred else self.flat.numel()
            if self._early is not None:       # [lo_e, hi_e) is being reduced on RCCL's stream: reduce the pieces around it, then wait
                lo_e, hi_e, works = self._early
                self._early = None
                if lo_e > 0:
                    self._reduce(self.flat[:min(lo_e, hi)], "all_reduce")
                if hi > hi_e:
                    self._reduce(self.flat[hi_e:hi], "all_reduce")
                for work, end in works:
                    if work is not None:
                        work.wait()           # the current stream waits for the collective's stream
                    if end is not None:
                        end.record()
            else:
                self._reduce(self.flat[:hi], "all_reduce")
            if average:
                self.flat[:hi].div_(dist.get_world_size())

    def all_reduce_early(self, indices: List[int], chunks: int = 1) -> None:
        """Start the all-reduce of the parameters `indices` (contiguous in the flat buffer) NOW, asynchronously, in `chunks` collectives of equal size:
        RCCL runs them on its own stream, ordered after what the current stream has enqueued so far, while the caller keeps launching kernels that do
        not touch those gradients. `all_reduce()` later reduces the rest and waits. The replicated data-parallel step sends the SH gradients this way
        (45 of 59 floats per Gaussian, final before the finish pass runs; several chunks so that the first bytes move while the last are still being
        summed up by the ring). No-op at world 1."""
        if not _active() or self._early is not None:
            return
        lo = min(self.offsets[i] for i in indices)
        hi = max(self.offsets[i] + self.sizes[i] for i in indices)
        assert sum(self.sizes[i] for i in indices) == hi - lo, "early segment must be contiguous in the bucket"
        chunks = max(1, min(int(chunks), hi - lo))
        step = -(-(hi - lo) // chunks)
        works = []
        for a in range(lo, hi, step):
            works.append(self._reduce(self.flat[a:min(a + step, hi)], "all_reduce_early", async_op=True))
        self._early = (lo, hi, works)


def _staged(t: torch.Tensor) -> bool:
    """gloo has no device collectives for every op: device tensors are staged through the host (tests on one GPU; RCCL needs no staging)."""
    return t.is_cuda and dist.get_backend() == "gloo"


class ShExchange:
    """SH-sharded data parallelism: the higher-degree SH coefficients (shN: 45 of a Gaussian's 59 floats at degree 3) and their
    Adam state are NOT replicated. Rank r owns the rows [r*S, (r+1)*S) of shN; per training step

        radii  (8 B / Gaussian / view)  --all-to-all-->  owners            (which Gaussians each view sees)
        owners evaluate SH for their rows for EVERY rank's view;  colours (12 B) --all-to-all--> the rendering ranks
        ... rasterize forward / backward on the own view ...
        dL/dcolour (12 B)  --all-to-all-->  owners;  owners run the SH backward for every view into their shard gradient
        (and add the view-direction term of dL/dmeans into their rows of the replicated means gradient)

    so the shN gradient never crosses xGMI: the step's all-reduce covers 14 floats per Gaussian instead of 59, and Adam touches
    1/world of shN on each rank. Per Gaussian and step a rank moves 32 B through all-to-all + 56 B through the all-reduce,
    against 236 B for the replicated layout; the SH arithmetic is the same in total (N Gaussian-views per rank either way).
    The kernels are passed in (the signatures of fused.sh_model_fwd_views / sh_model_bwd_views: ONE launch covers the views of all ranks,
    coefficient rows are read once) so the exchange logic is testable on CPU with the oracle as stand-in (tests/test_dist_gloo.py)."""

    def __init__(self, n_gaussians: int, world: int, rank: int):
        self.N, self.world, self.rank = n_gaussians, world, rank
        self.S = (n_gaussians + world - 1) // world
        self.r0 = min(rank * self.S, n_gaussians)
        self.r1 = min(self.r0 + self.S, n_gaussians)
        self.n = self.r1 - self.r0

    def shard(self, t: torch.Tensor) -> torch.Tensor:
        return t[self.r0:self.r1]

    def _pad(self, t: torch.Tensor) -> torch.Tensor:
        """[N, ...] -> [world, S, ...] (zero rows after N; a plain view when the ranks divide N - no fill, no copy)"""
        if self.N == self.world * self.S and t.is_contiguous():
            return t.view((self.world, self.S) + tuple(t.shape[1:]))
        out = torch.zeros((self.world * self.S,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        out[:self.N] = t
        return out.view((self.world, self.S) + tuple(t.shape[1:]))

    def _all_to_all(self, send: torch.Tensor, defer: bool = False):
        """send[j] goes to rank j; recv[j] came from rank j. defer=True (device collectives only): the exchange is enqueued asynchronously and
        (recv, finish) is returned - the caller's stream keeps running kernels that do not need `recv` and calls finish() before the first one
        that does (the collective runs on the library's own stream behind everything enqueued so far)."""
        send = send.contiguous()
        if not _active():
            out = send.clone()
            return (out, lambda: None) if defer else out
        end = _account("all_to_all", send)
        if _staged(send):
            host, out = send.cpu(), torch.empty(send.shape, dtype=send.dtype)
            dist.all_to_all_single(out, host)
            recv = out.to(send.device)
            if end is not None:
                end.record()
            return (recv, lambda: None) if defer else recv
        recv = torch.empty_like(send)
        if not defer:
            dist.all_to_all_single(recv, send)
            if end is not None:
                end.record()
            return recv
        work = dist.all_to_all_single(recv, send, async_op=True)

        def finish():
            work.wait()          # a stream-side wait: the host does not block
            if end is not None:
                end.record()
        return recv, finish

    def gather_rows(self, shard_rows: torch.Tensor) -> torch.Tensor:
        """Owner shards [n_r, ...] -> the full [N, ...] tensor on every rank (export, evaluation, strategies)."""
        pad = torch.zeros((self.S,) + tuple(shard_rows.shape[1:]), dtype=shard_rows.dtype, device=shard_rows.device)
        pad[:self.n] = shard_rows
        if not _active():
            return pad[:self.N]
        end = _account("all_gather", pad)
        if _staged(pad):
            parts = [torch.empty(pad.shape, dtype=pad.dtype) for _ in range(self.world)]
            dist.all_gather(parts, pad.cpu())
            out = torch.cat(parts)[:self.N].to(pad.device)
        else:
            parts = [torch.empty_like(pad) for _ in range(self.world)]
            dist.all_gather(parts, pad)
            out = torch.cat(parts)[:self.N]
        if end is not None:
            end.record()
        return out

    def begin_radii(self, radii):
        """Start the exchange of this rank's radii [N,2] as soon as the projection has produced them; hand the result to forward(radii_pending=)."""
        return self._all_to_all(self._pad(radii), defer=True)

    def forward(self, deg: int, means, sh0, shN_shard, radii, viewmats_all, sh_fwd_views, radii_pending=None, defer: bool = False):
        """radii [N,2] int32 of THIS rank's view; viewmats_all[j] = the [1,4,4] view matrix rank j renders now.
        -> (colours [N,3] of this rank's view, ctx for backward). `sh_fwd_views`: fused.sh_model_fwd_views (one launch for all views).
        Overlap (the fused step): radii_pending = begin_radii(radii) issued right after the projection, so that exchange runs next to the
        intersection count; defer=True returns the colours before their exchange has finished - call finish_forward(ctx) before the first kernel
        that reads them (the intersection scatter and sort run in between)."""
        if radii_pending is not None:
            radii_recv, fin = radii_pending
            fin()
        else:
            radii_recv = self._all_to_all(self._pad(radii))                 # [world, S, 2]: view j's radii for my rows
        vms = torch.cat([v.reshape(1, 4, 4) for v in viewmats_all]).contiguous()
        if self.n:
            colors_send = sh_fwd_views(deg, self.shard(means), vms, self.shard(sh0), shN_shard, radii_recv)   # [world, S, 3]
        else:
            colors_send = torch.zeros((self.world, self.S, 3), dtype=means.dtype, device=means.device)
        if defer:
            recv, fin = self._all_to_all(colors_send, defer=True)
            colors = recv.view(self.world * self.S, 3)[:self.N]
            assert colors.is_contiguous()   # a row prefix of the receive buffer: no copy may read it before finish_forward
            return colors, (radii_recv, colors_send, vms, fin)
        colors = self._all_to_all(colors_send).view(self.world * self.S, 3)[:self.N]
        return colors.contiguous(), (radii_recv, colors_send, vms, None)

    @staticmethod
    def finish_forward(ctx) -> None:
        if ctx is not None and len(ctx) > 3 and ctx[3] is not None:
            ctx[3]()

    def backward(self, ctx, deg: int, means, sh0, shN_shard, viewmats_all, v_colors, g_sh0, g_shN_shard, g_means, accumulate: bool, sh_bwd_views,
                 adam=None) -> None:
        """v_colors [N,3] = dL/dcolours of this rank's view. Writes (accumulate False) or adds to g_shN_shard and MY rows of g_sh0 (the
        other rows are zeroed / left alone: the all-reduce brings their owners' values), adds dL/d(dirs) into my rows of g_means.
        `adam` (FusedAdam.prepare_inline of the shard; needs accumulate False): the shard is updated in place, g_shN_shard is not written."""
        radii_recv, colors_send, vms = ctx[:3]
        v_recv = self._all_to_all(self._pad(v_colors))                      # [world, S, 3]: view j's dL/dcolour for my rows
        if not accumulate:
            g_sh0.zero_()
        if self.n:
            sh_bwd_views(deg, self.shard(means), vms, self.shard(sh0), shN_shard, radii_recv, colors_send, v_recv, self.shard(g_sh0), g_shN_shard,
                         self.shard(g_means), accumulate, adam)


class ColorGradExchange:
    """The FACTORED exchange of the SH gradients in the replicated data-parallel layout (north star: replicated Gaussians, per-rank forward / backward, collective
    before the fused Adam step). Per view the gradient of the SH coefficients is rank one per Gaussian - basis(direction) x dL/dcolour, 16 x 3 numbers from 3 - and
    every rank knows every rank's camera: instead of all-reducing the assembled [N, 16, 3] tensors (192 of the 236 bytes per Gaussian of the flat bucket at degree 3)
    the ranks ALL-GATHER the dL/dcolour rows of their views (12 bytes per Gaussian and view, masked by the clamp of the rendering rank) and each evaluates the
    multi-view SH backward (lfs_sh_model_bwd_views) over the views of ALL ranks in rank-major order - the same sums on every rank, in the same order, so the
    replicated parameters stay bit-identical - with shN's Adam update inside. sh0's gradient and the direction term of the means gradient fall out of the same pass;
    what remains in the flat all-reduce are means (rasterizer part), scales, quaternions, opacities: 11 floats per Gaussian.
    8 ranks x 1 view at 1 M Gaussians: 96 MB gathered + 44 MB all-reduced per rank instead of 236 MB all-reduced."""

    def __init__(self, n_gaussians: int, world: int, rank: int, views_per_rank: int, device):
        self.n, self.world, self.rank, self.vpr = n_gaussians, world, rank, views_per_rank
        self.send = torch.zeros(views_per_rank, n_gaussians, 3, dtype=torch.float32, device=device)
        self.recv = torch.zeros(world * views_per_rank, n_gaussians, 3, dtype=torch.float32, device=device) if world > 1 or _FORCE else self.send
        self.v_dirs = torch.zeros(n_gaussians, 3, dtype=torch.float32, device=device)   # sum over all views of dL/d(direction): added to the means gradient after the all-reduce

    def gather(self) -> torch.Tensor:
        """-> [world * views_per_rank, N, 3]: the rows of every rank's views, rank-major (view k of rank r at r * views_per_rank + k)"""
        if not _active():
            return self.send
        end = _account("all_gather", self.recv)
        if _staged(self.send):
            host = torch.empty(self.recv.shape, dtype=torch.float32)
            dist.all_gather_into_tensor(host, self.send.cpu())
            self.recv.copy_(host)
        else:
            dist.all_gather_into_tensor(self.recv, self.send)
        if end is not None:
            end.record()
        return self.recv


def all_reduce_sum(t: torch.Tensor) -> None:
    """In-place sum over ranks of a replicated-side tensor (bilateral-grid gradient, densification_info); no-op at world 1."""
    if _active():
        end = _account("all_reduce_small", t)
        if _staged(t):
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            t.copy_(host)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        if end is not None:
            end.record()


def barrier() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return value
