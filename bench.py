#!/usr/bin/env python3
"""Headline benchmark: train-images/sec (fwd + bwd + Adam) on the 3DGUT hot path.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1] (SYN-B of SURVEY.md §8d): 1 000 000 synthetic Gaussians,
1920x1080, SH degree 3, 16x16 tiles, 64 orbit cameras, lrs of default_optimization_params.json,
MSE loss against a fixed random target (rasterizer-only metric). One "step" = every rank renders
one view, backpropagates, the parameter gradients are summed over ranks (one RCCL all-reduce of
the flat 59*N-float bucket: the north-star layout, the headline for N > 1; the SH-sharded layout
is timed right after it and reported under "sh_sharded") and every rank applies the fused Adam
step: weak scaling, value = world * K / max-over-ranks(time).  Inputs are resident in HBM before the timed region.

Prints ONE JSON line on rank 0, with
  roofline     : the dominant kernel (by HIP-event time inside the timed region) against the
                 8 TB/s HBM peak, algorithmic bytes from SURVEY.md §8d with the measured V and I;
  cpu_baseline : the CPU oracle (oracle/, "port") timed on this box's host cores for ONE training
                 image of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md (spec; ~6.3 TB/s achievable)


def algorithmic_bytes(N: int, V: int, I: int, P: int, T: int, K: int, adam_elems: int) -> dict:
    """Compulsory HBM bytes per launch, keyed by the event names of csrc/."""
    return {
        "projection_ut": 52 * N + 24 * V,
        "activations_projection_ut": 44 * N + 32 * N + 8 * N + 12 * V,          # raw parameters in; activated values, radii, means2d + depth of the visible out
        "sh_fwd_pack": 13 * N + (12 * K + 12) * V + 44 * V + 96 * V,           # sh_fwd + the activated geometry of the visible in, record + culling record out
        "sh_fwd": 13 * N + (12 * K + 12) * V,
        "isect_count_scan": 32 * N + 4 * T,
        "isect_scatter": 28 * N + 12 * I,
        "isect_tile_sort": 24 * I,
        "raster_pack": 60 * N + 64 * N + 32 * N,
        # the workgroup of a tile gathers id + 32-B culling record ONCE per list entry and shares it with its 4 cells through LDS (36 I); each cell keeps
        # ~45 % of its tile's entries on SYN-B (measured: 8.1 M cell-list entries for 4.46 M intersections), 8 B each
        "raster_cull": 36 * I + 8 * int(1.8 * I),
        "raster_fwd": 60 * I + 20 * P,
        "raster_bwd": 60 * I + 24 * P + 56 * N + 112 * V,   # (with the MSE loss folded in it reads render + target instead of v_render: + 12 P, not charged)
        "raster_finish": 64 * N + 44 * N + 56 * N,
        "sh_bwd": 24 * V + 12 * K * V + 12 * K * N + 12 * K * V + 12 * V,
        # single-view steps: SH backward with shN's Adam update inside (no shN gradient tensor): read p, m, v + write p, m, v of shN,
        # dL/dcolour + colour + means + radii of the visible, sh0 gradient out
        "sh_bwd_adam": 6 * 12 * (K - 1) * N + 48 * V + 12 * N + 12 * V,
        "activations_fwd": 2 * 32 * N, "activations_bwd": 32 * N + 32 * N + 32 * N + 32 * N,
        "mse_loss": 36 * P, "photometric_loss": 36 * P + 2 * 36 * P,
        "adam_multi": 28 * adam_elems,
        "adam": 28 * adam_elems,
        # all-inline step: accumulator row (64) + means, raw and activated quats / scales / opacity (76) + 11 floats of each Adam moment (88) in, 3 x 44 out
        "finish_adam": (64 + 76 + 88 + 132) * N,
        # steps that need gradient tensors: accumulator row (64) + means, raw / activated quaternions, scales, opacity (60) in, five gradient tensors out (56)
        "finish_grads": (64 + 60 + 56) * N,
        # fastgs (EWA) path, SURVEY.md §8f row 1 - same accounting as the 3DGUT kernels: 64-B blend record + 4-B id per intersection, per-pixel
        # state, the 64-B accumulator rows; preprocess reads the 44 B of raw geometry and writes record (64) + mean2d / conic / bounds (32)
        "fastgs_preprocess": 44 * N + 96 * V + 8 * N,
        "fastgs_scatter": 16 * V + 12 * I,
        "fastgs_tile_sort": 24 * I,
        "fastgs_cull": 36 * I + 8 * int(1.8 * I),
        "fastgs_blend_fwd": 68 * I + 20 * P,
        "fastgs_blend_bwd": 68 * I + 28 * P + 64 * N + 64 * V,
        "fastgs_preprocess_bwd": 64 * N + 96 * V + 44 * N,
    }


def reference_torch_impl_baseline(threads: int):
    """SURVEY.md 8(d) "CPU baseline timing": the reference's OWN CPU code (tests/torch_impl.cpp:38,147,296,324, compiled in place into oracle/_ref by
    oracle/Makefile - no reference source is copied) on the host cores:
      * config 1 / SYN-A (10k Gaussians, 256x256, SH degree 0): quat_scale_to_covar_preci + fully_fused_projection + spherical_harmonics + isect_tiles,
        forward + autograd backward, median of 10 (of 3 when a pass takes longer than 2 s);
      * projection + SH only (forward + backward) at 1 M Gaussians / 1080p / SH degree 3 - the per-element `.item()` loop of isect_tiles
        (torch_impl.cpp:370-397) makes the intersection impractical there; median of 3 (seconds each; 1 when a pass takes longer than 12 s).
    torch_impl has no compositing and no unscented transform: this times what the reference can run on a CPU, it is not the benchmarked path.
    Returns None when the prebuilt library is absent."""
    import numpy as np

    import oracle
    from lichtfeld_studio_amd import scenes
    try:
        oracle.ref_lib()
    except Exception:
        return None
    med = lambda x: float(np.sort(x)[len(x) // 2])
    # libtorch's CPU kernels do not scale past a few cores at these sizes - with one thread per core of a 256-core host the SYN-A pass takes 12 s instead of 1 s
    # (measured, profiles/r04/cpu_baseline_threads.txt) - so the leg runs on at most 16 threads and reports that count; and it is BOUNDED: the repeats shrink when
    # a pass is slow, so the default bench run stays within a few minutes on any host.
    threads = max(1, min(int(threads), 16))
    sc = scenes.syn_a()
    run_a = lambda reps: oracle.ref_cpu_stage_fwd_bwd(sc.means.numpy(), sc.raw_quats.numpy(), np.exp(sc.raw_scales.numpy()), sc.sh0.numpy(), 0, sc.viewmats[0].numpy(),
                                                      sc.Ks[0].numpy(), sc.width, sc.height, True, threads, reps)
    sec, n_isects = run_a(3)
    sec = list(sec)
    if med(sec) < 2.0:
        sec += list(run_a(7)[0])
    out = {"value": round(med(sec) * 1e3, 2), "unit": "ms", "cores": threads, "kind": "reference",
           "sample": f"tests/torch_impl.cpp: quat_scale_to_covar_preci + fully_fused_projection + spherical_harmonics + isect_tiles, forward + autograd backward, "
                     f"SYN-A (10000 Gaussians, 256x256, SH deg 0, {n_isects} intersections), median of {len(sec)}, libtorch x{threads} threads"}
    try:
        sb = scenes.syn_b(n=1_000_000, n_views=1)
        coeffs = np.concatenate([sb.sh0.numpy(), sb.shN.numpy()], 1)
        run_b = lambda reps: oracle.ref_cpu_stage_fwd_bwd(sb.means.numpy(), sb.raw_quats.numpy(), np.exp(sb.raw_scales.numpy()), coeffs, 3, sb.viewmats[0].numpy(), sb.Ks[0].numpy(),
                                                          sb.width, sb.height, False, threads, reps)[0]
        sec1 = list(run_b(1))
        if sec1[0] < 12.0:
            sec1 += list(run_b(2))
        out["projection_sh_1M"] = {"value": round(med(sec1), 3), "unit": "s", "cores": threads,
                                   "sample": f"the same without isect_tiles at 1 000 000 Gaussians, 1920x1080, SH deg 3 (SYN-B view 0), forward + autograd backward, median of {len(sec1)}"}
    except Exception as e:
        out["projection_sh_1M"] = {"value": None, "sample": f"failed: {e}"}
    return out


def cpu_baseline(scene, view: int, target, threads: int, hip_step: dict | None = None) -> dict:
    """Oracle ("port" of the reference CUDA kernels) on the host cores: ONE training image of the
    same workload — activations, UT projection, SH, tile intersection + stable sort, compositing fwd,
    MSE gradient, compositing bwd, SH bwd, activation bwd (oracle/pipeline.py), Adam on all 59*N parameters.
    With `hip_step` (the HIP path's results for the same view from the same initial parameters) the oracle's
    results are not thrown away: `parity_vs_oracle` carries the comparison, so every bench line has its own parity evidence."""
    import numpy as np

    import oracle
    from oracle import pipeline
    oracle.lib()
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    sa = pipeline.scene_arrays(scene)
    t0 = time.perf_counter()
    orc = pipeline.train_image(sa, view, target.numpy())
    params = [sa["means"], sa["sh0"], sa["shN"], sa["raw_scales"], sa["raw_quats"], sa["raw_opacities"]]
    names = ["means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opacities"]
    for p, k in zip(params, names):
        z = np.zeros_like(p)
        oracle.adam_step(p, z, z, np.ascontiguousarray(orc["grads"][k], dtype=np.float32), 1e-3, 0.9, 0.999, 1e-15, 10.0, 31.6)
    dt = time.perf_counter() - t0
    out = {"value": 1.0 / dt, "unit": "train-images/sec", "cores": threads, "kind": "port",
           "sample": f"1 training image of the same workload (N={sa['means'].shape[0]}, {sa['width']}x{sa['height']}, SH deg {sa['sh_degree']}; "
                     f"V={int(orc['visible'].sum())}, I={int(len(orc['flatten_ids']))}): oracle fwd+bwd+Adam, {dt:.1f} s wall, OpenMP x{threads}"}
    if hip_step is not None:
        out["parity_vs_oracle"] = dict(pipeline.compare_step(hip_step, orc), view=view,
                                       note="HIP fused step vs oracle, same view, initial parameters; activations computed on each side")
    return out


def hip_reference_step(trainer, view: int, target_dev) -> dict:
    """One forward + backward of the HIP path at `view` from the trainer's CURRENT parameters, results on the host (for parity_vs_oracle): through the same
    C++ entry points the timed steps use (gut_step.GutStep -> lfs_gut_view_forward / lfs_gut_view_backward; the Adam-inline step enqueues the same kernels)."""
    from lichtfeld_studio_amd.gut_step import GutStep
    gs = GutStep(trainer.device)
    sc = trainer.scene
    params = [p.detach() for p in trainer.model.parameters()]
    grads = [torch.zeros_like(p) for p in params]
    loss = torch.zeros(1, device=target_dev.device)
    deg, N = trainer.model.get_active_sh_degree(), params[0].shape[0]
    n_isects = gs.view_forward(params, deg, sc.width, sc.height, sc.viewmats[view], sc.Ks[view], trainer.bg)
    gs.view_backward(params, deg, sc.width, sc.height, sc.viewmats[view], sc.Ks[view], trainer.bg, grads, False, target_chw=target_dev, weight=1.0, loss_acc=loss)
    torch.cuda.synchronize()
    names = ["means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opacities"]
    res = {"render": gs.view("render", torch.float32, (1, sc.height, sc.width, 3)).cpu().numpy(), "alpha": gs.view("alpha", torch.float32, (1, sc.height, sc.width, 1)).cpu().numpy(),
           "radii": gs.view("radii", torch.int32, (1, N, 2)).cpu().numpy(), "loss": float(loss), "grads": {k: g.cpu().numpy() for k, g in zip(names, grads)}}
    res["n_isects"] = n_isects
    return res


def measure(trainer, targets, args, world: int, device, profile: bool) -> dict:
    """W warm-up steps (the last up to 3 with every kernel scope timed: the per-kernel table, which names the dominant kernel), then exactly K timed steps
    bracketed by barrier + synchronize, max over ranks. Inside the timed region only the dominant kernel is bracketed with events (two hipEventRecord per
    step), so the event overhead of the other scopes stays out of the measured throughput."""
    from lichtfeld_studio_amd import capi
    from lichtfeld_studio_amd import dist as lfs_dist
    table_steps = min(3, args.warmup) if profile else 0
    # The host thread has to stay ahead of the device for K x ~10 launches; a generation-2 pass of Python's cycle collector over torch's object graph is a multi-millisecond
    # stall of exactly that thread (one run in three of lease 31 showed 31.7 instead of 24.4 ms for 20 steps with every kernel at its usual duration). Collect HERE, in front
    # of the warm-up steps, and keep the collector off until the timed region has ended - no work of the step is skipped. Not in front of the timed region itself: the
    # collection is a pause of tens of milliseconds in which the device idles and drops its clocks (measured, lease 32: 808 instead of 843 img/s, raster_bwd 0.42 instead of
    # 0.40 ms, 12 runs each). LFS_BENCH_GC=1 leaves the collector alone (the bench as it was).
    gc_was_on = gc.isenabled()
    if not os.environ.get("LFS_BENCH_GC"):
        gc.collect()
        gc.disable()
    for _ in range(args.warmup - table_steps):
        trainer.train_step(targets)
    table, coll_ms = {}, {}
    if table_steps:
        torch.cuda.synchronize()
        capi.profile_collect()
        capi.profile_filter(None)
        capi.profile_enable(True)
        # the collectives' device time is measured here as well (two timing events per collective), NOT inside the timed region
        lfs_dist.stats_enable(timing=world > 1 or bool(os.environ.get("LFS_DIST_FORCE_COLLECTIVES")))
        for _ in range(table_steps):
            trainer.train_step(targets)
        capi.profile_enable(False)
        table = capi.profile_collect()
        coll_ms = {k: v["ms"] / table_steps for k, v in lfs_dist.stats_collect().items()}
    dom = max(table.items(), key=lambda kv: kv[1][0])[0] if table else None
    lfs_dist.barrier()
    torch.cuda.synchronize()
    seen = lfs_dist.ranks_seen(device)
    if seen != world:
        raise SystemExit(f"the process group reaches {seen} ranks, --gpus says {world}")
    lfs_dist.stats_enable(timing=False)   # calls and payload bytes of the timed steps only (no events)
    if dom:
        capi.profile_filter(dom)
        capi.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.train_step(targets)
    lfs_dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if gc_was_on:
        gc.enable()
    kernels = {}
    if dom:
        capi.profile_enable(False)
        kernels = capi.profile_collect()
        capi.profile_filter(None)
    elapsed = lfs_dist.max_over_ranks(elapsed, device)
    coll = lfs_dist.stats_collect()   # this rank's collectives inside the timed steps: calls, payload bytes (device ms: from the profiled warm-up steps above)
    return dict(elapsed=elapsed, table=table, table_steps=table_steps, kernels=kernels, dom=dom, coll=coll, coll_ms=coll_ms, seen=seen)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="syn-b", choices=["syn-a", "syn-b", "syn-c", "syn-d"])
    ap.add_argument("--n", "--gaussians", dest="n", type=int, default=0, help="override the number of Gaussians (use --gaussians under torch.distributed.run, whose parser claims --n*)")
    ap.add_argument("--views-per-rank", type=int, default=1)
    ap.add_argument("--rasterizer", default="gut", choices=["gut", "fastgs"], help="gut = the north-star 3DGUT path (default); fastgs = the reference's default EWA rasterizer (SURVEY.md §8f row 1)")
    ap.add_argument("--loss", default="mse", choices=["mse", "l1_ssim"], help="mse = rasterizer-only metric of SURVEY.md §8d (default); l1_ssim = the reference's photometric loss")
    ap.add_argument("--start-iteration", type=int, default=3000,
                    help="iteration counter the timed steps start from. Default 3000 = the steady state of a 7k-iteration run: SH degree 3 active and "
                         "FusedAdam updating shN (the reference skips shN while iteration <= 1000, fused_adam.cpp:68-70; pass 0 for that cheaper phase)")
    ap.add_argument("--strategy", default="none", choices=["none", "mcmc"],
                    help="mcmc = BASELINE.json configs[4]: strategies.MCMC (mcmc_optimization_params.json: SGLD noise every step, relocation of dead Gaussians "
                         "every 100 iterations with the Relocation kernel, scale / opacity regularisers), max_cap = the scene's Gaussian count")
    ap.add_argument("--bilateral-grid", action="store_true", help="BASELINE.json configs[4]: per-image 16x16x8 bilateral grid between render and loss (+ its TV loss and Adam)")
    ap.add_argument("--replicated", action="store_true", help="multi-GPU: only the headline layout (replicated Gaussians, one all-reduce of the 59-float bucket); skips the SH-sharded side line")
    ap.add_argument("--sh-sharded", action="store_true", help="measure ONLY the SH-sharded layout, as the headline - with LFS_DIST_FORCE_COLLECTIVES=1 this runs its collectives on ONE GPU")
    ap.add_argument("--factored", action="store_true", help="measure ONLY the replicated layout with the factored SH exchange (dist.ColorGradExchange), as the headline - with "
                                                            "LFS_DIST_FORCE_COLLECTIVES=1 this runs its collectives on ONE GPU")
    ap.add_argument("--no-config4", action="store_true", help="multi-GPU: skip the BASELINE configs[3] side line (SYN-C, 3 M Gaussians, 1600x1200, 8 views per rank)")
    ap.add_argument("--path", default="step", choices=["step", "ops"],
                    help="step (default) = the C++ training step (csrc/gut_step.hip, one host call per step); ops = the DROP-IN route: the sequence "
                         "rasterizer.cpp:224-344 makes through the reference-signature C++ wrappers of _lfs_torch_ops.so, op by op under torch autograd, then six "
                         "adam_step_wrapper launches (fused_adam.cpp:22-95) - what a reference build linking csrc/torch_ops.cpp gets without touching its trainer")
    ap.add_argument("--no-fused-tail", action="store_true", help="single-GPU MSE step: lfs_gut_train_step (SH backward + Adam, finish + Adam, SH colours as three launches) instead of "
                                                                 "lfs_gut_train_step_ex (round 6: one launch for all three, the next view's colours from the rows in registers)")
    ap.add_argument("--pipeline", action="store_true", help="single-GPU MSE step: lfs_gut_train_step_pipelined (round 6: the SH Adam pass and the SH colours on the library's side stream, "
                                                            "under the next step's projection / tile lists / culling; measured -4.5 % at best, profiles/r06/pipeline/README.md)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ops-route", action="store_true", help="skip the 16 extra steps of the drop-in route (the `ops_route` block of the line)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()

    import lichtfeld_studio_amd as lfs
    from lichtfeld_studio_amd import capi, scenes
    from lichtfeld_studio_amd import dist as lfs_dist
    from lichtfeld_studio_amd.trainer import GutTrainer
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    capi.load_library()
    rank, world, local_rank = lfs_dist.init_distributed()
    lfs_dist.stats_enable(timing=False)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if os.environ.get("LFS_DIST_BACKEND") == "gloo":   # smoke mode: all ranks share the visible device(s)
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    maker = {"syn-a": scenes.syn_a, "syn-b": scenes.syn_b, "syn-c": scenes.syn_c, "syn-d": scenes.syn_d}[args.workload]
    kw = {"n": args.n} if args.n else {}
    if args.workload == "syn-a":
        kw["sh_degree"] = 0
    scene = maker(**kw)
    n_views = scene.viewmats.shape[0]
    extra = {}
    if args.strategy == "mcmc":
        from lichtfeld_studio_amd import strategies
        extra = dict(strategy="mcmc", opt_params=strategies.OptimizationParameters(iterations=30000, max_cap=scene.N))
    if args.path == "ops":
        if world != 1 or args.rasterizer != "gut" or args.strategy != "none" or args.bilateral_grid:
            raise SystemExit("--path ops measures the single-GPU 3DGUT drop-in route (no strategy, no bilateral grid)")
        from lichtfeld_studio_amd import torch_ops_route
        torch_ops_route.install()   # rasterizer.py / fused_adam.py now call the compiled reference-signature wrappers of _lfs_torch_ops.so

    def make_trainer(sh_sharded, factored=False, sc=None, vpr=None):
        ops_route = args.path == "ops"   # op by op under torch autograd + six adam_step_wrapper launches: what the reference's own trainer would execute
        return GutTrainer(sc or scene, device, iterations=30000 if args.strategy == "mcmc" else 7000, world=world, rank=rank, views_per_rank=vpr or args.views_per_rank,
                          loss=args.loss, rasterizer=args.rasterizer, sh_sharded=sh_sharded, factored_sh=factored, use_bilateral_grid=args.bilateral_grid,
                          fused_l2=not ops_route, fused_adam=not ops_route, **extra)
    # N > 1: the HEADLINE is the north-star layout - replicated Gaussians, per-rank forward / backward, one all-reduce of the flat gradient bucket before the
    # fused Adam step ("dpN-replicated"). The SH-sharded layout (dist.ShExchange: shN and its Adam state owned by one rank each, 14 instead of 59 floats per
    # Gaussian in the all-reduce) is measured right after it and reported beside it under "sh_sharded"; --sh-sharded / --replicated restrict the run to one.
    # Round 4: a third layout, "factored" - replicated Gaussians as the north star says, but the SH gradients travel as dL/dcolour rows (all-gather) and are assembled
    # by every rank (dist.ColorGradExchange): 11 instead of 59 floats per Gaussian in the all-reduce. It is measured beside the other two; --factored makes it the headline.
    # Which of the two exchanges of the replicated layout is the HEADLINE at N > 1: the factored one moves world x views_per_rank x 12 bytes per Gaussian, the flat
    # all-reduce ~2 x 236 whatever the number of views - factored while world x views_per_rank <= 16 (8 GPUs x 1 view: 96 + 44 MB per rank instead of 236 MB through the
    # ring), flat beyond (BASELINE configs[3]: 64 views per step). --replicated / --factored force one of them; the other is reported beside the headline.
    headline_sharded = bool(args.sh_sharded) and not args.replicated
    forced = bool(os.environ.get("LFS_DIST_FORCE_COLLECTIVES"))
    auto_factored = lambda vpr: world > 1 and world * vpr <= 16
    headline_factored = (not args.replicated and not headline_sharded and args.rasterizer == "gut" and args.path == "step"
                         and ((bool(args.factored) and (world > 1 or forced)) or (not args.factored and auto_factored(args.views_per_rank))))
    trainer = make_trainer(headline_sharded, headline_factored)
    trainer.pipelined = bool(args.pipeline)        # (only the one-call step form, plan "cxx_all", has these variants; every other form joins the side stream by itself)
    trainer.fused_tail = not args.no_fused_tail
    if headline_factored and not args.factored:
        # the factored exchange has never run between two GPUs (no multi-GPU box was ever available to this repository): one guarded trial step, on EVERY rank, before
        # anything is timed; if any rank's collective library refuses it the headline falls back to the flat all-reduce and says so
        # Round 5 (review of round 4): a rank that fails BEFORE it enters a data collective would leave its peers blocked inside it, and the flag all-reduce below
        # would never be reached. So the agreement comes first: every rank tries the two collectives of the layout on a few bytes (all_gather_into_tensor, an
        # asynchronous all_reduce), the ranks agree on the outcome with one MIN all-reduce - the one collective every torch.distributed backend has - and only a
        # layout that every rank accepted runs its full trial step (whose remaining failure modes - an unsupported problem shape, a kernel error - are functions of
        # replicated state and hit all ranks alike, outside the collectives).
        # Round 6 (ADVICE): the agreement itself runs over a gloo side group on CPU tensors (dist.agree_all) - not over the communicator under test, which a caught
        # RCCL error may have left aborted - and every group carries a bounded timeout (LFS_DIST_TIMEOUT_S, 300 s): a rank that dies inside the probe ends the run
        # within minutes instead of leaving its peers blocked.
        def agree(ok_here: int) -> bool:
            return lfs_dist.agree_all(bool(ok_here))
        agree(1)   # (creates the side group on every rank BEFORE anything can fail one-sidedly)
        ok = 1
        try:
            from lichtfeld_studio_amd import dist as lfs_dist   # (the layout's own collective wrappers: gloo ranks stage through the host, RCCL goes direct)
            probe = lfs_dist.ColorGradExchange(4, world, rank, 1, device)
            probe.send.fill_(1.0)
            rows = probe.gather()
            one = torch.ones(4, device=device)
            lfs_dist.all_reduce_sum(one)
            torch.cuda.synchronize()
            ok = int(float(rows.sum()) == 12.0 * world and float(one[0]) == float(world))
        except Exception:   # noqa: BLE001  (whatever the collective library throws: the answer is "not this layout")
            ok = 0
        if agree(ok):
            ok = 1
            try:
                trainer.iteration = args.start_iteration
                trainer.train_step([scenes.target_image(scene.height, scene.width, seed=43).to(device)])
                torch.cuda.synchronize()
            except Exception:   # noqa: BLE001
                ok = 0
        else:
            ok = 0
        headline_factored = agree(ok)
        trainer = make_trainer(False, headline_factored)   # (a fresh model either way: the trial step updated the parameters)
    if args.strategy == "mcmc" and args.start_iteration == 3000:
        # the warm-up must contain one refinement step (iteration 3000: relocation + its torch index kernels, whose first use loads ~20 code
        # objects at 20 - 200 ms each); the timed window then holds warm steps only, one of them (every 100th) a refinement step
        args.start_iteration = 3000 - max(2, args.warmup - 2)
    trainer.iteration = args.start_iteration
    from lichtfeld_studio_amd import fused as _fused
    targets = [scenes.target_image(scene.height, scene.width, seed=43).to(device)]
    hip_step = None
    if world == 1 and not args.no_cpu_baseline and args.rasterizer == "gut" and args.path == "step":
        hip_step = hip_reference_step(trainer, 0, targets[0])   # before any update: the oracle starts from the same parameters

    profile = not args.no_profile
    m = measure(trainer, targets, args, world, device, profile)
    elapsed, table, table_steps, kernels, dom, coll, coll_ms, seen = (m[k] for k in ("elapsed", "table", "table_steps", "kernels", "dom", "coll", "coll_ms", "seen"))

    def side_line(name, sh_sharded, factored, sc=None, vpr=None):
        """one trial step of another layout / configuration; if the collective library refuses one of its collectives on this node the side line says so instead of
        failing the run; then the same measurement as the headline (no kernel table)"""
        ok, why, tr2 = 1, None, None
        try:
            tr2 = make_trainer(sh_sharded, factored, sc, vpr)
            tr2.iteration = args.start_iteration
            tr2.train_step(targets if sc is None else [scenes.target_image(sc.height, sc.width, seed=43).to(device)])
            torch.cuda.synchronize()
        except RuntimeError as e:
            ok, why = 0, str(e).splitlines()[0][:200]
        flag = torch.tensor([float(ok)], device=device)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        v = vpr or args.views_per_rank
        if float(flag) != 1.0:
            return {"parallelism": name, "value": None, "failed": why or "another rank failed its trial step"}
        tr2.iteration = args.start_iteration
        tg = targets if sc is None else [scenes.target_image(sc.height, sc.width, seed=43).to(device)]
        m2 = measure(tr2, tg, args, world, device, False)
        line = {"parallelism": name, "value": round(world * v * args.steps / m2["elapsed"], 3), "ms_per_step": round(m2["elapsed"] / args.steps * 1e3, 4),
                "views_per_rank": v, "collectives_per_step": {k: {"calls": c["calls"] / args.steps, "MB": round(c["bytes"] / args.steps / 1e6, 3)} for k, c in m2["coll"].items()}}
        if sc is not None:
            line["workload"] = f"{sc.name}: {sc.N} Gaussians, {sc.width}x{sc.height}, {world * v} views per step"
        del tr2
        return line

    sharded_line = factored_line = config4_line = None
    multi_default = world > 1 and not args.replicated and not args.sh_sharded and not args.factored and args.rasterizer == "gut" and args.strategy == "none" and args.path == "step"
    if multi_default:
        # the OTHER exchange of the replicated layout, then the SH-sharded layout
        factored_line = side_line(f"dp{world}-replicated-flat-all-reduce", False, False) if headline_factored else side_line(f"dp{world}-replicated-factored-sh", False, True)
        sharded_line = side_line(f"dp{world}-sh-sharded", True, False)
        if not args.no_config4 and args.workload == "syn-b":
            # BASELINE.json configs[3]: replicated 3 M Gaussians, 1600x1200, 8 views per rank and step (64 at 8 GPUs) - the configuration the ">= 6x at 8 GPUs"
            # target is stated for: 8 views of compute per collective instead of 1 (64 views: the flat all-reduce, see above)
            try:
                sc4 = scenes.syn_c()
                f4 = auto_factored(8)
                config4_line = side_line(f"dp{world}-replicated-" + ("factored-sh" if f4 else "flat-all-reduce"), False, f4, sc=sc4, vpr=8)
                del sc4
            except Exception as e:   # (memory on a small box, ...): reported, not fatal
                config4_line = {"parallelism": f"dp{world}-replicated", "value": None, "failed": str(e).splitlines()[0][:200]}

    # The drop-in route next to the headline (round-5 review, "What's missing" 3): the SAME workload through the call sequence a reference build makes when only
    # gsplat_backend / fastgs_backend are swapped for this library - rasterizer.cpp:224-344 op by op under torch autograd through the compiled reference-signature
    # wrappers (csrc/torch_ops.cpp), then six adam_step_wrapper launches (fused_adam.cpp:22-95) - 10 steps after 3 warm-up steps, in this process, AFTER the timed
    # region of the headline. backend_kernel_ms = this library's kernels (event-bracketed scopes of three further steps), libtorch_glue_ms = the rest of the step.
    ops_route = None
    if (world == 1 and args.path == "step" and args.rasterizer == "gut" and args.strategy == "none" and not args.bilateral_grid and args.loss == "mse"
            and args.views_per_rank == 1 and not args.no_ops_route):
        try:
            from lichtfeld_studio_amd import torch_ops_route
            torch_ops_route.install()
            tr_ops = GutTrainer(scene, device, iterations=7000, world=1, rank=0, views_per_rank=1, loss=args.loss, rasterizer="gut", fused_l2=False, fused_adam=False)
            tr_ops.iteration = args.start_iteration
            for _ in range(3):
                tr_ops.train_step(targets)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                tr_ops.train_step(targets)
            torch.cuda.synchronize()
            ops_ms = (time.perf_counter() - t1) / 10 * 1e3
            capi.profile_collect(); capi.profile_filter(None); capi.profile_enable(True)
            for _ in range(3):
                tr_ops.train_step(targets)
            capi.profile_enable(False)
            tab = capi.profile_collect()
            backend_ms = sum(ms for ms, _cnt in tab.values()) / 3
            ops_route = {"ms_per_step": round(ops_ms, 4), "value": round(1e3 / ops_ms, 3), "steps": 10, "warmup": 3,
                         "backend_kernel_ms": round(backend_ms, 4), "libtorch_glue_ms": round(max(ops_ms - backend_ms, 0.0), 4),
                         "backend_kernels": {k: round(ms / 3, 4) for k, (ms, _cnt) in sorted(tab.items(), key=lambda kv: -kv[1][0])},
                         "what": "rasterizer.cpp:224-344 + fused_adam.cpp:22-95 through the reference-signature wrappers of csrc/torch_ops.cpp (Python autograd glue in place of the reference's C++ autograd: "
                                 "tests/test_gpu_reference_links.py times the reference's own C++ on the same library)"}
            del tr_ops
        except Exception as e:   # reported, never fatal for the headline
            ops_route = {"ms_per_step": None, "failed": str(e).splitlines()[0][:200]}
        finally:
            try:
                torch_ops_route.uninstall()
            except Exception:
                pass

    refine_ms = None
    if args.strategy == "mcmc" and world == 1:   # one refinement step on its own (relocation of the dead Gaussians + the step around it)
        trainer.iteration = (trainer.iteration // 100 + 1) * 100 - 1
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        trainer.train_step(targets)
        torch.cuda.synchronize()
        refine_ms = (time.perf_counter() - t1) * 1e3
    if rank != 0:
        return
    images = world * args.views_per_rank * args.steps
    N = scene.N
    K = (scene.sh_degree + 1) ** 2
    V = int(trainer.last_visible.sum().item()) if trainer.last_visible is not None else 0
    v_source = None
    if V == 0 and args.rasterizer == "fastgs":
        # the fastgs wrappers do not hand back a visibility mask; the byte table needs V: the 3DGUT projection of the same view (the EWA culling of
        # fastgs differs from it by a few rows per million)
        with torch.no_grad():
            m, cam = trainer.model, trainer.camera(0)
            radii = _fused.activations_project(m.means.detach(), m.raw_quats.detach(), m.raw_scales.detach(), m.raw_opacities.detach(), cam.world_view_transform, cam.K,
                                               scene.width, scene.height, None)[3]
        V, v_source = int((radii[0] > 0).all(-1).sum().item()), "3DGUT projection of the same view"
    I = int(trainer.last_n_isects)
    P = scene.width * scene.height
    T = ((scene.width + 15) // 16) * ((scene.height + 15) // 16)
    adam_elems = sum(p.numel() for p in trainer.model.parameters())
    if trainer.iteration <= 1000 or "sh_bwd_adam" in table:
        # shN is not in the optimizer launch: skipped while iteration <= 1000 (fused_adam.cpp:68-70), updated inside sh_bwd_adam afterwards
        adam_elems -= trainer.model.shN.numel()
    bytes_per = algorithmic_bytes(N, V, I, P, T, K, adam_elems)

    roofline = None
    per_kernel = {}
    if table:
        for name, (ms, cnt) in sorted(table.items(), key=lambda kv: -kv[1][0]):
            avg_ms = ms / max(cnt, 1)
            b = bytes_per.get(name)
            per_kernel[name] = {"avg_ms": round(avg_ms, 4), "launches_per_step": cnt // max(table_steps, 1),
                                "alg_GBps": round(b / (avg_ms * 1e-3) / 1e9, 1) if b else None}
    if kernels and dom in kernels:
        avg_s = kernels[dom][0] / max(kernels[dom][1], 1) * 1e-3   # live, inside the timed region
        achieved = bytes_per.get(dom, 0) / avg_s / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                    "alg_bytes_per_launch": int(bytes_per.get(dom, 0)), "avg_launch_ms": round(avg_s * 1e3, 4),
                    "launches_timed": kernels[dom][1]}
        # PMC-measured HBM traffic (separate rocprofv3 --pmc passes, see profiles/): filled when a
        # measurement for this exact workload has been committed.
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        lib_version = capi.load_library().lfs_version().decode()
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                measured_on = tj.get(f"{args.workload}:{N}", {}).get("_library")
                # a later build whose gfx950 code objects are byte for byte those of the measured one (a source edit that is compiled out by default changes the source
                # hash, not the machine code: tools/code_object_identity.py, its output committed under profiles/) is the measured library as far as counters go
                same_code = tj.get(f"{args.workload}:{N}", {}).get("_byte_identical_builds", {})
                if measured_on != lib_version and lib_version in same_code:
                    roofline["traffic_note"] = f"measured on '{measured_on}'; this build's code objects are byte-identical to it ({same_code[lib_version]})"
                elif measured_on != lib_version:
                    # counters of another build say nothing about this one: no number rather than a stale one
                    roofline["traffic_note"] = f"profiles/traffic.json was measured on '{measured_on}', this run is '{lib_version}': traffic withheld (tools/profile.sh + tools/update_traffic.py refresh it)"
                    tj = {}
                ent = tj.get(f"{args.workload}:{N}", {}).get(dom)
                if ent is not None:
                    roofline["traffic"] = ent
                    roofline["traffic_measured_on"] = measured_on
                # the rasterizer kernels are VALU-issue bound, not HBM bound: next to the HBM fraction, the share of the launch during which the
                # vector ALUs were issuing, from the PMC pass (SQ_INSTS_VALU wave-instructions x 4 cycles on a SIMD16, 1024 SIMDs, 2.4 GHz)
                valu = tj.get(f"{args.workload}:{N}", {}).get("_valu_insts", {}).get(dom)
                if valu is not None:
                    # the clock the rasterizer kernels actually run at: 2.0 - 2.2 GHz (profiles/r03/clock_pass_grbm.txt: GRBM_GUI_ACTIVE / SQ_BUSY_CYCLES per launch against the
                    # launch durations), not the 2.4 GHz peak - priced at the 2.4 GHz of rounds 1 - 3 this line understated how close to the issue limit the kernel is
                    ghz = 2.1e9
                    roofline["valu"] = {"wave_insts_per_launch": valu, "issue_ms": round(valu * 4 / 1024 / ghz * 1e3, 4),
                                        "issue_frac_of_launch": round(valu * 4 / 1024 / ghz / avg_s, 4),
                                        "source": "rocprofv3 --pmc SQ_INSTS_VALU (profiles/), 4 cycles per wave64 instruction, 1024 SIMDs, 2.1 GHz (measured under this kernel: profiles/r03/clock_pass_grbm.txt)"}
            except Exception:
                pass

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        threads = args.cpu_threads or (os.cpu_count() or 1)
        try:
            cpu = cpu_baseline(scene, 0, scenes.target_image(scene.height, scene.width, seed=43), threads, hip_step)
        except Exception as e:  # the oracle is optional at bench time; say so instead of failing the run
            cpu = {"value": None, "unit": "train-images/sec", "cores": threads, "kind": "port", "sample": f"failed: {e}"}
        try:  # and the reference's own CPU code at the size it can run (north star: "the reference's CPU torch_impl path timed on the same box")
            cpu["reference_torch_impl"] = reference_torch_impl_baseline(threads)
        except Exception as e:
            cpu["reference_torch_impl"] = {"value": None, "sample": f"failed: {e}"}

    out = {
        "metric": f"train-images/sec (fwd+bwd+Adam), {N / 1e6:g}M Gaussians @{scene.width}x{scene.height}",
        "value": round(images / elapsed, 3), "unit": "train-images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"rasterizer": args.rasterizer, "workload": f"{scene.name}: {N} Gaussians, {scene.width}x{scene.height}, SH degree {scene.sh_degree}, "
                               f"16x16 tiles, {n_views} orbit cameras, {'MSE' if args.loss == 'mse' else 'L1 + 0.2 D-SSIM'} loss, default_optimization_params lrs",
                   "global_batch": world * args.views_per_rank, "views_per_rank": args.views_per_rank,
                   "parallelism": f"dp{world}" + ("-sh-sharded" if trainer.sh_exchange is not None else ("-replicated-factored-sh" if trainer.factored_sh and (world > 1 or os.environ.get("LFS_DIST_FORCE_COLLECTIVES")) else ("-replicated-flat-all-reduce" if world > 1 else ""))),
                   "path": args.path, "step_form": (None if args.path != "step" or trainer.last_plan is None else trainer.last_plan.path + (("_pipelined" if trainer.pipelined else "_fused_tail" if trainer.fused_tail else "") if trainer.last_plan.path == "cxx_all" else "")),
                   "start_iteration": args.start_iteration,
                   "strategy": args.strategy, "bilateral_grid": bool(args.bilateral_grid), "refine_step_ms": None if refine_ms is None else round(refine_ms, 3),
                   "visible_gaussians": V, **({"visible_gaussians_source": v_source} if v_source else {}), "n_isects": I},
        "collectives": {"backend": (torch.distributed.get_backend() if torch.distributed.is_initialized() else None), "ranks_seen": seen,
                        "per_step": {k: {"calls": v["calls"] / args.steps, "MB": round(v["bytes"] / args.steps / 1e6, 3), "ms": round(coll_ms.get(k, 0.0), 4)} for k, v in coll.items()}},
        # what a multi-GPU reader must know about this repository (round-4 review): the collectives below ran through the backend named there, but no development box of
        # rounds 1 - 5 had two GPUs - before the run that produced THIS line no collective of any layout had moved a byte between two devices; every scaling figure in
        # DESIGN.md 7 is arithmetic (estimate: true), and which exchange is the headline (factored up to 16 views per step, flat beyond) was chosen from that arithmetic
        "multi_gpu": {"layout_choice": {"rule": "factored SH exchange while world x views_per_rank <= 16, flat all-reduce beyond", "basis": "arithmetic, not a measurement", "estimate": True},
                      "inter_gpu_collectives_before_this_run": "none: rounds 1 - 5 were developed on single-GPU boxes (gloo ranks on CPU / sharing one GPU, RCCL at world size 1)",
                      "expected_speedup_at_8_gpus": {"factored_1_view_per_rank": 5.2, "factored_chunked_gather": 5.9, "sh_sharded_1_view_per_rank": 6.0, "flat_1_view_per_rank": 4.1, "configs3_8_views_per_rank": 7.0, "estimate": True,
                                                     "source": "DESIGN.md 7 (per-rank compute measured at world size 1 + assumed xGMI bus bandwidth)"}},
        "roofline": roofline, "cpu_baseline": cpu, "kernels": per_kernel,
        "kernels_source": (f"{table_steps} event-bracketed warm-up steps BEFORE the timed region (every kernel scope between two hipEventRecord: the events stretch a step by a few per cent, "
                           "so the rows sum to more than ms_per_step; the dominant kernel's live figure from inside the timed region is roofline.avg_launch_ms)") if per_kernel else None,
        "ops_route": ops_route,
        # hand-written stream kernels on this box class (tools/hbm_stream.hip -> profiles/r06/hbm_stream_ceiling.json): what an HBM-bound kernel is priced against besides the 8 TB/s spec
        "hbm_practical_ceiling_TBps": {"read": 7.2, "write": 5.6, "copy": 5.8, "triad": 5.9, "adam_7_streams": 5.3, "rmw_32_streams": 5.1, "source": "profiles/r06/hbm_stream_ceiling.json"},
        "library": capi.load_library().lfs_version().decode(),
        **({"sh_sharded": sharded_line} if sharded_line is not None else {}),
        **({"replicated_other_exchange": factored_line} if factored_line is not None else {}),
        **({"config4": config4_line} if config4_line is not None else {}),
    }
    # C-level stdout first (RCCL prints its version banner through stdio; on a pipe that buffer would otherwise be flushed at exit, after our line):
    # the JSON line is the last thing rank 0 prints
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    try:
        main()
    finally:
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
