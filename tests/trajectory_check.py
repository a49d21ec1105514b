"""How far apart are the HIP training trajectory and the oracle's, step by step?  (GPU box; the oracle runs on its host cores.)

The PSNR criterion (tests/convergence_l1ssim.py) compares END POINTS of two chaotic trajectories: single seeds scatter by +-0.3 dB on both sides, and a
mean over seeds resolves a systematic difference only down to a few hundredths of a dB. This script looks for a systematic difference where it would have to
start: both sides train the SAME task from the SAME start on the SAME targets (the HIP renders), the HIP side in the deterministic accumulation mode, and the
parameters are compared at checkpoints - across the two schedule boundaries of the path (shN enters Adam after iteration 1000, fused_adam.cpp:68-70; the
single-rank step becomes ONE C++ call there, trainer.plan_step) and through the exponential decay of the means' learning rate. A skipped or doubled update, a
bias correction or a learning rate off by one step shows as a jump of the distance at that step; rounding differences grow smoothly.

    python tests/trajectory_check.py --steps 1100 --out gpurun_out/trajectory_check.json
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
NAMES = ["means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opacities"]


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--total-iters", type=int, default=7000, help="the schedule both sides are configured for (learning-rate decay)")
    ap.add_argument("--checkpoints", type=int, nargs="*", default=[1, 2, 5, 10, 50, 100, 300, 600, 999, 1000, 1001, 1002, 1005, 1020, 1100])
    ap.add_argument("--start-iteration", type=int, default=0, help="both sides begin at this iteration with a fresh optimizer (1000: the FIRST step already is the one-call form "
                                                                    "with shN in Adam - the boundary compared from identical states)")
    ap.add_argument("--out", default="")
    # round 5: the task itself, towards BASELINE's size (defaults = the 6000-Gaussian task of rounds 2 - 4), and a segment that starts from a HIP-trained state
    ap.add_argument("--n", type=int, default=6000)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--sh-degree", type=int, default=1)
    ap.add_argument("--scale", type=float, default=0.07, help="median scale of the ground-truth Gaussians")
    ap.add_argument("--flat", type=float, default=0.0, help="> 1: flat disks, aspect log-uniform in [1, FLAT]")
    ap.add_argument("--pretrain", type=int, default=0, help="the HIP side trains this many iterations first (float atomics, the benchmarked step); the compared segment then starts from THAT "
                                                           "state on both sides, at iteration = pretrain, with a fresh optimizer")
    args = ap.parse_args()
    import lichtfeld_studio_amd
    import oracle
    from convergence_check import make_task, render_views_hip
    from lichtfeld_studio_amd.trainer import GutTrainer
    from oracle import pipeline
    dev = torch.device("cuda:0")
    lib = lichtfeld_studio_amd.load_library()
    cps = sorted(c for c in set(args.checkpoints) if c <= args.steps)
    kw = dict(n=args.n, n_views=args.views, sh_degree=args.sh_degree, scale=args.scale, flat_max_aspect=args.flat if args.flat > 1 else None)
    if args.width and args.height:
        kw.update(width=args.width, height=args.height)
    gt, init = make_task(seed=100 + args.seed, **kw)
    targets = render_views_hip(gt, dev)
    nV = init.viewmats.shape[0]
    from convergence_check import psnr as _psnr
    from lichtfeld_studio_amd import scenes as _scenes

    def _scene_of(P):
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a))
        return _scenes.Scene("s", init.width, init.height, init.sh_degree, t(P["means"]), t(P["raw_quats"]), t(P["raw_scales"]), t(P["raw_opacities"]), t(P["sh0"]), t(P["shN"]),
                             init.viewmats, init.Ks)

    def _eval(P):
        return float(np.mean([_psnr(a.cpu().numpy(), b.cpu().numpy()) for a, b in zip(render_views_hip(_scene_of(P), dev), targets)]))

    pre = {}
    if args.pretrain > 0:
        t0 = time.time()
        trp = GutTrainer(init, dev, iterations=args.total_iters, loss="mse")
        for it in range(args.pretrain):
            trp.train_step([targets[it % nV]], views=[it % nV])
        torch.cuda.synchronize()
        Pp = {k: getattr(trp.model, k).detach().cpu().numpy().copy() for k in NAMES}
        s3 = Pp["raw_scales"]
        pre = {"pretrain_iterations": args.pretrain, "pretrain_seconds": round(time.time() - t0, 1), "psnr_start": round(_eval({k: np.asarray(pipeline.scene_arrays(init)[k]) for k in NAMES}), 3),
               "psnr_after_pretrain": round(_eval(Pp), 3), "median_aspect_after_pretrain": round(float(np.median(np.exp(s3.max(-1) - s3.min(-1)))), 2)}
        print(json.dumps(pre), flush=True)
        init = _scene_of(Pp)
        args.start_iteration = args.pretrain
        del trp

    # ---- HIP, deterministic accumulation, the trainer exactly as tests/convergence_l1ssim.py --loss mse runs it
    hip, paths = {}, {}
    lib.lfs_set_debug_flags(16)
    try:
        tr = GutTrainer(init, dev, iterations=args.total_iters, loss="mse")
        tr.iteration = args.start_iteration
        for it in range(args.steps):
            tr.train_step([targets[it % nV]], views=[it % nV])
            if it + 1 in cps:
                hip[it + 1] = {k: getattr(tr.model, k).detach().cpu().numpy().copy() for k in NAMES}
                paths[it + 1] = tr.last_plan.path
    finally:
        lib.lfs_set_debug_flags(0)

    # ---- oracle (CPU): tests/convergence_l1ssim.py::train_oracle with checkpoints, on the HIP-rendered targets
    tnp = [t.cpu().numpy() for t in targets]
    sa = pipeline.scene_arrays(init)
    P = {k: np.array(sa[k], copy=True) for k in NAMES}
    M = {k: np.zeros_like(v) for k, v in P.items()}
    V = {k: np.zeros_like(v) for k, v in P.items()}
    lrs = {"means": 1.6e-4, "sh0": 2.5e-3, "shN": 2.5e-3 / 20, "raw_scales": 5e-3, "raw_quats": 1e-3, "raw_opacities": 5e-2}
    gamma = 0.01 ** (1.0 / args.total_iters)
    rows = []
    t0 = time.time()
    for it in range(args.steps):
        g = pipeline.train_image(dict(sa, **P), it % nV, tnp[it % nV], loss_fn=None)["grads"]
        step = it + 1
        for k in NAMES:
            if k == "shN" and args.start_iteration + step <= 1000:   # (the optimizer's step count starts at 1 on both sides; the shortcut looks at the ITERATION)
                continue
            p, m, v = oracle.adam_step(P[k].reshape(-1), M[k].reshape(-1), V[k].reshape(-1), np.ascontiguousarray(g[k], np.float32).reshape(-1), lrs[k], 0.9, 0.999,
                                       1e-15, 1.0 / (1.0 - 0.9 ** step), 1.0 / math.sqrt(1.0 - 0.999 ** step))
            P[k], M[k], V[k] = p.reshape(P[k].shape), m.reshape(P[k].shape), v.reshape(P[k].shape)
        lrs["means"] *= gamma
        if step in cps:
            start = {k: np.asarray(sa[k]) for k in NAMES}
            row = {"step": step, "hip_step_form": paths.get(step)}
            for k in NAMES:
                # distance between the two trajectories, relative to how far the parameter has MOVED from the start (a distance relative to the
                # parameter's own norm would hide everything behind the unchanged bulk)
                moved = float(np.linalg.norm(np.asarray(P[k], np.float64) - start[k]))
                row[k] = float(np.linalg.norm(np.asarray(hip[step][k], np.float64) - P[k]) / max(moved, 1e-30)) if moved > 0 else 0.0
            row["max"] = max(row[k] for k in NAMES)
            # Adam's first updates are sign-like (m / sqrt(v) = +-1 whatever |g|): an element whose gradient is rounding noise moves by a full step in a random
            # direction on either side. Fraction of elements further apart than half the median distance moved:
            row["apart_by_half_a_step"] = {}
            for k in NAMES:
                d = np.abs(np.asarray(P[k], np.float64) - start[k]).reshape(-1)
                if (d > 0).any():
                    row["apart_by_half_a_step"][k] = float((np.abs(np.asarray(hip[step][k], np.float64) - P[k]).reshape(-1) > 0.5 * np.median(d[d > 0])).mean())
            rows.append(row)
            print(json.dumps(row), flush=True)
    fin = {"psnr_hip_after_segment": round(_eval(hip[cps[-1]]), 4), "psnr_oracle_after_segment": round(_eval(P), 4)} if cps and cps[-1] == args.steps else {}
    print(json.dumps(fin), flush=True)
    res = {"start_iteration": args.start_iteration, **pre, **fin, "task": f"seed {args.seed}: {args.n} Gaussians{f' (flat disks, aspect up to {args.flat:g})' if args.flat > 1 else ''}, {nV} views {init.width}x{init.height}, SH degree {init.sh_degree}, clamped MSE; distance HIP (deterministic mode) - oracle / distance moved from the start, per tensor",
           "library": lib.lfs_version().decode(), "oracle_seconds": round(time.time() - t0, 1), "rows": rows}
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
