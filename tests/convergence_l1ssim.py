"""PSNR after 7000 iterations, HIP path vs the CPU oracle (the restatement of the reference's kernels, pinned to them by tests/golden/refk_*),
with the REFERENCE'S loss (L1 + 0.2 D-SSIM, trainer.cpp:115-128), over several seeds - BASELINE.json north star: "PSNR within 0.05 dB of
reference after 7k iters". Task per seed: recover a ground-truth set of Gaussians from its own renders, starting from a perturbed copy
(tests/convergence_check.py::make_task; 6000 Gaussians, 8 views 192x192, SH degree 1: the scale of BASELINE config 1).

  python tests/convergence_l1ssim.py --oracle --seeds 0 1 2 3 4 --out profiles/r02/convergence_l1ssim_oracle.json     # CPU only (build container)
  python tests/convergence_l1ssim.py --hip    --seeds 0 1 2 3 4 --oracle-json profiles/r02/convergence_l1ssim_oracle.json   # GPU box

The oracle side needs no GPU (~30 min per seed on 8 cores) and stores the final parameters next to the JSON (.npz); the HIP side trains in the
default (float-atomic) mode and in the deterministic mode (lfs_set_debug_flags(16): bit-identical between runs) and evaluates EVERY final
model - its own and the oracle's - with the same HIP renderer, so the PSNR difference is a difference of the trained parameters only."""
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
ITERS = 7000


def psnr(a, b):
    return 10.0 * math.log10(1.0 / max(float(((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2).mean()), 1e-20))


def oracle_targets(gt):
    """the ground-truth views rendered by the oracle (CPU); the HIP side renders them with the HIP kernels (mean difference ~1e-7)"""
    from oracle import pipeline
    sa = pipeline.scene_arrays(gt)
    z = np.zeros((3, gt.height, gt.width), np.float32)
    return [np.clip(pipeline.train_image(sa, v, z, backward=False)["render"][0].transpose(2, 0, 1), 0, 1).astype(np.float32) for v in range(gt.viewmats.shape[0])]


def l1_ssim(raw, target):
    from ssim_reference import photometric_loss
    x = torch.from_numpy(raw).requires_grad_(True)
    loss = photometric_loss(x.clamp(0, 1)[None], torch.from_numpy(target)[None], 0.2)
    loss.backward()
    return float(loss), x.grad.numpy()


def train_oracle(init, targets, iters=ITERS, loss="l1_ssim"):
    import oracle
    from oracle import pipeline
    sa = pipeline.scene_arrays(init)
    P = {k: np.array(sa[k], copy=True) for k in NAMES}
    M = {k: np.zeros_like(v) for k, v in P.items()}
    V = {k: np.zeros_like(v) for k, v in P.items()}
    lrs = {"means": 1.6e-4, "sh0": 2.5e-3, "shN": 2.5e-3 / 20, "raw_scales": 5e-3, "raw_quats": 1e-3, "raw_opacities": 5e-2}
    gamma = 0.01 ** (1.0 / iters)
    nV = init.viewmats.shape[0]
    for it in range(iters):
        cur = dict(sa, **P)
        g = pipeline.train_image(cur, it % nV, targets[it % nV], loss_fn=l1_ssim if loss == "l1_ssim" else None)["grads"]   # (None: the clamped MSE of SURVEY.md 8d)
        step = it + 1
        for k in NAMES:
            if k == "shN" and step <= 1000:   # fused_adam.cpp:68-70
                continue
            p, m, v = oracle.adam_step(P[k].reshape(-1), M[k].reshape(-1), V[k].reshape(-1), np.ascontiguousarray(g[k], np.float32).reshape(-1), lrs[k], 0.9, 0.999,
                                       1e-15, 1.0 / (1.0 - 0.9 ** step), 1.0 / math.sqrt(1.0 - 0.999 ** step))
            P[k], M[k], V[k] = p.reshape(P[k].shape), m.reshape(P[k].shape), v.reshape(P[k].shape)
        lrs["means"] *= gamma
    return P


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--hip", action="store_true")
    ap.add_argument("--seeds", type=int, nargs="+", default=[0, 1, 2, 3, 4])
    ap.add_argument("--iters", type=int, default=ITERS)
    ap.add_argument("--loss", choices=["l1_ssim", "mse"], default="l1_ssim", help="mse: the loss of the BENCHMARKED step (SURVEY.md 8d), HIP side through the C++ step driver")
    ap.add_argument("--flat", type=float, default=0.0, help="> 1: the ground truth is made of flat disks, aspect ratio log-uniform in [1, FLAT] (convergence_check.make_task)")
    # round 5: the task's size, towards BASELINE (defaults: the 6000-Gaussian task every stored result of rounds 2 - 4 refers to)
    ap.add_argument("--n", type=int, default=6000)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--sh-degree", type=int, default=1)
    ap.add_argument("--scale", type=float, default=0.07, help="median scale of the ground-truth Gaussians")
    ap.add_argument("--atomic-runs", type=int, default=3)
    ap.add_argument("--det-runs", type=int, default=2, choices=[1, 2], help="2: the deterministic mode is run twice and the two results compared bit for bit")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02", "convergence_l1ssim_oracle.json"))
    ap.add_argument("--oracle-json", default=os.path.join(ROOT, "profiles", "r02", "convergence_l1ssim_oracle.json"))
    args = ap.parse_args()
    import lichtfeld_studio_amd  # noqa: F401
    from convergence_check import make_task as _make_task
    from lichtfeld_studio_amd import scenes

    def make_task(seed, flat_max_aspect=None):
        kw = dict(n=args.n, n_views=args.views, sh_degree=args.sh_degree, scale=args.scale, flat_max_aspect=flat_max_aspect)
        if args.width and args.height:
            kw.update(width=args.width, height=args.height)
        return _make_task(seed=seed, **kw)

    if args.oracle:
        from oracle import pipeline
        lname = "L1 + 0.2 D-SSIM" if args.loss == "l1_ssim" else "clamped MSE"
        res = json.load(open(args.out)) if os.path.exists(args.out) else {"task": f"recover {args.n} Gaussians{f' (flat disks, aspect up to {args.flat:g})' if args.flat > 1 else ''} from {args.views} views {args.width or 192}x{args.height or 192}, SH degree {args.sh_degree}, {lname} loss, {args.iters} iterations", "seeds": {}}
        for seed in args.seeds:
            if str(seed) in res["seeds"]:
                continue
            gt, init = make_task(seed=100 + seed, flat_max_aspect=args.flat if args.flat > 1 else None)
            targets = oracle_targets(gt)
            t0 = time.time()
            P = train_oracle(init, targets, args.iters, args.loss)
            fin = dict(pipeline.scene_arrays(init), **P)
            z = np.zeros((3, gt.height, gt.width), np.float32)
            ps = [psnr(np.clip(pipeline.train_image(fin, v, z, backward=False)["render"][0].transpose(2, 0, 1), 0, 1), targets[v]) for v in range(len(targets))]
            np.savez_compressed(args.out.replace(".json", f"_seed{seed}.npz"), **{k: P[k] for k in NAMES})   # (the HIP side renders its own targets: 1e-7 apart)
            res["seeds"][str(seed)] = {"oracle_psnr_oracle_renderer": round(float(np.mean(ps)), 4), "seconds": round(time.time() - t0, 1)}
            json.dump(res, open(args.out, "w"), indent=1)
            print(seed, res["seeds"][str(seed)], flush=True)
        return

    assert args.hip
    from convergence_check import render_views_hip
    from lichtfeld_studio_amd.trainer import GutTrainer
    dev = torch.device("cuda:0")
    lib = lichtfeld_studio_amd.load_library()
    ores = json.load(open(args.oracle_json)) if os.path.exists(args.oracle_json) else {"seeds": {}}
    out = {"task": ores.get("task", "recover 6000 Gaussians from 8 views 192x192, SH degree 1, L1 + 0.2 D-SSIM loss, 7000 iterations"), "seeds": {}}

    def scene_of(init, P):
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a))
        return scenes.Scene("fin", init.width, init.height, init.sh_degree, t(P["means"]), t(P["raw_quats"]), t(P["raw_scales"]), t(P["raw_opacities"]), t(P["sh0"]), t(P["shN"]),
                            init.viewmats, init.Ks)

    def eval_psnr(sc, targets):
        return float(np.mean([psnr(a.cpu().numpy(), b.cpu().numpy()) for a, b in zip(render_views_hip(sc, dev), targets)]))

    def train(init, targets, det):
        lib.lfs_set_debug_flags(16 if det else 0)
        try:
            tr = GutTrainer(init, dev, iterations=args.iters, loss=args.loss)
            if args.loss == "mse":
                assert tr.cxx_step, "the MSE run is meant to go through the benchmarked C++ step (lfs_gut_train_step)"
            V = init.viewmats.shape[0]
            for it in range(args.iters):
                tr.train_step([targets[it % V]], views=[it % V])
            m = tr.model
            return {k: getattr(m, k).detach().cpu().numpy() for k in NAMES}
        finally:
            lib.lfs_set_debug_flags(0)

    for seed in args.seeds:
        gt, init = make_task(seed=100 + seed, flat_max_aspect=args.flat if args.flat > 1 else None)
        f = args.oracle_json.replace(".json", f"_seed{seed}.npz")
        o = dict(np.load(f)) if os.path.exists(f) else None
        targets = render_views_hip(gt, dev)
        r = {"psnr_start": round(eval_psnr(init, targets), 4)}
        d1 = train(init, targets, True)
        if args.det_runs == 2:
            d2 = train(init, targets, True)
            r["deterministic_runs_bit_identical"] = bool(all(np.array_equal(d1[k], d2[k]) for k in NAMES))
        r["hip_deterministic"] = round(eval_psnr(scene_of(init, d1), targets), 4)
        r["hip_atomic"] = [round(eval_psnr(scene_of(init, train(init, targets, False)), targets), 4) for _ in range(args.atomic_runs)]
        if o is not None:
            r["oracle"] = round(eval_psnr(scene_of(init, o), targets), 4)
            r["oracle_psnr_oracle_renderer"] = ores["seeds"][str(seed)]["oracle_psnr_oracle_renderer"]
            r["gap_deterministic_db"] = round(r["hip_deterministic"] - r["oracle"], 4)
        elif str(seed) in ores.get("seeds", {}):
            # no stored model for this seed: the oracle's final PSNR as its OWN renderer measured it (the two renderers differ by ~1e-7 in the image: where both
            # numbers exist they agree to the fourth decimal - seeds 0 - 4 of profiles/r04/convergence_mse_hip_5seeds.json)
            r["oracle"] = r["oracle_psnr_oracle_renderer"] = ores["seeds"][str(seed)]["oracle_psnr_oracle_renderer"]
            r["gap_deterministic_db"] = round(r["hip_deterministic"] - r["oracle"], 4)
        out["seeds"][str(seed)] = r
        print(seed, r, flush=True)
    hd = [v["hip_deterministic"] for v in out["seeds"].values()]
    ha = [x for v in out["seeds"].values() for x in v["hip_atomic"]]
    out["summary"] = {"hip_deterministic_mean": round(float(np.mean(hd)), 4), "hip_deterministic_std": round(float(np.std(hd)), 4),
                      "hip_atomic_mean": round(float(np.mean(ha)), 4) if ha else None, "hip_atomic_std": round(float(np.std(ha)), 4) if ha else None}
    oo = [v["oracle"] for v in out["seeds"].values() if "oracle" in v]
    if oo:
        gaps = [v["gap_deterministic_db"] for v in out["seeds"].values() if "oracle" in v]
        # the criterion (BASELINE.json: "PSNR within 0.05 dB of reference after 7k iters") on the MEAN over the seeds, with its 95 % interval (Student t): one
        # trajectory is a sample - the HIP runs of ONE seed scatter by sigma 0.02 - 0.16 dB between float-atomic orders, and so would two builds of the reference
        ga = [x - v["oracle"] for v in out["seeds"].values() if "oracle" in v for x in v["hip_atomic"]]
        per_seed_atomic = [float(np.mean(v["hip_atomic"])) - v["oracle"] for v in out["seeds"].values() if "oracle" in v and v["hip_atomic"]]
        from scipy.stats import t as student_t
        ci = lambda xs: round(float(student_t.ppf(0.975, len(xs) - 1) * np.std(xs, ddof=1) / math.sqrt(len(xs))), 4) if len(xs) > 1 else None
        out["summary"].update(oracle_mean=round(float(np.mean(oo)), 4), oracle_std=round(float(np.std(oo)), 4), mean_gap_db=round(float(np.mean(gaps)), 4),
                              mean_abs_gap_db=round(float(np.mean(np.abs(gaps))), 4), n_seeds_with_oracle=len(oo), mean_gap_ci95_db=ci(gaps),
                              mean_gap_atomic_db=round(float(np.mean(per_seed_atomic)), 4) if per_seed_atomic else None, mean_gap_atomic_ci95_db=ci(per_seed_atomic),
                              atomic_runs_within_0p05=int(sum(abs(x) <= 0.05 for x in ga)), atomic_runs=len(ga))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
