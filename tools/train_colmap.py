#!/usr/bin/env python
"""Train on a COLMAP directory end to end on one MI355X (the reference's `LichtFeld-Studio -d <data> --images images_4 --test-every 8
--eval -i 30000`, eval/benchmark_mipnerf360.sh:37-44, reduced to the data path this repository covers):

    python tools/train_colmap.py -d /data/garden --images images_4 --iterations 30000 --strategy default --eval -o out/garden

COLMAP reader -> point-cloud initialisation -> fastgs (default) or 3DGUT training with the L1 + SSIM loss and the ADC / MCMC strategy ->
PSNR / SSIM on the held-out views -> splat PLY. Prints one JSON line. All views must share one image size.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-d", "--data-path", required=True)
    ap.add_argument("--images", default="images")
    ap.add_argument("-i", "--iterations", type=int, default=30000)
    ap.add_argument("--strategy", default="default", choices=["default", "mcmc", "none"])
    ap.add_argument("--gut", action="store_true", help="3DGUT rasterizer instead of the default EWA (fastgs) one")
    ap.add_argument("--test-every", type=int, default=8)
    ap.add_argument("--resize-factor", type=int, default=-1)
    ap.add_argument("--max-width", type=int, default=3840)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--bilateral-grid", action="store_true")
    ap.add_argument("--eval", action="store_true")
    ap.add_argument("-o", "--output-path", default="output")
    ap.add_argument("--text", action="store_true", help="read cameras.txt / images.txt / points3D.txt")
    ap.add_argument("--eval-every", type=int, default=0, help="with --eval: held-out PSNR (the renderer the model trains with) every N iterations -> 'psnr_curve' (the turbulence of an MCMC run)")
    args = ap.parse_args()

    import lichtfeld_studio_amd  # noqa: F401
    from lichtfeld_studio_amd import evaluate, loader, strategies
    from lichtfeld_studio_amd.rasterizer import Camera
    from lichtfeld_studio_amd.trainer import GutTrainer

    dev = torch.device("cuda:0")
    mcmc = args.strategy == "mcmc"
    op = strategies.OptimizationParameters.for_strategy(args.strategy, iterations=args.iterations)   # ADC: stop_refine 15000, no regularisers
    init_scaling, init_opacity = (0.1, 0.5) if mcmc else (1.0, 0.1)        # parameter/{mcmc,default}_optimization_params.json
    t0 = time.time()
    split = "train" if args.eval else "all"
    scene, ds, scene_scale = loader.colmap_scene(args.data_path, args.images, split=split, test_every=args.test_every, resize_factor=args.resize_factor,
                                                 max_width=args.max_width, sh_degree=args.sh_degree, init_scaling=init_scaling, init_opacity=init_opacity,
                                                 text=args.text, device=dev)
    targets = loader.preload(ds, dev, workers=os.cpu_count() or 8)          # resident in HBM: 288 GB hold a Mip-NeRF360 scene at full resolution
    t_load = time.time() - t0
    rast = "gut" if args.gut else "fastgs"
    if args.bilateral_grid and rast != "fastgs":
        raise SystemExit("--bilateral-grid needs the fastgs rasterizer")
    tr = GutTrainer(scene, dev, iterations=args.iterations, loss="l1_ssim", strategy=None if args.strategy == "none" else args.strategy, opt_params=op,
                    scene_scale=scene_scale, rasterizer=rast, use_bilateral_grid=args.bilateral_grid)
    g = torch.Generator().manual_seed(0)
    val_set = None
    if args.eval and args.eval_every > 0:
        cams_all, _ = (loader.read_colmap_cameras_and_images_text if args.text else loader.read_colmap_cameras_and_images)(args.data_path, args.images)
        val = loader.CameraDataset(cams_all, "val", args.test_every, args.resize_factor, args.max_width)
        vc, vi = [], []
        for k, img in enumerate(loader.preload(val, dev)):
            cam = val.cameras[val.indices[k]]
            h, w = img.shape[1:]
            vc.append(Camera(torch.from_numpy(loader.world_to_view(cam))[None].to(dev), torch.from_numpy(loader.intrinsics(cam, w, h))[None].to(dev), w, h))
            vi.append(img)
        val_set = (vc, vi)
    curve = []
    t0 = time.time()
    t_eval = 0.0
    order = []
    for it in range(args.iterations):
        if not order:
            order = torch.randperm(len(ds), generator=g).tolist()          # infinite random sampler, one view per step
        v = order.pop()
        tr.train_step([targets[v]], views=[v])
        if val_set is not None and (it + 1) % args.eval_every == 0:
            torch.cuda.synchronize(); te = time.time()
            m = evaluate.evaluate(tr.model, val_set[0], val_set[1], it + 1, rasterizer=rast if args.gut else "fastgs")
            # the same metric on 12 TRAINING views: a held-out PSNR that falls while this one holds is a generalisation gap (Gaussians fitted to the rays the training views
            # sample), both falling is an unstable optimisation
            tsel = list(range(0, len(ds), max(1, len(ds) // 12)))[:12]
            tcams = [tr.camera(v) for v in tsel]
            mt = evaluate.evaluate(tr.model, tcams, [targets[v] for v in tsel], it + 1, rasterizer=rast if args.gut else "fastgs")
            s3 = tr.model.raw_scales.detach()
            asp = (s3.max(-1).values - s3.min(-1).values).exp()
            # projected size of the SMALLEST axis of every Gaussian in pixels at the first training camera (focal length x scale / depth): the share below half a pixel
            cam0 = tr.camera(0)
            with torch.no_grad():
                pc = tr.model.means.detach() @ cam0.world_view_transform[0, :3, :3].T + cam0.world_view_transform[0, :3, 3]
                px = float(cam0.K[0, 0, 0]) * s3.exp().min(-1).values / pc[:, 2].clamp_min(1e-3)
                sub = float(((px < 0.5) & (pc[:, 2] > 0.01)).float().mean())
            curve.append({"iteration": it + 1, "psnr": round(m.psnr, 3), "psnr_train_views": round(mt.psnr, 3), "thin_axis_below_half_pixel": round(sub, 4),
                          "gaussians": int(tr.model.means.shape[0]), "mean_log_scale": round(float(s3.mean()), 3),
                          "median_aspect": round(float(asp.median()), 2), "frac_aspect_ge_10": round(float((asp >= 10).float().mean()), 4)})
            print(json.dumps(curve[-1]), file=sys.stderr, flush=True)
            t_eval += time.time() - te
    torch.cuda.synchronize()
    t_train = time.time() - t0 - t_eval
    out = {"data": args.data_path, "images": len(ds), "size": [scene.width, scene.height], "iterations": args.iterations, "rasterizer": rast,
           "strategy": args.strategy, "gaussians": int(tr.model.means.shape[0]), "load_s": round(t_load, 1), "train_s": round(t_train, 1),
           "iters_per_s": round(args.iterations / max(t_train, 1e-9), 1)}
    if args.eval:
        cams_all, _ = (loader.read_colmap_cameras_and_images_text if args.text else loader.read_colmap_cameras_and_images)(args.data_path, args.images)
        val = loader.CameraDataset(cams_all, "val", args.test_every, args.resize_factor, args.max_width)
        cameras, images = [], []
        for k, img in enumerate(loader.preload(val, dev)):
            cam = val.cameras[val.indices[k]]
            h, w = img.shape[1:]
            cameras.append(Camera(torch.from_numpy(loader.world_to_view(cam))[None].to(dev), torch.from_numpy(loader.intrinsics(cam, w, h))[None].to(dev), w, h))
            images.append(img)
        m = evaluate.evaluate(tr.model, cameras, images, args.iterations)
        out.update(psnr=round(m.psnr, 4), ssim=round(m.ssim, 5), val_images=m.n_images)
        if args.gut:   # the reference's protocol above renders with the EWA rasterizer; this is the renderer the model was trained with
            mg = evaluate.evaluate(tr.model, cameras, images, args.iterations, rasterizer="gut")
            out.update(psnr_gut=round(mg.psnr, 4), ssim_gut=round(mg.ssim, 5))
    if curve:
        out["psnr_curve"] = curve
    os.makedirs(args.output_path, exist_ok=True)
    ply = os.path.join(args.output_path, f"splat_{args.iterations}.ply")
    loader.save_ply(tr.model, ply)
    out["ply"] = ply
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
