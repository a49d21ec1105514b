#!/usr/bin/env python
"""Debug driver: tools/train_colmap.py's loop with PSNR on four held-out views and parameter statistics every --every iterations (which iteration does a run go wrong at?)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-d", "--data-path", required=True)
    ap.add_argument("-i", "--iterations", type=int, default=30000)
    ap.add_argument("--strategy", default="mcmc")
    ap.add_argument("--gut", action="store_true")
    ap.add_argument("--every", type=int, default=2500)
    ap.add_argument("--max-cap", type=int, default=0)
    ap.add_argument("--noise-lr", type=float, default=-1.0)
    ap.add_argument("--det", action="store_true", help="deterministic accumulation mode (debug bit 4)")
    ap.add_argument("--diagnose", action="store_true", help="after the last iteration: tile-list sortedness and lengths, depth histogram, renders without the near Gaussians, HIP vs oracle render")
    args = ap.parse_args()
    import lichtfeld_studio_amd
    from lichtfeld_studio_amd import evaluate, loader, strategies
    from lichtfeld_studio_amd.rasterizer import Camera
    from lichtfeld_studio_amd.trainer import GutTrainer
    dev = torch.device("cuda:0")
    mcmc = args.strategy == "mcmc"
    op = strategies.OptimizationParameters.for_strategy("mcmc" if args.strategy == "none" else args.strategy, iterations=args.iterations)
    if args.det:
        lichtfeld_studio_amd.load_library().lfs_set_debug_flags(16)
    if args.max_cap:
        op.max_cap = args.max_cap
    init_scaling, init_opacity = (0.1, 0.5) if mcmc else (1.0, 0.1)
    scene, ds, scene_scale = loader.colmap_scene(args.data_path, "images", split="train", test_every=8, sh_degree=3, init_scaling=init_scaling, init_opacity=init_opacity, device=dev)
    targets = loader.preload(ds, dev, workers=os.cpu_count() or 8)
    cams_all, _ = loader.read_colmap_cameras_and_images(args.data_path, "images")
    val = loader.CameraDataset(cams_all, "val", 8, -1, 3840)
    cameras, images = [], []
    for k, img in enumerate(loader.preload(val, dev)[:4]):
        cam = val.cameras[val.indices[k]]
        h, w = img.shape[1:]
        cameras.append(Camera(torch.from_numpy(loader.world_to_view(cam))[None].to(dev), torch.from_numpy(loader.intrinsics(cam, w, h))[None].to(dev), w, h))
        images.append(img)
    tr = GutTrainer(scene, dev, iterations=args.iterations, loss="l1_ssim", strategy=None if args.strategy == "none" else args.strategy, opt_params=op, scene_scale=scene_scale, rasterizer="gut" if args.gut else "fastgs")
    if args.noise_lr >= 0 and tr.strategy is not None:
        tr.strategy.noise_lr = args.noise_lr
    g = torch.Generator().manual_seed(0)
    torch.manual_seed(0)
    order = []
    worst = 0.0
    t0 = time.time()
    for it in range(args.iterations):
        if not order:
            order = torch.randperm(len(ds), generator=g).tolist()
        v = order.pop()
        loss = tr.train_step([targets[v]], views=[v])
        if (it + 1) % 50 == 0:
            worst = max(worst, float(loss))   # (a host read every 50 steps: debug driver)
        if (it + 1) % args.every == 0 or it + 1 in (100, 500, 1000):
            m = tr.model
            ev = evaluate.evaluate(m, cameras, images, it + 1)
            from lichtfeld_studio_amd.rasterizer import rasterize
            with torch.no_grad():
                bg0 = torch.zeros(3, device=dev)
                pg = [evaluate.psnr(torch.clamp(rasterize(c, m, bg0).image, 0, 1), im) for c, im in zip(cameras, images)]
                sc_ = m.raw_scales.detach().exp().max(dim=1).values
                op_ = torch.sigmoid(m.raw_opacities.detach()).reshape(-1)
                far = m.means.detach().norm(dim=1)
                big = sc_ > 10.0
            st = {"iter": it + 1, "loss": float(loss) if loss is not None else None, "worst_loss_sampled": round(worst, 4), "psnr_ewa": round(ev.psnr, 3), "psnr_gut": round(float(sum(pg) / len(pg)), 3), "big(scale>10)": int(big.sum()), "big_mean_opacity": round(float(op_[big].mean()), 4) if big.any() else None,
                  "far(|mean|>100)": int((far > 100).sum()), "opacity<0.005": int((op_ < 0.005).sum()), "N": int(m.means.shape[0]),
                  "plan": tr.last_plan.path, "active_sh": m.get_active_sh_degree(), "retries": tr._gut_step.retries if tr._gut_step is not None else None,
                  "n_isects": getattr(tr, "last_n_isects", None), "s": round(time.time() - t0, 1)}
            for name in ("means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opacities"):
                t = getattr(m, name).detach()
                st[name] = [bool(torch.isfinite(t).all()), round(float(t.abs().max()), 3), round(float(t.float().mean()), 4)]
            print(json.dumps(st), flush=True)
            worst = 0.0
    if args.diagnose:
        diagnose(tr, cameras, images, dev)


def diagnose(tr, cameras, images, dev):
    import lichtfeld_studio_amd as lfs
    from lichtfeld_studio_amd import evaluate, ops, scenes
    from lichtfeld_studio_amd.fastgs import fast_rasterize
    from lichtfeld_studio_amd.rasterizer import SplatModel, rasterize
    m = tr.model
    bg0 = torch.zeros(3, device=dev)
    out = {}
    with torch.no_grad():
        r = rasterize(cameras[0], m, bg0)
        radii2 = torch.stack([r.radii, r.radii], -1)[None].int()
        W, H = cameras[0].image_width, cameras[0].image_height
        tw, th = (W + 15) // 16, (H + 15) // 16
        # (the op again, standalone: keys sorted? longest list?)
        from lichtfeld_studio_amd.rasterizer import fully_fused_projection_with_ut, UnscentedTransformParameters
        rad, m2, dep, _, _ = fully_fused_projection_with_ut(m.get_means(), m.get_rotation(), m.get_scaling(), m.get_opacity().reshape(-1), cameras[0].world_view_transform, cameras[0].K,
                                                            None, None, None, W, H, 0.3, 0.01, 10000.0, 0.0, 1.0, cameras[0].camera_model_type, UnscentedTransformParameters())
        tpg, ids, flat, offs = ops.intersect_tile(m2.contiguous(), rad, dep, None, None, 1, 16, tw, th, True, return_offsets=True)
        out["n_isects"] = int(flat.shape[0])
        out["isect_ids_sorted"] = bool((ids[1:] >= ids[:-1]).all())
        o = torch.cat([offs.reshape(-1).long(), torch.tensor([flat.shape[0]], device=dev)])
        lens = o[1:] - o[:-1]
        out["longest_tile_list"] = int(lens.max()); out["tiles_over_4096"] = int((lens > 4096).sum()); out["tiles_over_16384"] = int((lens > 16384).sum())
        vis = (rad[0] > 0).all(-1)
        op = m.get_opacity().reshape(-1)
        d0 = dep[0]
        out["visible"] = int(vis.sum())
        for thr in (0.05, 0.2, 0.5, 1.0, 2.0):
            sel = vis & (d0 < thr)
            out[f"depth<{thr}"] = [int(sel.sum()), int((sel & (op > 0.05)).sum())]
        out["radius>500px"] = int((rad[0].max(-1).values > 500).sum())
        out["depth_quantiles_visible"] = [round(float(x), 3) for x in torch.quantile(d0[vis][:2000000].float(), torch.tensor([0.001, 0.01, 0.1, 0.5, 0.9], device=dev))]
        # renders without the Gaussians nearer than thr to ANY of the evaluation cameras
        for thr in (0.5, 1.5):
            near = torch.zeros_like(vis)
            for c in cameras:
                vm = c.world_view_transform[0]
                z = (m.get_means() @ vm[:3, :3].T + vm[:3, 3])[:, 2]
                near |= (z > 0) & (z < thr)
            ro = m.raw_opacities.detach().clone(); ro[near] = -20.0
            m2_ = SplatModel(m.means.detach(), m.sh0.detach(), m.shN.detach(), m.raw_scales.detach(), m.raw_quats.detach(), ro, m.max_sh_degree, active_sh_degree=m.active_sh_degree)
            pe = [evaluate.psnr(torch.clamp(fast_rasterize(c, m2_, bg0).image, 0, 1), im) for c, im in zip(cameras, images)]
            pg = [evaluate.psnr(torch.clamp(rasterize(c, m2_, bg0).image, 0, 1), im) for c, im in zip(cameras, images)]
            out[f"without_z<{thr}"] = {"dropped": int(near.sum()), "psnr_ewa": round(float(np.mean(pe)), 3), "psnr_gut": round(float(np.mean(pg)), 3)}
        img_hip = r.image.cpu().numpy()
        img_ewa = torch.clamp(fast_rasterize(cameras[0], m, bg0).image, 0, 1).cpu().numpy()
    # the oracle (CPU restatement of the reference's 3DGUT kernels) on the same model, same view
    sys.path.insert(0, ROOT)
    from oracle import pipeline
    sc = scenes.Scene("fin", W, H, m.max_sh_degree, m.means.detach().cpu(), m.raw_quats.detach().cpu(), m.raw_scales.detach().cpu(), m.raw_opacities.detach().cpu().reshape(-1),
                      m.sh0.detach().cpu(), m.shN.detach().cpu(), cameras[0].world_view_transform.cpu(), cameras[0].K.cpu())
    t0 = time.time()
    o = pipeline.train_image(pipeline.scene_arrays(sc), 0, np.zeros((3, H, W), np.float32), backward=False)
    img_or = np.clip(o["render"][0].transpose(2, 0, 1), 0, 1)
    mse = lambda a, b: float(((a.astype(np.float64) - b) ** 2).mean())
    out["oracle_seconds"] = round(time.time() - t0, 1)
    out["psnr(hip_gut, oracle_gut)"] = round(10 * np.log10(1.0 / max(mse(img_hip, img_or), 1e-20)), 2)
    out["psnr(hip_ewa, oracle_gut)"] = round(10 * np.log10(1.0 / max(mse(img_ewa, img_or), 1e-20)), 2)
    out["psnr(oracle_gut, target)"] = round(10 * np.log10(1.0 / max(mse(img_or, images[0].cpu().numpy()), 1e-20)), 2)
    out["oracle_n_isects"] = int(len(o["flatten_ids"]))
    print("DIAG", json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
