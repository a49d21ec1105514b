#!/usr/bin/env python
"""Writes a COLMAP directory (sparse/0/{cameras,images,points3D}.bin + images/*.png) rendered from a known Gaussian scene with the 3DGUT
rasterizer: a stand-in for a Mip-NeRF360 capture where no dataset can be downloaded (BASELINE config 3 analogue). The sparse points are a
subsample of the ground-truth centres with their base colours, as a structure-from-motion cloud would be.

    python tools/make_synthetic_colmap.py out_dir --views 96 --width 1296 --height 840 --gaussians 300000 --points 60000
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--views", type=int, default=96)
    ap.add_argument("--width", type=int, default=1296)
    ap.add_argument("--height", type=int, default=840)
    ap.add_argument("--gaussians", type=int, default=300000)
    ap.add_argument("--points", type=int, default=60000)
    ap.add_argument("--seed", type=int, default=11)
    args = ap.parse_args()
    import lichtfeld_studio_amd  # noqa: F401
    from lichtfeld_studio_amd import loader, scenes
    from lichtfeld_studio_amd.rasterizer import rasterize
    from lichtfeld_studio_amd.trainer import GutTrainer
    from oracle import colmap_io as oc       # the documented COLMAP byte layouts (a test / tooling helper, not the product path)
    from scipy.spatial.transform import Rotation
    dev = torch.device("cuda:0")
    sc = scenes._syn_box("SYN-COLMAP", args.seed, args.gaussians, args.width, args.height, 0.9 * args.width, args.views, sh_degree=1)
    sc.raw_scales += float(np.log(3.0))
    tr = GutTrainer(sc, dev, iterations=10)
    base = args.out
    os.makedirs(os.path.join(base, "sparse", "0"), exist_ok=True)
    os.makedirs(os.path.join(base, "images"), exist_ok=True)
    images = []
    for v in range(args.views):
        with torch.no_grad():
            img = rasterize(tr.camera(v), tr.model, tr.bg, 1.0, False, False).image
        loader.write_png(os.path.join(base, "images", f"v{v:03d}.png"), (img.clamp(0, 1).permute(1, 2, 0) * 255 + 0.5).to(torch.uint8).cpu().numpy())
        m = sc.viewmats[v].double().cpu().numpy()
        q = Rotation.from_matrix(m[:3, :3]).as_quat()
        images.append((v + 1, [q[3], q[0], q[1], q[2]], list(m[:3, 3]), 1, f"v{v:03d}.png"))
    K = sc.Ks[0]
    oc.write_cameras_bin(os.path.join(base, "sparse", "0", "cameras.bin"), [(1, 1, args.width, args.height, [float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])])])
    oc.write_images_bin(os.path.join(base, "sparse", "0", "images.bin"), images)
    idx = np.random.default_rng(args.seed).choice(args.gaussians, size=min(args.points, args.gaussians), replace=False)
    xyz = sc.means.numpy()[idx]
    rgb = np.clip((sc.sh0.numpy()[idx, 0] * 0.28209479177387814 + 0.5) * 255, 0, 255).astype(np.uint8)
    oc.write_points3d_bin(os.path.join(base, "sparse", "0", "points3D.bin"), xyz, rgb)
    print(f"wrote {args.views} views {args.width}x{args.height}, {len(idx)} points to {base}")


if __name__ == "__main__":
    main()
