#!/usr/bin/env python3
"""Conditioning probe for the backward rasterizer (K8) on the EMULATED product library (no GPU; with --gpu: on the MI355X through the shipped library): ONE flat Gaussian - two scales 0.04, the third from 0.04 down to 5e-4 - in front of a
pinhole camera; dL/d(quaternion), dL/d(scales), dL/d(mean) of the HIP kernels (host IEEE fp32 under emulation) and of the oracle evaluated in fp32, each against the oracle in fp64,
on a 48 x 48 image, six random orientations each (worst shown); the forward image against the fp64 oracle beside it. Found by tools/fuzz_emulated.py --oracle (seed 12, case 1106):
profiles/r04/fuzz_emulated.txt.    python tools/aniso_probe.py        LFS_EMUL_DEFINES=-DLFS_BWD_REORTH=1 python tools/aniso_probe.py   (the build with the re-orthogonalised foot vector)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import emul_util  # noqa: E402


def main():
    """prints the table; with --json <path> also writes the rows [{size, thin, aspect, v_quats_hip, v_quats_o32, v_scales_hip, v_scales_o32, v_means_hip, v_means_o32, fwd}]"""
    import json
    rows_out = []
    import oracle as orc
    orc.build(ref=False)
    import contextlib
    on_gpu = "--gpu" in sys.argv
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt).contiguous().to("cuda:0" if on_gpu else "cpu")
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / (np.linalg.norm(b) + 1e-300))
    with (contextlib.nullcontext() if on_gpu else emul_util.installed()):
        import lichtfeld_studio_amd as lfs
        from lichtfeld_studio_amd import ops
        cm, st = lfs.CameraModelType(0), lfs.ShutterType(4)
        if on_gpu and "--deterministic" in sys.argv:   # order-independent accumulation (debug bit 4): the table is then the same on every run - with float atomics dL/dscales moves by 1e-4 .. 1e-3 between runs
            lfs.load_library().lfs_set_debug_flags(16)
        print("image  thin scale  aspect | v_quats: HIP / oracle-fp32 (rel. to fp64) | v_scales: HIP / oracle-fp32 | v_means: HIP / oracle-fp32 | forward: max |colour - fp64 oracle|")
        for size in (48,):
            for thin in (0.04, 0.01, 0.004, 0.002, 0.001, 0.0005):
                worst = np.zeros(7)
                for seed in range(6):
                    rng = np.random.default_rng(seed)
                    W = H = size
                    means = np.array([[0.0, 0.0, 5.0]], np.float32) + (rng.standard_normal((1, 3)) * 0.01).astype(np.float32)
                    quats = rng.standard_normal((1, 4)).astype(np.float32)
                    scales = np.array([[thin, 0.04, 0.04]], np.float32)
                    opac = np.array([0.8], np.float32)
                    vm = np.eye(4, dtype=np.float32)[None]
                    f = 600.0 * size / 48 if size > 1 else 600.0
                    K = np.array([[[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]]], np.float32)
                    radii, m2, d, conics, _ = ops.projection_ut_3dgs_fused(t(means), t(quats), t(scales), t(opac), t(vm), None, t(K), W, H, 0.3, 0.01, 1e4, 0.0, False, cm, None, st, None, None, None)
                    ts = 16
                    tw, th = (W + ts - 1) // ts, (H + ts - 1) // ts
                    _, ids, flat, offs = ops.intersect_tile(m2, radii, d, None, None, 1, ts, tw, th, True, return_offsets=True)
                    if len(flat) == 0:
                        continue
                    colors = rng.random((1, 1, 3)).astype(np.float32)
                    args = (t(means), t(quats), t(scales), t(colors), t(opac[None]), None, None, W, H, ts, t(vm), None, t(K), cm, None, st, None, None, None, offs, flat)
                    rc, ra, li = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)
                    v_rc, v_ra = t(rng.standard_normal(tuple(rc.shape)).astype(np.float32)), t(rng.standard_normal(tuple(ra.shape)).astype(np.float32))
                    g = [x.cpu() for x in ops.rasterize_to_pixels_from_world_3dgs_bwd(*args, ra, li, v_rc, v_ra)]
                    rc, ra, li, v_rc, v_ra, offs, flat = [x.cpu() for x in (rc, ra, li, v_rc, v_ra, offs, flat)]
                    oargs = (means, quats, scales, colors, opac[None], None, None, W, H, ts, vm, None, K, 0, 4, None, None, None, offs.numpy(), flat.numpy())
                    o64 = orc.rasterize_bwd(*oargs, ra.numpy(), li.numpy(), v_rc.numpy(), v_ra.numpy(), dtype=np.float64)
                    o32 = orc.rasterize_bwd(*oargs, ra.numpy(), li.numpy(), v_rc.numpy(), v_ra.numpy(), dtype=np.float32)
                    if np.linalg.norm(o64[1]) == 0:
                        continue
                    e = [rel(g[1].numpy(), o64[1]), rel(o32[1], o64[1]), rel(g[2].numpy(), o64[2]), rel(o32[2], o64[2]), rel(g[0].numpy(), o64[0]), rel(o32[0], o64[0])]
                    o_rc, o_ra, _ = orc.rasterize_fwd(*oargs, dtype=np.float64)
                    e.append(float(np.abs(rc.numpy() - o_rc).max()))
                    worst = np.maximum(worst, e)
                rows_out.append(dict(size=size, thin=thin, aspect=0.04 / thin, v_quats_hip=worst[0], v_quats_o32=worst[1], v_scales_hip=worst[2], v_scales_o32=worst[3],
                                     v_means_hip=worst[4], v_means_o32=worst[5], fwd=worst[6]))
                print(f"{size:3d}x{size:<3d} {thin:8.4f}  {0.04 / thin:6.0f} | {worst[0]:.1e} / {worst[1]:.1e} | {worst[2]:.1e} / {worst[3]:.1e} | {worst[4]:.1e} / {worst[5]:.1e} | {worst[6]:.1e}")
    if "--json" in sys.argv:
        json.dump(rows_out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
