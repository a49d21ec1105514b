#!/usr/bin/env python3
"""Shape fuzzer for the hot-path operator chain on the EMULATED product library (tests/emul_util.py; no GPU): random small problems - Gaussian counts 0 .. 400, images
from 1 x 1 to ~120 x 120, tile sizes 8 / 16 / 24 / 32 / 48, 1 - 2 cameras, 1 - 4 channels, SH degree 0 - 4, every camera model / shutter, optional masks and
backgrounds - through  projection_ut_3dgs_fused -> spherical_harmonics_fwd -> intersect_tile (sorted, with offsets) -> rasterize fwd -> rasterize bwd ->
spherical_harmonics_bwd -> adam_step.  Checked per case: outputs finite, tile lists sorted by (tile, depth), offsets consistent with the list, alpha in [0, 1],
culling on / off bit-identical in the forward, and - with the library built under AddressSanitizer - every global-memory access of every emulated lane in bounds:

    ASAN_LIB=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
    LFS_EMUL_SANITIZE=1 ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1 LD_PRELOAD=$ASAN_LIB python tools/fuzz_emulated.py --seconds 600

Results of the runs of record: profiles/r04/fuzz_emulated.txt."""
import argparse
import os
import re
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import emul_util  # noqa: E402
from gpu_util import small_rotation_viewmat  # noqa: E402


STATS = {}
DEV = "cpu"          # --gpu: "cuda:0" - the same cases through the SHIPPED gfx950 library on the MI355X (the parity run proper; the emulated run checks the logic)
FLAT_P = 0.0         # --flat P: with probability P a case's Gaussians are flat disks (one random axis 10 / 30 / 100 x thinner): the regime of K8's re-orthogonalisation


def _np(x):
    return x.detach().cpu().numpy()


def _stat(name, value):
    a = STATS.setdefault(name, [0, 0.0])
    a[0] += 1
    a[1] = max(a[1], float(value))


REPLAY = False


def one_case(rng, lfs, ops, lib, idx, orc=None):
    N = int(rng.choice([0, 1, 2, 63, 64, 65, int(rng.integers(3, 400))]))
    W, H = int(rng.integers(1, 121)), int(rng.integers(1, 121))
    ts = int(rng.choice([8, 16, 24, 32, 48]))
    Cn = int(rng.choice([1, 1, 2]))
    cdim = int(rng.choice([1, 2, 3, 3, 4]))
    model = int(rng.choice([0, 0, 0, 2]))                 # PINHOLE / FISHEYE
    shutter = int(rng.choice([4, 4, 4, 0, 1, 2, 3]))
    desc = dict(idx=idx, N=N, W=W, H=H, ts=ts, C=Cn, cdim=cdim, model=model, shutter=shutter)
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt).contiguous().to(DEV)
    means = rng.standard_normal((N, 3)).astype(np.float32) * float(rng.choice([0.3, 1.0, 3.0]))
    means[:, 2] = np.abs(means[:, 2]) + float(rng.choice([0.2, 1.5, 3.0]))
    quats = rng.standard_normal((N, 4)).astype(np.float32)
    smax = float(rng.choice([0.05, 0.3, 2.0]))
    scales = (rng.random((N, 3)) * smax + 1e-3).astype(np.float32)
    opac = (rng.random(N) * 0.98 + 0.01).astype(np.float32)
    if FLAT_P > 0 and rng.random() < FLAT_P and N:
        aspect = float(rng.choice([10.0, 30.0, 100.0]))
        thin_axis = rng.integers(0, 3, N)
        scales[np.arange(N), thin_axis] = np.maximum(scales.max(-1) / np.float32(aspect), np.float32(2e-4))
        desc["flat"] = aspect
    vm0 = np.stack([small_rotation_viewmat(rng, 0.05 + 0.2 * c, 0.2) for c in range(Cn)]).astype(np.float32)
    vm1 = np.stack([small_rotation_viewmat(rng, 0.1, 0.3) for _ in range(Cn)]).astype(np.float32) if shutter != 4 else None
    f = float(rng.uniform(0.4, 1.5)) * max(W, H)
    K = np.tile(np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], np.float32), (Cn, 1, 1))
    rad = tan = thin = None
    if model == 2:
        rad = (rng.standard_normal((Cn, 4)) * 0.01).astype(np.float32)
    elif rng.random() < 0.3:
        rad = (rng.standard_normal((Cn, 4 if rng.random() < 0.5 else 6)) * 0.02).astype(np.float32)
        tan = (rng.standard_normal((Cn, 2)) * 0.002).astype(np.float32)
        if rng.random() < 0.5:
            thin = (rng.standard_normal((Cn, 4)) * 0.001).astype(np.float32)
    cm, st = lfs.CameraModelType(model), lfs.ShutterType(shutter)
    radii, m2, d, conics, _ = ops.projection_ut_3dgs_fused(t(means), t(quats), t(scales), t(opac), t(vm0), None if vm1 is None else t(vm1), t(K), W, H, 0.3, 0.01, 1e4, 0.0,
                                                           False, cm, None, st, None if rad is None else t(rad), None if tan is None else t(tan), None if thin is None else t(thin))
    for x in (m2, d, conics):
        assert torch.isfinite(x).all(), (desc, "projection")
    if orc is not None and N:
        # --oracle: the differential mode - every stage against the CPU restatement of the reference's kernels (oracle/), at the bars of the GPU parity tests
        o_radii, o_m2, o_d, o_conics, _ = orc.projection_ut_3dgs_fused(means, quats, scales, opac, vm0, vm1, K, W, H, 0.3, 0.01, 1e4, 0.0, False, model, None, shutter, rad, tan, thin)
        r_n = _np(radii)
        both = (r_n > 0).all(-1) & (o_radii > 0).all(-1)
        vis_diff = int(((r_n > 0).all(-1) != (o_radii > 0).all(-1)).sum())      # a Gaussian on the culling boundary may land on either side (radii +-1)
        _stat("projection: visibility differs (Gaussians per case)", vis_diff)
        cond = None
        if DEV == "cpu":
            assert vis_diff <= max(1, (Cn * N) // 100), (desc, "projection visibility", vis_diff)
        else:
            # On the GPU the transcendental functions of the camera models (fisheye: atan / sin / cos, the rolling shutter's ten fixed-point iterations) differ from the host's
            # by an ulp, and the unscented transform (weights -99 / +16.67) amplifies that by the CONDITIONING of the case - screen-filling Gaussians under a rolling-shutter
            # fisheye camera move their conics by percent under a 3e-7 relative change of the inputs. The bar therefore grows with the measured sensitivity: the oracle is
            # evaluated a second time on inputs perturbed by 1e-6 (relative, random signs, own generator: the case stream is untouched), and 512 x that response is allowed on top (well-conditioned rows respond with ~1e-7: their bar stays the plain one).
            rng2 = np.random.default_rng(idx + 7919)
            pert = lambda x: (x * (1.0 + 1e-6 * rng2.choice([-1.0, 1.0], x.shape))).astype(np.float32)   # (+-8 ulp on EVERY element: 3e-7 x randn left one element in eight unchanged)
            # (the camera too: the rolling shutter's pose interpolation - acos / sin of the slerp - is where the GPU's libm and the host's differ first)
            p_radii, p_m2, p_d, p_conics, _ = orc.projection_ut_3dgs_fused(pert(means), pert(quats), pert(scales), opac, pert(vm0), None if vm1 is None else pert(vm1), pert(K), W, H, 0.3,
                                                                           0.01, 1e4, 0.0, False, model, None, shutter, rad, tan, thin)
            vis_flips = int(((p_radii > 0).all(-1) != (o_radii > 0).all(-1)).sum())   # Gaussians whose visibility the 1e-6 perturbation alone flips: sigma points on the margin
            _stat("projection: visibility flips of the oracle under the 1e-6 perturbation (Gaussians per case)", vis_flips)
            # (rolling shutter / distortion: a sigma point whose ten-iteration fixed point ends within an ulp of the image margin lands on either side - two such Gaussians
            #  of 130 in case 1016 of seed 47, none of them moved by the perturbation of the Gaussians)
            base = 2 if (shutter != 4 or rad is not None or model == 2) else 1
            assert vis_diff <= max(base, (Cn * N) // 100, 4 * vis_flips), (desc, "projection visibility", vis_diff, "flipped by the perturbation alone", vis_flips)
            both &= (p_radii > 0).all(-1)
            # degenerate outcomes of the unscented transform - the weighted mean (-99 x centre + 16.67 x the six others) lands more than ten image sizes away from the image,
            # radii of thousands of pixels on a 26 x 4 image: sigma points on both sides of a rolling-shutter / distortion fold - are chaotic in fp32 on every implementation;
            # the rows are counted and left out of the value comparison (their visibility and radii-within-bar checks above still apply)
            far = (np.abs(o_m2) > 10.0 * max(W, H)).any(-1)
            _stat("projection: degenerate UT rows left out (|means2d| > 10 image sizes; Gaussians per case)", int((far & both).sum()))
            both &= ~far
            cond = {"means2d": np.abs(p_m2 - o_m2), "depths": np.abs(p_d - o_d), "conics": np.abs(p_conics - o_conics), "radii": np.abs(p_radii - o_radii)}
        if both.any():
            r_bar = 1 if cond is None else 1 + 512 * int(cond["radii"][both].max())
            if cond is not None:
                # (round 5, case 2782 of seed 59 on the MI355X: a needle under a rolling shutter, 2-D covariance of condition number ~1e4 - radius 1517 px on a 68 x 106 image,
                #  the conic moving by 0.9 % under the 1e-6 perturbation and by 8 % between the two libms, the INTEGER radius by 0 and by 2: the perturbation response of an
                #  integer cannot resolve a sensitivity below one pixel. A radius of more than ten image sizes - clamped to the image by the tile stage whatever its value -
                #  gets 1 % of itself on top of the bar; counted.)
                huge = (o_radii > 10.0 * max(W, H)) & both[..., None]
                _stat("projection: radii beyond ten image sizes (held to the bar + 1 % of the radius; values per case)", int(huge.sum()))
                r_bar = r_bar + np.where(huge, np.ceil(1e-2 * o_radii), 0).astype(np.int64)
            else:
                r_bar = np.full(o_radii.shape, r_bar, np.int64)
            if (np.abs(r_n - o_radii)[both] > r_bar[both]).any():   # where, and what the three evaluations say there
                for cg in np.argwhere(both & (np.abs(r_n - o_radii) > r_bar).any(-1)):
                    c_, g_ = int(cg[0]), int(cg[1])
                    print(f"  radii beyond the bar {r_bar[c_, g_]} at (camera, Gaussian) = ({c_}, {g_}): HIP {r_n[c_, g_]} oracle {o_radii[c_, g_]} perturbed oracle {None if cond is None else p_radii[c_, g_]}; "
                          f"means2d HIP {_np(m2)[c_, g_]} oracle {o_m2[c_, g_]}{'' if cond is None else f' perturbed {p_m2[c_, g_]}'}; conics HIP {_np(conics)[c_, g_]} oracle {o_conics[c_, g_]}"
                          f"{'' if cond is None else f' perturbed {p_conics[c_, g_]}'}; depth HIP {_np(d)[c_, g_]} oracle {o_d[c_, g_]}; scales {scales[g_]} quat {quats[g_]} mean {means[g_]} "
                          f"opacity {opac[g_]}; K {K[c_].tolist()} distortion {None if rad is None else rad[c_].tolist()}", flush=True)
            assert (np.abs(r_n - o_radii)[both] <= r_bar[both]).all(), (desc, "radii")
            for nm, a, b in (("means2d", _np(m2)[both], o_m2[both]), ("depths", _np(d)[both], o_d[both]), ("conics", _np(conics)[both], o_conics[both])):
                allow = 1e-4 + 1e-4 * np.abs(b) + (0.0 if cond is None else 512.0 * cond[nm][both])
                if cond is not None:   # never tighter than what the -m gpu suite holds the projection to (tests/test_oracle_refk_golden.py::check_projection: the UT's fp32 noise floor)
                    suite = {"means2d": 1e-2, "depths": 1e-5 * max(1.0, float(np.abs(b).max())), "conics": 1e-2 * np.abs(b).max(-1, keepdims=True)}[nm]
                    allow = np.maximum(allow, suite)
                err = float(np.max(np.abs(a - b) / allow))
                _stat(f"projection: {nm} |diff| / (atol 1e-4 + rtol 1e-4{'' if cond is None else ' + 512 x the response to a 1e-6 input perturbation'})", err)
                if cond is not None:
                    _stat(f"projection: {nm} conditioning term / (atol + rtol) (how far the plain bar was exceeded by the case itself)", float(np.max(512.0 * cond[nm][both] / (1e-4 + 1e-4 * np.abs(b)))))
                if err > 1.0:   # where, and what both sides say there
                    flat_i = int(np.argmax((np.abs(a - b) / allow).reshape(-1)))
                    row = flat_i // (a.shape[-1] if a.ndim > 1 else 1)
                    cg = np.argwhere(both)[row]
                    print(f"  {nm} worst at (camera, Gaussian) = {tuple(int(x) for x in cg)}: HIP {a[row]} oracle {b[row]}; radii HIP {r_n[tuple(cg)]} oracle {o_radii[tuple(cg)]}; "
                          f"means2d HIP {_np(m2)[tuple(cg)]} oracle {o_m2[tuple(cg)]}; depth {o_d[tuple(cg)]}; scales {scales[cg[1]]}; mean {means[cg[1]]}; "
                          f"response to the perturbation {None if cond is None else cond[nm][tuple(cg)]}", flush=True)
                assert err <= 1.0, (desc, nm, err)
    deg = int(rng.integers(0, 5))
    Kc = int(rng.choice([k for k in (1, 4, 9, 16, 25) if k >= (deg + 1) ** 2]))
    coeffs = (rng.standard_normal((N, Kc, 3)) * 0.3).astype(np.float32)
    dirs = rng.standard_normal((N, 3)).astype(np.float32)
    smask = (rng.random(N) < 0.8)
    use_smask = rng.random() < 0.7
    col3 = ops.spherical_harmonics_fwd(deg, t(dirs), t(coeffs), t(smask, torch.bool) if use_smask else None)
    assert torch.isfinite(col3).all(), (desc, "sh_fwd")
    if orc is not None and N:
        want = orc.spherical_harmonics_fwd(deg, dirs, coeffs, smask if use_smask else None)
        err = float(np.max(np.abs(_np(col3) - want) / (1e-5 + 1e-5 * np.abs(want))))
        _stat("sh_fwd: |diff| / (atol 1e-5 + rtol 1e-5)", err)
        assert err <= 1.0, (desc, "sh_fwd vs oracle", err)
    tw, th = (W + ts - 1) // ts, (H + ts - 1) // ts
    tpg, ids, flat, offs = ops.intersect_tile(m2, radii, d, None, None, Cn, ts, tw, th, True, return_offsets=True)
    ids_n, flat_n, offs_n = _np(ids), _np(flat), _np(offs).reshape(-1)
    assert int(tpg.sum()) == len(ids_n) == len(flat_n), (desc, "isect count")
    assert np.all(np.diff(ids_n) >= 0), (desc, "isect order")                      # (camera | tile | depth bits) ascending
    assert np.all(np.diff(offs_n) >= 0) and (len(offs_n) == 0 or offs_n[-1] <= len(ids_n)), (desc, "offsets")
    assert np.all((flat_n >= 0) & (flat_n < max(Cn * N, 1))), (desc, "flatten ids")
    if orc is not None and N:
        o_tpg, o_ids, o_flat = orc.intersect_tile(_np(m2), _np(radii), _np(d), Cn, ts, tw, th, True)     # integer stage: bit-exact on identical inputs
        assert np.array_equal(_np(tpg).reshape(o_tpg.shape), o_tpg) and np.array_equal(ids_n, o_ids) and np.array_equal(flat_n, o_flat), (desc, "isect vs oracle")
        assert np.array_equal(offs_n, orc.intersect_offset(o_ids, Cn, tw, th).reshape(-1)), (desc, "offsets vs oracle")
        _stat("intersect_tile / intersect_offset: bit-exact cases", 0)
    colors = rng.random((Cn, N, cdim)).astype(np.float32)
    opacs = np.tile(opac[None], (Cn, 1)).astype(np.float32)
    bg = rng.random((Cn, cdim)).astype(np.float32) if rng.random() < 0.6 else None
    masks = (rng.random((Cn, th, tw)) > 0.25) if rng.random() < 0.3 else None
    args = (t(means), t(quats), t(scales), t(colors), t(opacs), None if bg is None else t(bg), None if masks is None else t(masks, torch.bool), W, H, ts, t(vm0),
            None if vm1 is None else t(vm1), t(K), cm, None, st, None if rad is None else t(rad), None if tan is None else t(tan), None if thin is None else t(thin), offs, flat)
    rc, ra, li = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)
    assert torch.isfinite(rc).all() and float(ra.min()) >= 0.0 and float(ra.max()) <= 1.0 + 1e-6, (desc, "raster fwd")
    if orc is not None and N:
        oargs = (means, quats, scales, colors, opacs, bg, masks, W, H, ts, vm0, vm1, K, model, shutter, rad, tan, thin, _np(offs), flat_n)
        o_rc, o_ra, o_li = orc.rasterize_fwd(*oargs)
        h_rc, h_ra = _np(rc).copy(), _np(ra).copy()
        if DEV != "cpu" and (rad is not None or model == 2):
            # distorted cameras: a pixel's ray exists only if the undistortion converged (Newton, 5 iterations, |step| < 1e-6: Cameras.cuh:473-755; fisheye: 20 iterations) -
            # a pixel whose last step sits AT the threshold has a ray on one side and none on the other (one pixel of 10 824 in case 1086 of seed 31, found by this fuzzer).
            # Such pixels - nothing composited on exactly one side - are counted, bounded, and taken out of the image comparison.
            none_h = (h_ra.reshape(o_li.shape) == 0) & (_np(li) == 0)
            none_o = (o_ra.reshape(o_li.shape) == 0) & (o_li == 0)
            flip = none_h != none_o
            _stat("raster fwd: ray-validity flips (distorted cameras; pixels per case)", int(flip.sum()))
            assert int(flip.sum()) <= max(2, int(1e-3 * Cn * W * H)), (desc, "ray validity flips", int(flip.sum()))
            if flip.any():
                h_rc.reshape(o_li.shape + (cdim,))[flip] = o_rc.reshape(o_li.shape + (cdim,))[flip]
                h_ra.reshape(o_li.shape)[flip] = o_ra.reshape(o_li.shape)[flip]
        dimg, dalp = np.abs(h_rc - o_rc), np.abs(h_ra - o_ra)
        # a contribution whose alpha sits on the 1/255 threshold is taken on one side only: |d alpha| = T / 255 at that pixel - below the "beyond" bar of the next lines, yet on a
        # 4 x 84 image ONE such pixel is a mean of 1.17e-5 (case 284 of seed 71, emulated: mean-alpha bar 1e-5 failed on exactly one flip, last_ids differing at that one pixel).
        # Pixels that moved by more than half a threshold contribution are counted, bounded like the other flips, and left out of the MEANS.
        beyond = int((dimg.reshape(-1, cdim).max(-1) > 1 / 255 + 1e-4).sum())
        if dimg.size:
            pix = np.maximum(dimg.reshape(-1, cdim).max(-1), dalp.reshape(-1))
            tflip = pix > 0.5 / 255
            _stat("raster fwd: threshold-flip pixels left out of the means (per case)", int(tflip.sum()))
            assert int(tflip.sum()) <= max(2, int(2e-3 * Cn * W * H)), (desc, "raster fwd threshold flips", int(tflip.sum()))
            if tflip.any() and not tflip.all():
                dimg = dimg.reshape(-1, cdim)[~tflip]
                dalp = dalp.reshape(-1)[~tflip]
        _stat("raster fwd: mean |colour diff| (bar 1e-5)", dimg.mean() if dimg.size else 0.0)
        _stat("raster fwd: pixels beyond 1/255 + 1e-4", beyond)
        fwd_ok = (dimg.mean() if dimg.size else 0.0) <= 1e-5 and dalp.mean() <= 1e-5
        limited = False
        if not fwd_ok:
            # flat / huge / near Gaussians: gro = M (o - mu) is 1e4 .. 1e5 long and fp32 resolves alpha only to 1e-3 - in the reference's arithmetic as much as in ours
            # (the oracle's fp32 evaluation is the reference's arithmetic in the reference's precision). Such a case is a parity failure only if the kernels are further
            # from the TRUTH (the oracle in fp64) than four times the oracle's own fp32 evaluation is.
            t_rc, t_ra, _ = orc.rasterize_fwd(*oargs, dtype=np.float64)
            e_hip = (float(np.abs(h_rc - t_rc).mean()), float(np.abs(h_ra - t_ra).mean()))
            e_o32 = (float(np.abs(o_rc - t_rc).mean()), float(np.abs(o_ra - t_ra).mean()))
            _stat("raster fwd: fp32-limited cases, (HIP vs fp64) / (oracle fp32 vs fp64), colour", e_hip[0] / max(e_o32[0], 1e-30))
            _stat("raster fwd: fp32-limited cases, (HIP vs fp64) / (oracle fp32 vs fp64), alpha", e_hip[1] / max(e_o32[1], 1e-30))
            limited = e_hip[0] <= max(1e-5, 4 * e_o32[0]) and e_hip[1] <= max(1e-5, 4 * e_o32[1])
            e_resp = None
            if not limited and DEV != "cpu":
                # ... or the case itself is that sensitive: the GPU's sin / cos / acos (rolling-shutter pose interpolation per pixel) and v_rsq / v_rcp differ from the host's by an
                # ulp, which the oracle's fp32-vs-fp64 distance does not see (both use the host's libm). The oracle's response, in fp64, to a 1e-6 relative perturbation
                # of the Gaussians measures what an ulp-level difference of the ray / record arithmetic can do to THIS image.
                rng3 = np.random.default_rng(idx + 104729)
                pert = lambda x: (x * (1.0 + 1e-6 * rng3.choice([-1.0, 1.0], x.shape))).astype(np.float32)
                p_rc, p_ra, _ = orc.rasterize_fwd(pert(means), pert(quats), pert(scales), *oargs[3:10], pert(vm0), None if vm1 is None else pert(vm1), pert(K), *oargs[13:], dtype=np.float64)
                e_resp = (float(np.abs(p_rc - t_rc).mean()), float(np.abs(p_ra - t_ra).mean()))
                _stat("raster fwd: conditioning-limited cases, (HIP vs fp64) / (response of the fp64 oracle to a 1e-6 input perturbation), alpha", e_hip[1] / max(e_resp[1], 1e-30))
                limited = e_hip[0] <= max(1e-5, 16 * e_resp[0]) and e_hip[1] <= max(1e-5, 16 * e_resp[1])
            assert limited, (desc, "raster fwd vs oracle", float(dimg.mean()), float(dalp.mean()), "vs fp64: HIP", e_hip, "oracle fp32", e_o32, "response to 1e-6", e_resp)
        lim = max(1, int(1e-3 * Cn * W * H)) * (20 if limited else 1)
        assert beyond <= lim, (desc, "raster fwd flips", beyond)
        li_diff = int((_np(li) != o_li).sum())
        _stat("raster fwd: last_ids differing pixels", li_diff)
        assert li_diff <= lim, (desc, "last_ids", li_diff)
    lib.lfs_set_debug_flags(1)       # culling off: the cell lists are the tile lists - the same image, bit for bit
    try:
        rc2, ra2, li2 = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)
    finally:
        lib.lfs_set_debug_flags(0)
    assert torch.equal(rc, rc2) and torch.equal(ra, ra2) and torch.equal(li, li2), (desc, "culling on / off")
    v_rc, v_ra = t(rng.standard_normal(tuple(rc.shape)).astype(np.float32)), t(rng.standard_normal(tuple(ra.shape)).astype(np.float32))
    grads = ops.rasterize_to_pixels_from_world_3dgs_bwd(*args, ra, li, v_rc, v_ra)
    for g in grads:
        assert torch.isfinite(g).all(), (desc, "raster bwd")
    if orc is not None and N and len(flat_n):
        # both sides continue from THIS forward's alphas / last ids; the oracle in fp64 is the truth (tests/test_gpu_raster.py: <= 2e-4 with counted flip rows)
        og = orc.rasterize_bwd(*oargs, _np(ra), _np(li), _np(v_rc), _np(v_ra), dtype=np.float64)
        from gpu_util import rows_check
        for nm, a, b in zip(("v_means", "v_quats", "v_scales", "v_colors", "v_opacities"), grads, og):
            a, b = _np(a).astype(np.float64), np.asarray(b, np.float64)
            if nm in ("v_colors", "v_opacities"):
                a, b = a.reshape(Cn * N, -1), b.reshape(Cn * N, -1)
            else:
                a, b = a.reshape(N, -1), b.reshape(N, -1)
            if np.sqrt((b ** 2).sum()) < 1e-12:
                assert np.abs(a).max() < 1e-9, (desc, nm, "oracle gradient is zero")
                continue
            # (one threshold flip at a pixel moves the gradient of every Gaussian behind it in that pixel's list: case 1798 of seed 47 has six such rows of 65, every other row at 3e-7)
            MF = max(4, N // 8)
            e, flips, rest = rows_check(a, b, bar=2e-4, max_flips=MF)
            if REPLAY:   # --replay: where the difference sits, and what the ORACLE's own fp32 evaluation does against its fp64 one on the same case
                o32 = np.asarray(orc.rasterize_bwd(*oargs, _np(ra), _np(li), _np(v_rc), _np(v_ra), dtype=np.float32)[("v_means", "v_quats", "v_scales", "v_colors", "v_opacities").index(nm)], np.float64).reshape(b.shape)
                rows = np.sqrt(((a - b) ** 2).sum(1)) / np.sqrt((b ** 2).sum())
                rows32 = np.sqrt(((o32 - b) ** 2).sum(1)) / np.sqrt((b ** 2).sum())
                top = np.argsort(rows)[::-1][:8]
                print(f"  {nm}: rel-L2 {e:.3e}, flips {flips}, rest {rest:.3e}; oracle fp32 vs fp64: {rows_check(o32, b, bar=2e-4, max_flips=4)}; HIP vs oracle fp32: {rows_check(a, o32, bar=2e-4, max_flips=4)}")
                print("    worst rows (row, HIP-vs-fp64 share, oracle-fp32-vs-fp64 share):", [(int(r), float(f"{rows[r]:.2e}"), float(f"{rows32[r]:.2e}")) for r in top])
                if nm == "v_quats":
                    for r in top[:6]:
                        print(f"    row {int(r)}: scales {scales[r]}, opacity {opac[r]:.3f}, mean {means[r]}, |v_quats| HIP {np.linalg.norm(a[r]):.4e} oracle64 {np.linalg.norm(b[r]):.4e}, "
                              f"row-relative error {np.linalg.norm(a[r] - b[r]) / (np.linalg.norm(b[r]) + 1e-30):.2e}, v_scales row-relative error "
                              f"{np.linalg.norm(_np(grads[2]).reshape(N, -1)[r] - np.asarray(og[2]).reshape(N, -1)[r]) / (np.linalg.norm(np.asarray(og[2]).reshape(N, -1)[r]) + 1e-30):.2e}")
                    print("    median row-relative error over all rows:", float(np.median(np.linalg.norm(a - b, axis=1) / (np.linalg.norm(b, axis=1) + 1e-30))), "smax", smax)
                    # threshold flips or a defect? A flip is a discontinuity at ONE (pixel, Gaussian) pair: under a 3e-5 relative change of the opacities the pairs that sit
                    # on a threshold are other pairs, so the rows that stand out change; a defect in the arithmetic of a row stays with the row.
                    for eps in (3e-5, -3e-5, 1e-4):
                        op2 = (opacs * (1 + eps)).astype(np.float32)
                        args2 = args[:4] + (t(op2),) + args[5:]
                        rc_p, ra_p, li_p = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args2)
                        g_p = _np(ops.rasterize_to_pixels_from_world_3dgs_bwd(*args2, ra_p, li_p, v_rc, v_ra)[1]).astype(np.float64).reshape(N, -1)
                        oargs2 = oargs[:4] + (op2,) + oargs[5:]
                        b_p = np.asarray(orc.rasterize_bwd(*oargs2, _np(ra_p), _np(li_p), _np(v_rc), _np(v_ra), dtype=np.float64)[1], np.float64).reshape(N, -1)
                        rows_p = np.sqrt(((g_p - b_p) ** 2).sum(1)) / np.sqrt((b_p ** 2).sum())
                        top_p = np.argsort(rows_p)[::-1][:6]
                        print(f"    opacities x (1 {eps:+.0e}): {rows_check(g_p, b_p, bar=2e-4, max_flips=4)}; worst rows", [(int(r), float(f"{rows_p[r]:.2e}")) for r in top_p])
            _stat(f"raster bwd: {nm} rel-L2 vs fp64 oracle without <= max(4, N / 8) flip rows (bar 2e-4)", rest)
            _stat(f"raster bwd: {nm} flip rows", flips)
            if rest > 2e-4:
                # the reference computes in fp32: a case that fp32 itself cannot resolve (huge Gaussians next to the camera ...) is a parity failure only if the kernels ALSO differ from
                # the oracle's fp32 evaluation - the reference's arithmetic in the reference's precision
                k32 = ("v_means", "v_quats", "v_scales", "v_colors", "v_opacities").index(nm)
                o32 = np.asarray(orc.rasterize_bwd(*oargs, _np(ra), _np(li), _np(v_rc), _np(v_ra), dtype=np.float32)[k32], np.float64).reshape(b.shape)
                e32, flips32, rest32 = rows_check(a, o32, bar=2e-4, max_flips=MF)
                _stat(f"raster bwd: {nm} cases beyond the bar against fp64 but within it against the oracle in fp32 (fp32-limited)", rest32)
                # ... or the kernels are as close to the fp64 truth as the reference's arithmetic in fp32 is (twice its distance): flat Gaussians - dL/dscale of the thin axis
                # is ill-conditioned in fp32 for every implementation, two fp32 evaluations of it are further from each other than either is from the truth
                e3264, flips3264, rest3264 = rows_check(o32, b, bar=2e-4, max_flips=MF)
                _stat(f"raster bwd: {nm} fp32-limited cases: (HIP vs fp64) / (oracle fp32 vs fp64)", rest / max(rest3264, 1e-30))
                if not (rest32 <= 2e-4 or rest <= 2 * rest3264):   # where the difference sits: a few rows (threshold flips: one flip moves every Gaussian behind it at that pixel) or all of them
                    rows_e = np.sqrt(((a - b) ** 2).sum(1)) / np.sqrt((b ** 2).sum())
                    top = np.argsort(rows_e)[::-1][:12]
                    print(f"  {nm}: share of the relative L2 error per row, worst 12 of {len(rows_e)}: {[(int(r), float(f'{rows_e[r]:.2e}')) for r in top]}; "
                          f"all other rows together {float(np.sqrt((rows_e ** 2).sum() - (rows_e[top] ** 2).sum())):.2e}", flush=True)
                assert rest32 <= 2e-4 or rest <= 2 * rest3264, (desc, nm, "vs fp64", e, flips, rest, "vs oracle fp32", e32, flips32, rest32, "oracle fp32 vs fp64", rest3264)
    v_col = rng.standard_normal((N, 3)).astype(np.float32)
    v_coeffs, v_dirs = ops.spherical_harmonics_bwd(Kc, deg, t(dirs), t(coeffs), None, t(v_col), True)
    assert torch.isfinite(v_coeffs).all() and torch.isfinite(v_dirs).all(), (desc, "sh_bwd")
    if orc is not None and N:
        o_vc, o_vd = orc.spherical_harmonics_bwd(deg, dirs, coeffs, None, v_col, True)
        e1 = float(np.max(np.abs(_np(v_coeffs) - o_vc) / (1e-5 + 1e-5 * np.abs(o_vc))))
        e2 = float(np.max(np.abs(_np(v_dirs) - o_vd) / (1e-5 * max(1.0, float(np.abs(o_vd).max())) + 1e-4 * np.abs(o_vd))))
        _stat("sh_bwd: v_coeffs |diff| / (atol 1e-5 + rtol 1e-5)", e1)
        _stat("sh_bwd: v_dirs |diff| / (atol 1e-5 max|v_dirs| + rtol 1e-4)", e2)
        assert e1 <= 1.0 and e2 <= 1.0, (desc, "sh_bwd vs oracle", e1, e2)
    if N:
        p, m, v = t(means).clone(), torch.zeros(N, 3, device=DEV), torch.zeros(N, 3, device=DEV)
        ops.adam_step_wrapper(p, m, v, grads[0].reshape(-1, 3)[:N].contiguous(), 1e-3, 0.9, 0.999, 1e-15, 10.0, 31.6)
        assert torch.isfinite(p).all(), (desc, "adam")
        if orc is not None:
            g0 = _np(grads[0].reshape(-1, 3)[:N].contiguous())
            wp, wm, wv = orc.adam_step(means.reshape(-1), np.zeros(3 * N, np.float32), np.zeros(3 * N, np.float32), g0.reshape(-1), 1e-3, 0.9, 0.999, 1e-15, 10.0, 31.6)
            assert np.array_equal(_np(p).reshape(-1), wp) and np.array_equal(_np(m).reshape(-1), wm) and np.array_equal(_np(v).reshape(-1), wv), (desc, "adam bit-exact")
            _stat("adam_step: bit-exact cases", 0)
    return desc, len(ids_n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--replay", default=None, help="a JSON file written by a failed run (the generator state before the failing case): run that one case again")
    ap.add_argument("--oracle", action="store_true", help="differential mode: every stage is compared with the CPU oracle (tests-only code, as in tests/)")
    ap.add_argument("--gpu", action="store_true", help="run the cases on cuda:0 through the shipped gfx950 library instead of the emulated host build")
    ap.add_argument("--flat", type=float, default=0.0, help="probability of a flat-Gaussian case (aspect 10 / 30 / 100); 0 keeps the case stream of the round-4 records")
    ap.add_argument("--cases", type=int, default=0, help="stop after this many cases (0: run for --seconds)")
    ap.add_argument("--keep-going", action="store_true", help="record a failed case (state file + message) and continue with the next one; the exit code says whether any failed")
    ap.add_argument("--state-dir", default="", help="where the generator state of a failed case is written (default /tmp; on a GPU lease: somewhere under gpurun_out/)")
    a = ap.parse_args()
    global DEV, FLAT_P, REPLAY
    FLAT_P = a.flat
    if a.gpu:
        DEV = "cuda:0"
    orc = None
    if a.oracle:
        import oracle as orc
        orc.build(ref=False)
    rng = np.random.default_rng(a.seed)
    import contextlib
    with (contextlib.nullcontext() if a.gpu else emul_util.installed()) as lib:
        import lichtfeld_studio_amd as lfs
        from lichtfeld_studio_amd import ops
        import json
        if a.gpu:
            from lichtfeld_studio_amd import capi
            lib = capi.load_library()
            print("library:", lib.lfs_version().decode())
        if a.replay:
            rng.bit_generator.state = json.load(open(a.replay))["state"]
            # (the case's own index again: the conditioning bars draw their perturbation from a generator seeded with it)
            rec = json.load(open(a.replay))
            m = re.search(r"'idx': (\d+)", rec.get("error", ""))
            REPLAY = True
            print("replayed:", one_case(rng, lfs, ops, lib, int(rec["idx"]) if "idx" in rec else int(m.group(1)) if m else -1, orc))
            for k, (cnt, worst) in sorted(STATS.items()):
                print(f"  {k}: {worst:.3g}")
            return
        t0, n, isects, biggest, failed = time.time(), 0, 0, 0, []
        while time.time() - t0 < a.seconds and (a.cases <= 0 or n < a.cases):
            state = rng.bit_generator.state
            try:
                desc, k = one_case(rng, lfs, ops, lib, n, orc)
            except AssertionError as e:
                sdir = a.state_dir or "/tmp"
                os.makedirs(sdir, exist_ok=True)
                path = os.path.join(sdir, f"fuzz_emulated_seed{a.seed}_case{n}.json")
                json.dump({"state": state, "idx": n, "error": str(e)}, open(path, "w"))
                print(f"case {n} failed; generator state before it written to {path} (--replay {path}{' --oracle' if orc is not None else ''})", flush=True)
                if not a.keep_going:
                    raise
                # --keep-going: the failure is recorded, the case is run again WITHOUT the comparisons (the main generator's draws do not depend on them), the stream goes on
                failed.append((n, str(e)[:300]))
                rng.bit_generator.state = state
                desc, k = one_case(rng, lfs, ops, lib, n, None)
            n += 1
            isects += k
            biggest = max(biggest, k)
        print(f"fuzz_emulated{' --gpu (shipped gfx950 library)' if a.gpu else ''}{f' --flat {a.flat}' if a.flat else ''}: {n} cases in {time.time() - t0:.0f} s (seed {a.seed}), {isects} tile intersections walked in total, largest case {biggest}; "
              f"sanitizer {'ON' if os.environ.get('LFS_EMUL_SANITIZE') else 'off'}; oracle comparison {'ON' if orc is not None else 'off'}; {'no assertion failed' if not failed else f'{len(failed)} case(s) FAILED'}")
        for k, (cnt, worst) in sorted(STATS.items()):
            print(f"  {k}: {cnt} comparisons, worst {worst:.3g}")
        if failed:
            print(f"FAILED cases ({len(failed)}):")
            for n_, e_ in failed:
                print(f"  case {n_}: {e_}")
            sys.exit(1)


if __name__ == "__main__":
    main()
