"""Generate tests/golden/refk_*.npz from the REFERENCE's own device kernels run on the CPU (oracle/_ref/libref_kernels.so: gsplat/*.cu and
the fastgs Adam kernel compiled in place as host code under oracle/ref_emul/, `make -C oracle refk`). Run in the build container, where
/root/reference exists:   python oracle/make_golden_refk.py
These files pin projection_ut_3dgs_fused, rasterize_to_pixels_from_world_3dgs_{fwd,bwd}, relocation, add_noise, quats_to_rotmats and
adam_step - the ops SURVEY.md §8c lists as unpinned by the reference's tests - to the reference itself. The tile lists between projection
and rasterization come from the oracle's intersect_tile, which tests/test_oracle_golden.py pins bit-exactly to the reference's torch_impl."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from oracle import refk_cases as rc  # noqa: E402

out = os.path.join(ROOT, "tests", "golden")
os.makedirs(out, exist_ok=True)
assert oracle.refk_lib() is not None, "build oracle/_ref/libref_kernels.so first (make -C oracle refk)"


def projection(cfg, inp, meta):
    return oracle.refk_projection_ut(inp["means"], inp["quats"], inp["scales"], None if cfg.get("no_opacity") else inp["opacities"], inp["viewmats0"],
                                     inp.get("viewmats1"), inp["Ks"], meta["W"], meta["H"], eps2d=meta["eps2d"], radius_clip=meta["radius_clip"],
                                     calc_compensations=meta["calc_compensations"], camera_model=meta["camera_model"], ut_params=meta["ut_params"],
                                     rs_type=meta["rs_type"], radial_coeffs=inp.get("radial"), tangential_coeffs=inp.get("tangential"),
                                     thin_prism_coeffs=inp.get("thin_prism"))


proj = {}
for name, cfg in rc.PROJECTION_CASES.items():
    inp, meta = rc.build_inputs(cfg), rc.case_meta(cfg)
    radii, m2, d, con, comp = projection(cfg, inp, meta)
    vis = int((radii > 0).all(-1).sum())
    print(f"projection {name}: visible {vis} / {radii.shape[0] * radii.shape[1]}")
    assert vis > 200, "degenerate case"
    for k in ("means", "quats", "scales", "opacities", "viewmats0", "viewmats1", "Ks", "radial", "tangential", "thin_prism"):
        if k in inp:
            proj[f"{name}/{k}"] = inp[k]
    for k, v in meta.items():
        proj[f"{name}/meta_{k}"] = np.asarray(v)
    proj[f"{name}/meta_no_opacity"] = np.asarray(bool(cfg.get("no_opacity", False)))
    proj[f"{name}/radii"], proj[f"{name}/means2d"], proj[f"{name}/depths"], proj[f"{name}/conics"] = radii, m2, d, con
    if comp is not None:
        proj[f"{name}/compensations"] = comp
np.savez_compressed(os.path.join(out, "refk_projection.npz"), **proj)

for name, cfg in rc.RASTER_CASES.items():
    inp, meta = rc.build_inputs(cfg), rc.case_meta(cfg)
    W, H, ts = meta["W"], meta["H"], meta["tile"]
    C = inp["Ks"].shape[0]
    radii, m2, d, con, _ = projection(cfg, inp, meta)
    tw, th = (W + ts - 1) // ts, (H + ts - 1) // ts
    _, ids, flat = oracle.intersect_tile(m2, radii, d, C, ts, tw, th, True)
    offs = oracle.intersect_offset(ids, C, tw, th)
    args = (inp["means"], inp["quats"], inp["scales"], inp["colors"], inp["opacities_cn"], inp.get("backgrounds"), inp.get("masks"), W, H, ts,
            inp["viewmats0"], inp.get("viewmats1"), inp["Ks"], meta["camera_model"], meta["rs_type"], inp.get("radial"), inp.get("tangential"),
            inp.get("thin_prism"), offs, flat)
    render, alpha, last_ids = oracle.refk_rasterize_fwd(*args)
    rng = np.random.default_rng(1000 + cfg["seed"])
    v_render = rng.standard_normal(render.shape).astype(np.float32)
    v_alpha = rng.standard_normal(alpha.shape).astype(np.float32)
    v_means, v_quats, v_scales, v_colors, v_opac = oracle.refk_rasterize_bwd(*args, alpha, last_ids, v_render, v_alpha)
    print(f"raster {name}: I = {len(flat)}, coverage {(alpha > 0.05).mean():.2f}")
    assert (alpha > 0.05).mean() > 0.05, "degenerate case"
    data = {k: v for k, v in inp.items() if k != "opacities"}
    data.update({f"meta_{k}": np.asarray(v) for k, v in meta.items()})
    data.update(radii=radii, means2d=m2, depths=d, offsets=offs, flatten_ids=flat, render=render, alpha=alpha, last_ids=last_ids,
                v_render=v_render.astype(np.float16).astype(np.float32), v_alpha=v_alpha.astype(np.float16).astype(np.float32))
    # (v_render / v_alpha are rounded to fp16-representable values so the files compress; the backward below uses exactly the stored values)
    v_means, v_quats, v_scales, v_colors, v_opac = oracle.refk_rasterize_bwd(*args, alpha, last_ids, data["v_render"], data["v_alpha"])
    data.update(v_means=v_means, v_quats=v_quats, v_scales=v_scales, v_colors=v_colors, v_opacities=v_opac)
    np.savez_compressed(os.path.join(out, f"refk_raster_{name}.npz"), **data)

# ---- small ops ----------------------------------------------------------------------------------------------------------------
rng = np.random.default_rng(77)
N = 1500
binoms = np.zeros((51, 51), np.float32)   # Pascal table in fp32, mcmc.cpp:459-472
for n in range(51):
    for k in range(n + 1):
        binoms[n, k] = 1.0 if k in (0, n) else binoms[n - 1, k - 1] + binoms[n - 1, k]
opac = rng.uniform(0.005, 0.995, N).astype(np.float32)
scales = np.exp(rng.normal(-4, 1, (N, 3))).astype(np.float32)
ratios = rng.integers(1, 52, N).astype(np.int32)
ratios[:51] = np.arange(1, 52)
no, ns = oracle.refk_relocation(opac, scales, ratios, binoms, 51)
raw_o, raw_s, raw_q = (2 * rng.standard_normal(N)).astype(np.float32), rng.normal(-4, 1, (N, 3)).astype(np.float32), rng.standard_normal((N, 4)).astype(np.float32)
noise, means = rng.standard_normal((N, 3)).astype(np.float32), rng.standard_normal((N, 3)).astype(np.float32)
means_out = oracle.refk_add_noise(raw_o, raw_s, raw_q, noise, means, 1.6e-4 * 5e5)
q = rng.standard_normal((512, 4)).astype(np.float32)
q[:4] *= 1e-3
R = oracle.refk_quats_to_rotmats(q)
# Adam: five chained steps with the host-side bias corrections of fused_adam.cpp:78-92 (double, then float)
n = 5003
p0, g = rng.standard_normal(n).astype(np.float32), rng.standard_normal((5, n)).astype(np.float32) * np.float32(1e-3)
g[:, :100] = 0.0
p, m, v = p0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
lr, b1, b2, eps = 1.6e-4, 0.9, 0.999, 1e-15
adam = {}
for t in range(1, 6):
    bc1, bc2 = np.float32(1.0 / (1.0 - b1 ** t)), np.float32(1.0 / np.sqrt(1.0 - b2 ** t))
    p, m, v = oracle.refk_adam_step(p, m, v, g[t - 1], lr, b1, b2, eps, bc1, bc2)
    adam[f"p{t}"], adam[f"m{t}"], adam[f"v{t}"] = p.copy(), m.copy(), v.copy()
np.savez_compressed(os.path.join(out, "refk_small_ops.npz"), binoms=binoms, reloc_opacities=opac, reloc_scales=scales, reloc_ratios=ratios,
                    reloc_new_opacities=no, reloc_new_scales=ns, noise_raw_opacities=raw_o, noise_raw_scales=raw_s, noise_raw_quats=raw_q,
                    noise_noise=noise, noise_means=means, noise_lr=np.float32(1.6e-4 * 5e5), noise_means_out=means_out,
                    quats=q, rotmats=R, adam_p0=p0, adam_grads=g, adam_hyper=np.asarray([lr, b1, b2, eps], np.float64), **adam)
print("reference-kernel golden vectors written:", sorted(f for f in os.listdir(out) if f.startswith("refk_")))
