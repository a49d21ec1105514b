"""GPU parity AT THE HEADLINE SIZES: every stage of the hot path, HIP vs the CPU oracle, on the exact configurations bench.py measures -
SYN-B (BASELINE.json configs[1]: 1M Gaussians, 1920x1080, SH 3, 16x16 tiles, view 0), one SYN-C view (config 4: 3M, 1600x1200) and one SYN-D
view (config 5: 2M, 1080p). The small-scene tests (test_gpu_raster.py, ...) cover the op matrix; these close the "no HIP-vs-oracle comparison
at the size that is benchmarked" gap. The oracle image takes a few seconds on the GPU box's host cores (OpenMP).

Bars (written where they are asserted):
  integer stage  : radii, tiles_per_gauss, isect_ids, flatten_ids, offsets bit-exact           (reference: tests/test_garden_data.cpp:531-569, exact)
  projection     : pinhole / global shutter is bit-exact (projection_ut.hip is built -ffp-contract=off)
  forward        : mean |rgb| <= 2e-6 (reference bar: mean abs < 1e-4, tests/test_rasterization.cpp:352-353), <= 1e-5 of the pixels beyond
                   1/255 + 1e-4 (one alpha-threshold or early-termination flip), last_ids equal on >= 99.95 %
  backward       : per-tensor relative L2 <= 1e-3 vs the oracle (tests/test_numerical_gradients.cpp:59-81 uses 1e-3), and the
                   number of "flip rows" (Gaussians with an alpha that crossed 1/255 or the 0.999 cap at one pixel, v_exp_f32 vs exp:
                   gpu_util.rows_check) is counted and bounded explicitly (<= 4 of 1-3 M) instead of being dropped by fraction;
                   measured: SYN-B 0 flips and 2e-5 .. 8e-5, SYN-C 1 flip (quats: 1.9e-3 with it, 1.9e-4 without), SYN-D 0 flips and <= 1.5e-4;
                   SYN-B-flat (aspect up to 100): printed by the run - the library before LFS_BWD_REORTH fails this bar on v_means / v_quats there
  raw-parameter gradients of the whole step (activations + SH + clamp + loss): relative L2 <= 1e-3
"""
import numpy as np
import pytest
import torch

from gpu_util import n, rel_l2, t
from gpu_util import rows_check as _rows_check

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
NAMES = ["means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opacities"]


MAX_FLIP_ROWS = 4
# Gaussians per scene whose integer radius differs from the oracle's by 1 (GPU logf / sqrtf vs the host's within an ulp of an integer): measured 0 / 0 / 0 / 1 in
# round 5 on the MI355X, pinned as upper bounds (+1 of slack for another box's libm-equivalent rounding on the flat scene only)
MAX_RADIUS_OFF = {"syn_b": 0, "syn_c": 0, "syn_d": 0, "syn_b_flat": 2}


SCENES = {
    "syn_b": dict(maker="syn_b", view=0),
    "syn_c": dict(maker="syn_c", view=5),
    "syn_d": dict(maker="syn_d", view=9),
    # round 5: SYN-B made of flat disks (aspect ratio log-uniform in 1 .. 100: scenes.syn_b_flat) - the regime K8's re-orthogonalisation step exists for; no parity
    # scene of rounds 1 - 4 had a Gaussian flatter than 6 : 1. Same bars as the other three.
    "syn_b_flat": dict(maker="syn_b_flat", view=0),
}


@pytest.fixture(scope="module", params=list(SCENES))
def case(request, lfs, oracle_mod):
    from oracle import pipeline
    from lichtfeld_studio_amd import scenes
    cfg = SCENES[request.param]
    sc = getattr(scenes, cfg["maker"])()
    target = scenes.target_image(sc.height, sc.width, seed=43)
    sa = pipeline.scene_arrays(sc)
    orc = pipeline.train_image(sa, cfg["view"], target.numpy())
    print(f"[{request.param}] oracle image: {orc['seconds']:.1f} s, V={int(orc['visible'].sum())}, I={len(orc['flatten_ids'])}")
    return dict(name=request.param, sc=sc, view=cfg["view"], target=target, orc=orc)


def test_projection_and_integer_stage_bit_exact(lfs, case):
    from lichtfeld_studio_amd import ops
    sc, v, o = case["sc"], case["view"], case["orc"]
    W, H = sc.width, sc.height
    vm, K = sc.viewmats[v:v + 1].contiguous().to(DEV), sc.Ks[v:v + 1].contiguous().to(DEV)
    radii, m2, d, con, _ = ops.projection_ut_3dgs_fused(sc.means.to(DEV), t(o["quats"]), t(o["scales"]), t(o["opacities"]), vm, None, K, W, H,
                                                        0.3, 0.01, 1e4, 0.0, False, lfs.CameraModelType.PINHOLE)
    # radii: ceil(min(extend sqrt(c00), r1)) with extend = sqrt(2 log(255 opacity)) - the GPU's logf / sqrtf and the host's differ by an ulp now and then, and once in
    # ~1e6 Gaussians the product sits within that ulp of an integer ("radii +-1"). Counted, never more than 3 per scene, never more than +-1; measured:
    # 0 on SYN-B / C / D, 1 on SYN-B-flat. Everything else of the projection is bit-exact (built -ffp-contract=off).
    r_gpu, r_ref = n(radii), o["radii"]
    off = (r_gpu != r_ref).any(-1)
    print(f"[{case['name']}] projection: {int(off.sum())} of {r_ref.shape[1]} Gaussians with a radius that differs (by at most {int(np.abs(r_gpu - r_ref).max())})")
    assert int(off.sum()) <= 3 and int(np.abs(r_gpu - r_ref).max()) <= 1, (int(off.sum()), int(np.abs(r_gpu - r_ref).max()))
    #  the measured counts per scene are pinned as upper bounds: a regression that produces "only" three wrong radii per scene does not slip through
    assert int(off.sum()) <= MAX_RADIUS_OFF[case["name"]], (case["name"], int(off.sum()), MAX_RADIUS_OFF[case["name"]])
    assert np.array_equal((r_gpu > 0).all(-1), (r_ref > 0).all(-1))     # the visibility set itself is identical
    vis = o["visible"]
    assert np.array_equal(n(m2)[0][vis], o["means2d"][0][vis]) and np.array_equal(n(d)[0][vis], o["depths"][0][vis])
    assert np.array_equal(n(con)[0][vis], o["conics"][0][vis])
    tw, th = (W + 15) // 16, (H + 15) // 16
    # the integer stage on IDENTICAL inputs (the oracle's radii where the two differ): bit-exact
    radii_in = radii if not off.any() else t(r_ref, torch.int32)
    tpg, ids, flat, offs = ops.intersect_tile(m2, radii_in, d, None, None, 1, 16, tw, th, True, return_offsets=True)
    assert np.array_equal(n(tpg), o["tiles_per_gauss"])
    assert ids.shape[0] == len(o["isect_ids"]) and np.array_equal(n(ids), o["isect_ids"]) and np.array_equal(n(flat), o["flatten_ids"])
    assert np.array_equal(n(offs), o["offsets"]) and np.array_equal(n(ops.intersect_offset(ids, 1, tw, th)), o["offsets"])
    if off.any():
        #  and the integer stage on the GPU's OWN radii: tiles_per_gauss may differ from the oracle's only for the flagged Gaussians, and only by what
        # one more / one fewer pixel of radius explains - at most one more tile row and one more tile column: (w + 1)(h + 1) - w h = w + h + 1 tiles
        tpg_own = n(ops.intersect_tile(m2, radii, d, None, None, 1, 16, tw, th, True)[0])[0]
        ref_t = o["tiles_per_gauss"][0]
        changed = tpg_own != ref_t
        assert not (changed & ~off[0]).any(), "tiles_per_gauss differs for a Gaussian whose radii agree"
        r_max = np.maximum(r_gpu, r_ref)[0]
        span = (np.ceil((2.0 * r_max[:, 0] + 1) / 16) + 1) + (np.ceil((2.0 * r_max[:, 1] + 1) / 16) + 1) + 1
        assert (np.abs(tpg_own.astype(np.int64) - ref_t)[changed] <= span[changed]).all(), (tpg_own[changed], ref_t[changed], span[changed])
        print(f"[{case['name']}] integer stage on the GPU's own radii: {int(changed.sum())} of the {int(off.sum())} flagged Gaussians cover a different number of tiles")


def test_forward_and_backward_against_oracle(lfs, case):
    from lichtfeld_studio_amd import fused, ops
    sc, v, o = case["sc"], case["view"], case["orc"]
    W, H = sc.width, sc.height
    vm, K = sc.viewmats[v:v + 1].contiguous().to(DEV), sc.Ks[v:v + 1].contiguous().to(DEV)
    means = sc.means.to(DEV)
    colors = fused.sh_model_fwd(sc.sh_degree, means, vm, sc.sh0.to(DEV), sc.shN.to(DEV), t(o["radii"], torch.int32))
    dc = np.abs(n(colors) - o["colors"])[o["visible"]]
    assert dc.max() < 1e-5, dc.max()          # (reference bar for SH: 1e-4, tests/test_numerical_gradients.cpp:440-460)
    args = (means, t(o["quats"]), t(o["scales"]), t(o["colors"][None]), t(o["opacities"][None]), torch.zeros(1, 3, device=DEV), None, W, H, 16, vm, None, K,
            lfs.CameraModelType.PINHOLE, None, lfs.ShutterType.GLOBAL, None, None, None, t(o["offsets"], torch.int32), t(o["flatten_ids"], torch.int32))
    rc, ra, li = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)
    d = np.abs(n(rc) - o["render"])
    da = np.abs(n(ra) - o["alpha"])
    beyond = float((d.max(-1) > 1 / 255 + 1e-4).mean())
    same_last = float((n(li) == o["last_ids"]).mean())
    print(f"[{case['name']}] fwd: mean|rgb| {d.mean():.2e} max {d.max():.2e}, px beyond 1/255 {beyond:.2e}, alpha mean {da.mean():.2e}, last_ids equal {same_last:.6f}")
    assert d.mean() < 2e-6 and da.mean() < 2e-6
    assert beyond < 1e-5 and same_last > 0.9995   # (SURVEY.md §8c proposes 99.9 %; measured 99.98 %: an alpha within an ulp of 1/255 at the LAST contributor)
    # backward on the ORACLE's forward outputs and loss gradient, so both sides differentiate the same function values
    gg = ops.rasterize_to_pixels_from_world_3dgs_bwd(*args, t(o["alpha"]), t(o["last_ids"], torch.int32), t(o["v_render"]), None)
    ref = [o["v_means_raster"], o["v_quats"], o["v_scales"], o["v_colors"][None], o["v_opacities"][None]]
    for name, a, b in zip(["v_means", "v_quats", "v_scales", "v_colors", "v_opacities"], gg, ref):
        e, flips, e_rest = _rows_check(n(a).reshape(sc.N, -1), b.reshape(sc.N, -1), bar=1e-3, max_flips=MAX_FLIP_ROWS)
        print(f"[{case['name']}] bwd {name}: rel-L2 {e:.2e}, flip rows {flips}, rel-L2 without them {e_rest:.2e}")
        assert np.isfinite(n(a)).all(), name
        assert flips <= MAX_FLIP_ROWS and e_rest < 1e-3, (name, e, flips, e_rest)


def test_full_step_raw_parameter_gradients(lfs, case):
    """the fused training step itself (what bench.py times): loss, image, n_isects and the six raw-parameter gradients"""
    from lichtfeld_studio_amd.fused import render_and_backward
    from lichtfeld_studio_amd.trainer import GutTrainer
    sc, v, o = case["sc"], case["view"], case["orc"]
    tr = GutTrainer(sc, DEV, iterations=7000)
    params = tr.model.parameters()
    grads = [torch.full_like(p, 3.0) for p in params]
    loss = torch.zeros(1, device=DEV)
    out = render_and_backward(tr.camera(v), tr.model, tr.bg, case["target"].to(DEV), 1.0, grads, loss, accumulate=False)
    # (the step computes normalize / exp / sigmoid on the GPU, the oracle with numpy: an ulp on a scale moves a radius across an integer for a
    #  couple of Gaussians in a million; with identical activated inputs the integer stage is bit-exact - first test of this file)
    assert abs(out.n_isects - len(o["flatten_ids"])) <= 16, (out.n_isects, len(o["flatten_ids"]))
    assert abs(float(loss) - o["loss"]) < 2e-6 * max(1.0, o["loss"]), (float(loss), o["loss"])
    d = np.abs(n(out.image_hwc) - o["render"])
    assert d.mean() < 2e-6, d.mean()
    res = {name: _rows_check(n(g).reshape(sc.N, -1), o["grads"][name].reshape(sc.N, -1), bar=1e-3, max_flips=MAX_FLIP_ROWS) for name, g in zip(NAMES, grads)}
    print(f"[{case['name']}] step: loss {float(loss):.8f} vs {o['loss']:.8f}; raw-grad (rel-L2, flip rows, rel-L2 without them): "
          + ", ".join(f"{k} ({v[0]:.1e}, {v[1]}, {v[2]:.1e})" for k, v in res.items()))
    for name, (e, flips, e_rest) in res.items():
        assert flips <= MAX_FLIP_ROWS and e_rest < 1e-3, (name, e, flips, e_rest)
