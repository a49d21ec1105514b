"""CPU: the fastgs (EWA) oracle (oracle/oracle_fastgs.hpp) — pinned where the reference has CPU code of its own
(tests/torch_impl.cpp: EWA projection and SH evaluation, compiled in place into oracle/_ref), validated by internal
consistency elsewhere (analytic backward vs central finite differences of the fp64 forward; fp32 vs fp64)."""
import math

import numpy as np
import pytest


def _scene(N=400, W=80, H=64, seed=0, deg=2, spread=1.2):
    rng = np.random.default_rng(seed)
    means = rng.standard_normal((N, 3)) * spread
    means[:, 2] = np.abs(means[:, 2]) + 3.0
    scales_raw = np.log(rng.uniform(0.03, 0.25, (N, 3)))
    rot_raw = rng.standard_normal((N, 4))
    opac_raw = rng.standard_normal(N) * 1.5
    K = (deg + 1) ** 2
    sh0 = rng.standard_normal((N, 1, 3)) * 0.5
    sh_rest = rng.standard_normal((N, 15, 3)) * 0.2       # storage for degree 3, `deg` of them active
    ang = 0.1
    R = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
    w2c = np.eye(4); w2c[:3, :3] = R; w2c[:3, 3] = [0.1, -0.05, 0.2]
    cam_pos = -R.T @ w2c[:3, 3]
    intr = dict(W=W, H=H, fx=70.0, fy=75.0, cx=W / 2 + 0.3, cy=H / 2 - 0.2)
    return dict(means=means, scales_raw=scales_raw, rot_raw=rot_raw, opac_raw=opac_raw, sh0=sh0, sh_rest=sh_rest, w2c=w2c, cam_pos=cam_pos,
                active_sh_bases=K, **intr)


def _fwd(o, sc, dtype):
    return o.fastgs_forward(sc["means"], sc["scales_raw"], sc["rot_raw"], sc["opac_raw"], sc["sh0"], sc["sh_rest"], sc["w2c"], sc["cam_pos"],
                            sc["active_sh_bases"], sc["W"], sc["H"], sc["fx"], sc["fy"], sc["cx"], sc["cy"], dtype=dtype)


def test_fastgs_oracle_projection_and_colour_match_the_reference_cpu_code(oracle_mod):
    o = oracle_mod
    if o.ref_lib() is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    sc = _scene(N=600, W=256, H=256, seed=1, deg=3)
    sc.update(fx=200.0, fy=200.0, cx=128.0, cy=128.0)
    f = _fwd(o, sc, np.float32)
    quats = sc["rot_raw"] / np.linalg.norm(sc["rot_raw"], axis=1, keepdims=True)
    K = np.array([[200.0, 0, 128], [0, 200.0, 128], [0, 0, 1]])
    radii, m2, d, con = o.ref_fully_fused_projection(sc["means"], quats, np.exp(sc["scales_raw"]), sc["w2c"], K, 256, 256, eps2d=0.3)
    vis = f["n_touched"] > 0
    # compare where the fastgs Jacobian clamp (1.15 x the image) is inactive and the reference kept the Gaussian too
    inside = (f["mean2d"][:, 0] > -20) & (f["mean2d"][:, 0] < 276) & (f["mean2d"][:, 1] > -20) & (f["mean2d"][:, 1] < 276)
    sel = vis & inside & (radii > 0).all(-1)
    assert sel.sum() > 100
    np.testing.assert_allclose(f["mean2d"][sel], m2[sel], rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(f["conic_opacity"][sel, :3], con[sel], rtol=2e-3, atol=1e-5)
    col_ref = o.ref_spherical_harmonics(3, sc["means"] - sc["cam_pos"], np.concatenate([sc["sh0"], sc["sh_rest"]], 1)) + 0.5
    np.testing.assert_allclose(f["color"][vis], col_ref[vis], rtol=1e-4, atol=1e-5)


def test_fastgs_oracle_lists_are_sorted_and_exact(oracle_mod):
    sc = _scene(N=500, seed=2)
    f = _fwd(oracle_mod, sc, np.float32)
    gw = (sc["W"] + 15) // 16
    offs, ids = f["offsets"], f["ids"]
    assert offs[-1] == len(ids) == f["n_touched"].sum()
    depth = (sc["means"] @ sc["w2c"][2, :3] + sc["w2c"][2, 3]).astype(np.float32)
    for t in range(len(offs) - 1):
        seg = ids[offs[t]:offs[t + 1]]
        key = [(depth[i].view(np.uint32), i) for i in seg]
        assert key == sorted(key)
        ty, tx = divmod(t, gw)
        b = f["bounds"][seg]
        assert ((b[:, 0] <= tx) & (tx < b[:, 1]) & (b[:, 2] <= ty) & (ty < b[:, 3])).all()
    assert (f["alpha"] >= 0).all() and (f["alpha"] <= 1).all() and f["alpha"].max() > 0.5


def test_fastgs_oracle_backward_matches_finite_differences(oracle_mod):
    o = oracle_mod
    sc = _scene(N=120, W=48, H=40, seed=3, deg=2, spread=0.8)
    rng = np.random.default_rng(9)
    gi, ga = rng.standard_normal((3, sc["H"], sc["W"])), rng.standard_normal((sc["H"], sc["W"]))
    f = _fwd(o, sc, np.float64)
    g = o.fastgs_backward(f, sc["means"], sc["scales_raw"], sc["rot_raw"], sc["opac_raw"], sc["sh0"], sc["sh_rest"], sc["w2c"], sc["cam_pos"],
                          sc["active_sh_bases"], sc["W"], sc["H"], sc["fx"], sc["fy"], sc["cx"], sc["cy"], gi, ga, dtype=np.float64)
    names = ["means", "scales_raw", "rot_raw", "opac_raw", "sh0", "sh_rest"]

    def loss(s):
        ff = _fwd(o, s, np.float64)
        return float((ff["image"] * gi).sum() + (ff["alpha"] * ga).sum())

    ok = total = 0
    eps = 1e-6
    for k, (name, grad) in enumerate(zip(names, g[:6])):
        flat = np.flatnonzero(np.abs(grad.reshape(-1)) > 1e-6)
        for idx in rng.choice(flat, min(10, len(flat)), replace=False):
            sp, sm = dict(sc), dict(sc)
            a = sc[name].astype(np.float64).copy().reshape(-1); a[idx] += eps; sp[name] = a.reshape(sc[name].shape)
            b = sc[name].astype(np.float64).copy().reshape(-1); b[idx] -= eps; sm[name] = b.reshape(sc[name].shape)
            fd = (loss(sp) - loss(sm)) / (2 * eps)
            an = grad.reshape(-1)[idx]
            total += 1
            ok += abs(fd - an) <= 1e-3 * max(1.0, abs(an))       # (alpha-threshold / termination flips make a few probes jump)
    assert total >= 50 and ok >= 0.9 * total, (ok, total)
    dens = g[6]
    assert dens.shape == (2, 120) and (dens[0] == (f["n_touched"] > 0)).all() and (dens[1] >= 0).all()


def test_fastgs_oracle_fp32_tracks_fp64(oracle_mod):
    sc = _scene(N=300, seed=4)
    a, b = _fwd(oracle_mod, sc, np.float32), _fwd(oracle_mod, sc, np.float64)
    d = np.abs(a["image"] - b["image"])
    assert d.mean() < 1e-5 and (d > 1 / 255 + 1e-4).mean() < 1e-3
    assert (a["n_touched"] != b["n_touched"]).mean() < 0.01
