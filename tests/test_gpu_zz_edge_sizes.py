"""GPU: the edge sizes of SURVEY.md §8(b) "Errors" for the operators that had no such case yet - empty problems skip the launch and hand back empty tensors
(reference: ProjectionUT3DGSFused.cu:242-245, RasterizeToPixelsFromWorld3DGSBwd.cu:434-437, IntersectTile.cu:268-271 do the same for their operators; those three are
covered in test_gpu_projection_sh.py / test_gpu_raster.py / test_gpu_intersect.py) - and ONE-element problems against the oracle (a single lane of a single
wavefront live: the tail handling of every streaming kernel). The file sorts last on purpose: these are additions of the round's last session, run on the
emulated library in the CPU suite (tests/test_emulated_gpu_suite.py) but not yet on a leased MI355X - a surprise here must not cost the `-x` run anything else."""
import math

import numpy as np
import pytest
import torch

from gpu_util import n, t

pytestmark = pytest.mark.gpu


def _binoms():
    b = np.zeros((51, 51), np.float32)   # mcmc.cpp:459-472
    for a in range(51):
        for k in range(a + 1):
            b[a, k] = math.comb(a, k)
    return b


def _e(*shape, dtype=torch.float32):
    return t(np.zeros(shape, np.float32), dtype)


@pytest.mark.parametrize("degree,K", [(0, 1), (3, 16), (4, 25)])
def test_sh_empty(lfs, degree, K):
    from lichtfeld_studio_amd import ops
    colors = ops.spherical_harmonics_fwd(degree, _e(0, 3), _e(0, K, 3), None)
    assert tuple(colors.shape) == (0, 3)
    v_coeffs, v_dirs = ops.spherical_harmonics_bwd(K, degree, _e(0, 3), _e(0, K, 3), None, _e(0, 3), True)
    assert tuple(v_coeffs.shape) == (0, K, 3) and tuple(v_dirs.shape) == (0, 3)
    v_coeffs, v_dirs = ops.spherical_harmonics_bwd(K, degree, _e(0, 3), _e(0, K, 3), t(np.zeros(0, bool), torch.bool), _e(0, 3), False)
    assert tuple(v_coeffs.shape) == (0, K, 3) and v_dirs is None


@pytest.mark.parametrize("degree", [0, 1, 2, 3, 4])
def test_sh_single_row_matches_oracle(lfs, oracle_mod, degree):
    from lichtfeld_studio_amd import ops
    rng = np.random.default_rng(40 + degree)
    K = 25
    d = rng.standard_normal((1, 3)).astype(np.float32)
    c = rng.standard_normal((1, K, 3)).astype(np.float32)
    vc = rng.standard_normal((1, 3)).astype(np.float32)
    want = oracle_mod.spherical_harmonics_fwd(degree, d, c, None)
    np.testing.assert_allclose(n(ops.spherical_harmonics_fwd(degree, t(d), t(c), None)), want, rtol=1e-5, atol=1e-5)   # the bars of test_gpu_projection_sh.py::test_spherical_harmonics_fwd_bwd
    w_coeffs, w_dirs = oracle_mod.spherical_harmonics_bwd(degree, d, c, None, vc, True)
    g_coeffs, g_dirs = ops.spherical_harmonics_bwd(K, degree, t(d), t(c), None, t(vc), True)
    np.testing.assert_allclose(n(g_coeffs), w_coeffs, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(n(g_dirs), w_dirs, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(w_dirs).max())))
    # a masked-out single row: colour 0, gradients 0
    off = t(np.zeros(1, bool), torch.bool)
    assert float(ops.spherical_harmonics_fwd(degree, t(d), t(c), off).abs().max()) == 0.0
    g_coeffs, g_dirs = ops.spherical_harmonics_bwd(K, degree, t(d), t(c), off, t(vc), True)
    assert float(g_coeffs.abs().max()) == 0.0 and float(g_dirs.abs().max()) == 0.0


def test_small_ops_empty(lfs):
    from lichtfeld_studio_amd import ops
    assert tuple(ops.quats_to_rotmats(_e(0, 4)).shape) == (0, 3, 3)
    no, ns = ops.relocation(_e(0), _e(0, 3), t(np.zeros(0, np.int32), torch.int32), t(_binoms()), 51)
    assert tuple(no.shape) == (0,) and tuple(ns.shape) == (0, 3)
    means = _e(0, 3)
    assert ops.add_noise(_e(0), _e(0, 3), _e(0, 4), _e(0, 3), means, 80.0) is None and tuple(means.shape) == (0, 3)
    p = _e(0)
    ops.adam_step_wrapper(p, _e(0), _e(0), _e(0), 1e-3, 0.9, 0.999, 1e-15, 10.0, 31.6)
    assert p.numel() == 0


def test_small_ops_single_element_match_oracle(lfs, oracle_mod):
    from lichtfeld_studio_amd import ops
    rng = np.random.default_rng(77)
    q = rng.standard_normal((1, 4)).astype(np.float32)
    np.testing.assert_allclose(n(ops.quats_to_rotmats(t(q))), oracle_mod.quats_to_rotmats(q), rtol=0, atol=1e-6)
    o = np.array([0.37], np.float32)
    s = np.array([[0.02, 0.05, 0.01]], np.float32)
    r = np.array([3], np.int32)
    wo, ws = oracle_mod.relocation(o, s, r, _binoms(), 51)
    go, gs = ops.relocation(t(o), t(s), t(r, torch.int32), t(_binoms()), 51)
    np.testing.assert_allclose(n(go), wo, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(n(gs), ws, rtol=2e-3)
    ro, rs, rq = np.array([0.3], np.float32), np.log(np.array([[0.02, 0.03, 0.01]], np.float32)), rng.standard_normal((1, 4)).astype(np.float32)
    nz, m = rng.standard_normal((1, 3)).astype(np.float32), rng.standard_normal((1, 3)).astype(np.float32)
    want = oracle_mod.add_noise(ro, rs, rq, nz, m, 80.0)
    got = t(m).clone()
    ops.add_noise(t(ro), t(rs), t(rq), t(nz), got, 80.0)
    np.testing.assert_allclose(n(got), want, rtol=1e-5, atol=1e-6 * max(1.0, float(np.abs(want - m).max())))


def test_adam_multi_with_empty_members_is_bit_exact(lfs, oracle_mod):
    """the fused six-group launch (FusedAdam::step, fused_adam.cpp:22-95) when some groups have no elements (a model before shN exists: K = 1) - the empty members
    take no workgroup and the others come out as from adam_kernels.cuh:13-36, bit for bit"""
    from lichtfeld_studio_amd import ops
    rng = np.random.default_rng(5)
    args = (1.6e-4, 0.9, 0.999, 1e-15, 1 / (1 - 0.9 ** 3), 1 / math.sqrt(1 - 0.999 ** 3))
    sizes = [3 * 777, 0, 1, 0, 4 * 777]
    host, dev = [], []
    for sz in sizes:
        mk = lambda s=1.0: (rng.standard_normal(sz) * s).astype(np.float32)
        p, m, v, g = mk(), mk(0.1), np.abs(mk(0.01)), mk()
        host.append((p, m, v, g))
        dev.append((t(p), t(m), t(v), t(g)))
    ops.adam_step_multi([(p, m, v, g, *args) for (p, m, v, g) in dev])
    for (p, m, v, g), (gp, gm, gv, _) in zip(host, dev):
        if p.size == 0:
            assert gp.numel() == 0
            continue
        wp, wm, wv = oracle_mod.adam_step(p, m, v, g, *args)
        assert np.array_equal(n(gp), wp) and np.array_equal(n(gm), wm) and np.array_equal(n(gv), wv)
    ops.adam_step_multi([])   # nothing to do: no launch, no error


def _fastgs_run(sc, lfs_oracle):
    """forward + backward of the fastgs (EWA) rasterizer on `sc`, HIP and fp32 / fp64 oracle (as tests/test_gpu_fastgs.py does it)"""
    from lichtfeld_studio_amd import fastgs
    from test_gpu_fastgs import _dev_args, _oracle_fwd, _settings
    s = _settings(sc)
    image, alpha, pws, iws, n_inst = fastgs.forward_wrapper(*_dev_args(sc), s)
    f32 = _oracle_fwd(lfs_oracle, sc, np.float32)
    rng = np.random.default_rng(5)
    gi, ga = rng.standard_normal((3, sc["H"], sc["W"])).astype(np.float32), rng.standard_normal((1, sc["H"], sc["W"])).astype(np.float32)
    N = sc["means"].shape[0]
    dens = torch.zeros(2, N, device=image.device)
    g = fastgs.backward_wrapper(dens, t(gi), t(ga), image, alpha, *[t(sc[k]) for k in ("means", "scales_raw", "rot_raw", "sh0", "sh_rest")], pws, iws, t(sc["w2c"]), s, n_inst)
    f64 = _oracle_fwd(lfs_oracle, sc, np.float64)
    og = lfs_oracle.fastgs_backward(f64, sc["means"], sc["scales_raw"], sc["rot_raw"], sc["opac_raw"], sc["sh0"], sc["sh_rest"], sc["w2c"], sc["cam_pos"],
                                    sc["active_sh_bases"], sc["W"], sc["H"], sc["fx"], sc["fy"], sc["cx"], sc["cy"], gi, ga, dtype=np.float64)
    return image, alpha, n_inst, f32, g, og, dens


def test_fastgs_nothing_in_front_of_the_camera(lfs, oracle_mod):
    """every primitive behind the near plane: no instance, an empty image, exact-zero gradients (the reference returns before its sorts when
    n_instances == 0 and leaves the zero-initialised outputs, fastgs/rasterization/src/forward.cu)"""
    from test_oracle_fastgs import _scene
    sc = _scene(N=300, W=70, H=50, seed=9, deg=1)
    sc["means"][:, 2] = -np.abs(sc["means"][:, 2]) - 3.0
    image, alpha, n_inst, f32, g, og, dens = _fastgs_run(sc, oracle_mod)
    assert n_inst == 0 and len(f32["ids"]) == 0
    assert float(image.abs().max()) == 0.0 and float(alpha.abs().max()) == 0.0
    for a, b in zip(g, og[:6]):
        assert float(a.abs().max()) == 0.0 and float(np.abs(b).max()) == 0.0
    assert float(dens.abs().max()) == 0.0


def test_fastgs_single_primitive_matches_oracle(lfs, oracle_mod):
    from test_oracle_fastgs import _scene
    sc = _scene(N=1, W=70, H=50, seed=10, deg=3)
    sc["means"][:] = [[0.05, -0.02, 3.0]]
    sc["opac_raw"][:] = 2.0
    image, alpha, n_inst, f32, g, og, dens = _fastgs_run(sc, oracle_mod)
    assert n_inst == len(f32["ids"]) and n_inst > 0
    assert f32["alpha"].max() > 0.3, "degenerate scene"
    d = np.abs(n(image) - f32["image"])
    assert d.mean() < 2e-6 and (d > 1 / 255 + 1e-4).mean() < 1e-3, (d.mean(), d.max())
    for name, a, b in zip(["means", "scales_raw", "rot_raw", "opac_raw", "sh0", "sh_rest"], g, og[:6]):
        a = n(a).reshape(b.shape)
        assert np.isfinite(a).all(), name
        if np.abs(b).max() == 0:
            assert np.abs(a).max() == 0, name
        else:
            e = float(np.linalg.norm(a - b) / np.linalg.norm(b))
            print(f"fastgs single primitive, {name}: rel-L2 {e:.2e}")
            assert e < 5e-4, (name, e)
    assert np.array_equal(n(dens)[0], og[6][0].astype(np.float32))


def test_mcmc_relocation_without_dead_gaussians_and_growth_at_the_cap_touch_nothing(lfs):
    """mcmc.cpp:113-194 with an empty dead set (`dead_indices.numel() == 0`: return) and mcmc.cpp:196-340 at `max_cap` (n_new == 0: return): every parameter and
    every Adam moment keeps its bits; the noise injection that follows still moves the means"""
    from lichtfeld_studio_amd import strategies
    from test_gpu_strategies import _model, _prime_optimizer
    model, dead = _model(N=1500, dead_frac=0.0)
    assert int(dead.sum()) == 0
    st = strategies.MCMC(model, strategies.OptimizationParameters(max_cap=1500), generator=torch.Generator(device=model.means.device).manual_seed(3))
    _prime_optimizer(st)
    before = {k: n(getattr(model, k)).copy() for k in strategies._PARAM_NAMES}
    moments = {k: (n(st.optimizer.state[id(getattr(model, k))]["exp_avg"]).copy(), n(st.optimizer.state[id(getattr(model, k))]["exp_avg_sq"]).copy())
               for k in strategies._PARAM_NAMES}
    assert int(st.relocate_gs()) == 0
    assert st.add_new_gs() == 0 and model.means.shape[0] == 1500
    for k in strategies._PARAM_NAMES:
        assert np.array_equal(n(getattr(model, k)), before[k]), k
        s = st.optimizer.state[id(getattr(model, k))]
        assert np.array_equal(n(s["exp_avg"]), moments[k][0]) and np.array_equal(n(s["exp_avg_sq"]), moments[k][1]), k
    st.inject_noise()
    assert not np.array_equal(n(model.means), before["means"])
    for k in strategies._PARAM_NAMES:
        if k != "means":
            assert np.array_equal(n(getattr(model, k)), before[k]), k


@pytest.mark.parametrize("fused", [False, True])
def test_adc_refinement_with_nothing_to_grow_or_prune_touches_nothing(lfs, fused):
    """default_strategy.cpp:162-195 / :229-249 when no Gaussian passes the gradient threshold and none is below the pruning opacity (empty duplicate / split / prune
    sets): the model, the Adam moments and the Gaussian count keep their bits - in the reference's sequence (grow_gs + prune_gs) and in the fused device-side form"""
    from lichtfeld_studio_amd import strategies
    from test_gpu_strategies import _model, _prime_optimizer
    N = 1200
    model, _ = _model(N=N, dead_frac=0.0)
    dev = model.means.device
    p = strategies.OptimizationParameters(grow_scale3d=0.035, prune_opacity=0.005, reset_every=300)
    st = strategies.DefaultStrategy(model, p, generator=torch.Generator(device=dev).manual_seed(2))
    _prime_optimizer(st)
    assert float(model.get_opacity().detach().min()) > 0.005
    before = {k: n(getattr(model, k)).copy() for k in strategies._PARAM_NAMES}
    moments = {k: n(st.optimizer.state[id(getattr(model, k))]["exp_avg"]).copy() for k in strategies._PARAM_NAMES}
    info = torch.zeros(2, N, device=dev)
    info[0] = 4.0
    info[1] = 4.0 * 1e-5                                   # average gradient 1e-5 < grad_threshold 2e-4 everywhere
    rnd = torch.randn(2, N, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(9))
    if fused:
        st.grow_and_prune_fused(700, info, rnd)
    else:
        st.grow_gs(700, info, rnd)
        st.prune_gs(700)
    assert model.means.shape[0] == N
    for k in strategies._PARAM_NAMES:
        assert np.array_equal(n(getattr(model, k)), before[k]), k
        assert np.array_equal(n(st.optimizer.state[id(getattr(model, k))]["exp_avg"]), moments[k]), k
