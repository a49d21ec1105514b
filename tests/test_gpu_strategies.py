"""GPU: the strategy layer (mirrors of mcmc.cpp / default_strategy.cpp / strategy_utils.cpp) over the
★ MCMC kernels. The kernels themselves are pinned against the oracle in test_gpu_small_ops.py; here: the host logic —
which Gaussians are touched, how parameters and Adam state are rewritten — against a numpy restatement of the same rules
(the oracle's relocation op for the values), determinism across replicas, and that MCMC training converges."""
import math

import numpy as np
import pytest
import torch

from gpu_util import n

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _model(N=4000, seed=0, sh_degree=1, dead_frac=0.1):
    from lichtfeld_studio_amd import scenes
    from lichtfeld_studio_amd.rasterizer import SplatModel
    sc = scenes.syn_a(seed=seed, n=N, sh_degree=sh_degree).to(DEV)
    g = torch.Generator().manual_seed(seed + 1)
    raw_o = sc.raw_opacities.clone()
    dead = torch.rand(N, generator=g) < dead_frac
    raw_o[dead.to(DEV)] = -8.0          # sigmoid(-8) = 3e-4 < min_opacity
    mk = lambda t: t.clone().contiguous().requires_grad_(True)
    return SplatModel(mk(sc.means), mk(sc.sh0), mk(sc.shN), mk(sc.raw_scales), mk(sc.raw_quats), mk(raw_o), sh_degree), dead


def _prime_optimizer(strategy):
    """one Adam step with random gradients so that every parameter has non-zero state"""
    g = torch.Generator(device=DEV).manual_seed(5)
    for p in strategy.model.parameters():
        p.grad = torch.randn(p.shape, device=DEV, generator=g)
    strategy.optimizer.step(2000)   # > 1000: the shN group is stepped too
    strategy.optimizer.zero_grad()


def test_mcmc_relocate_rewrites_dead_gaussians_and_resets_state(lfs, oracle_mod):
    from lichtfeld_studio_amd import strategies
    model, dead = _model()
    st = strategies.MCMC(model, strategies.OptimizationParameters(), generator=torch.Generator(device=DEV).manual_seed(3))
    _prime_optimizer(st)
    before = {k: n(getattr(model, k)).copy() for k in strategies._PARAM_NAMES}
    opac0, scales0 = n(model.get_opacity()).copy(), n(model.get_scaling()).copy()
    # replay the sampling on the host: the same uniforms (identically seeded generator), inverse CDF over the alive opacities in fp64
    g2 = torch.Generator(device=DEV).manual_seed(3)
    u = torch.rand(len(opac0), dtype=torch.float64, device=DEV, generator=g2).cpu().numpy()
    w = np.where(dead.numpy(), 0.0, opac0.astype(np.float64))
    cdf = np.cumsum(w)
    sampled = np.searchsorted(cdf, u[dead.numpy()] * cdf[-1], side="right")
    assert not dead.numpy()[sampled].any()
    n_dead = int(st.relocate_gs())            # a device tensor: the relocation itself needs no host round trip
    assert n_dead == int(dead.sum()) > 0
    dead_idx = dead.nonzero().squeeze(-1).numpy()
    counts = np.bincount(sampled, minlength=len(opac0))
    ratios = np.minimum(counts[sampled] + 1, 51).astype(np.int32)
    binoms = n(st.binoms)
    new_o, new_s = oracle_mod.relocation(opac0[sampled], scales0[sampled], ratios, binoms, 51)
    new_o = np.clip(new_o, 0.005, 1 - 1e-7)
    after = {k: n(getattr(model, k)) for k in strategies._PARAM_NAMES}
    np.testing.assert_allclose(1 / (1 + np.exp(-after["raw_opacities"][sampled])), new_o, rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(np.exp(after["raw_scales"][sampled]), new_s, rtol=2e-5)
    for k in strategies._PARAM_NAMES:                       # dead <- copy of its sampled source (after the update)
        assert np.array_equal(after[k][dead_idx], after[k][sampled]), k
    untouched = np.ones(len(opac0), bool); untouched[dead_idx] = False; untouched[sampled] = False
    for k in strategies._PARAM_NAMES:
        assert np.array_equal(after[k][untouched], before[k][untouched]), k
    for k in strategies._PARAM_NAMES:                       # Adam moments of the sampled sources are reset, others kept
        s = st.optimizer.state[id(getattr(model, k))]
        assert float(s["exp_avg"][torch.from_numpy(sampled).to(DEV)].abs().max()) == 0.0
        assert float(s["exp_avg_sq"][torch.from_numpy(sampled).to(DEV)].abs().max()) == 0.0
        assert float(s["exp_avg"][torch.from_numpy(np.nonzero(untouched)[0]).to(DEV)].abs().min()) > 0.0


def test_mcmc_add_new_grows_by_five_percent_and_extends_state(lfs):
    from lichtfeld_studio_amd import strategies
    model, _ = _model(N=2000, dead_frac=0.0)
    resized = []
    st = strategies.MCMC(model, strategies.OptimizationParameters(max_cap=2080), generator=torch.Generator(device=DEV).manual_seed(1),
                         on_resize=lambda: resized.append(1))
    _prime_optimizer(st)
    old = {k: n(getattr(model, k)).copy() for k in strategies._PARAM_NAMES}
    steps = {k: st.optimizer.state[id(getattr(model, k))]["step_count"] for k in strategies._PARAM_NAMES}
    assert st.add_new_gs() == 80 and resized            # min(max_cap, int(1.05 * 2000)) - 2000
    for k in strategies._PARAM_NAMES:
        p = getattr(model, k)
        assert p.shape[0] == 2080 and p.requires_grad and p.is_leaf and p.grad is None
        s = st.optimizer.state[id(p)]
        assert s["exp_avg"].shape == p.shape and s["step_count"] == steps[k]
        assert float(s["exp_avg"][2000:].abs().max()) == 0.0 and float(s["exp_avg_sq"][2000:].abs().max()) == 0.0
        assert st.optimizer.param_groups[strategies._PARAM_NAMES.index(k)]["params"][0] is p
    # the new rows are copies of (updated) existing rows; means / sh / quats of existing rows are unchanged
    cur = {k: n(getattr(model, k)) for k in strategies._PARAM_NAMES}
    for k in ["means", "sh0", "shN", "raw_quats"]:
        assert np.array_equal(cur[k][:2000], old[k])
    src = [int(np.nonzero((cur["means"][:2000] == row).all(1))[0][0]) for row in cur["means"][2000:]]
    for k in strategies._PARAM_NAMES:
        assert np.array_equal(cur[k][2000:], cur[k][src]), k
    assert st.add_new_gs() == 0                          # capped


def test_mcmc_noise_and_sh_schedule(lfs):
    from lichtfeld_studio_amd import strategies
    model, _ = _model(N=1000, sh_degree=2, dead_frac=0.0)
    model.active_sh_degree = 0
    st = strategies.MCMC(model, strategies.OptimizationParameters(sh_degree_interval=10), generator=torch.Generator(device=DEV).manual_seed(1))
    others = {k: n(getattr(model, k)).copy() for k in strategies._PARAM_NAMES if k != "means"}
    m0 = n(model.means).copy()
    for it in range(1, 41):
        st.post_backward(it)
    assert model.active_sh_degree == 2                   # incremented at 10, 20, capped at the maximum
    assert np.abs(n(model.means) - m0).max() > 0 and np.isfinite(n(model.means)).all()
    for k, v in others.items():
        assert np.array_equal(n(getattr(model, k)), v), k


def test_default_strategy_duplicate_split_prune_reset(lfs):
    from lichtfeld_studio_amd import strategies
    model, _ = _model(N=3000, dead_frac=0.05)
    p = strategies.OptimizationParameters(grow_scale3d=0.035, prune_opacity=0.005)
    st = strategies.DefaultStrategy(model, p, generator=torch.Generator(device=DEV).manual_seed(2))
    _prime_optimizer(st)
    N = 3000
    info = torch.zeros(2, N, device=DEV)
    info[0] = 4.0
    info[1, :600] = 4.0 * 1e-3                            # the first 600 have a high average gradient (1e-3 > 2e-4)
    scales_max = model.get_scaling().detach().max(-1).values
    is_small = scales_max <= 0.035
    n_dup, n_split = int(is_small[:600].sum()), int((~is_small[:600]).sum())
    assert n_dup > 50 and n_split > 50
    means0 = model.means.detach().clone()
    scales0 = model.get_scaling().detach().clone()
    st.grow_gs(600, info)
    assert model.means.shape[0] == N + n_dup - n_split + 2 * n_split
    # split children: scale = parent / 1.6, means around the parent; Adam state of every new row is zero
    s = st.optimizer.state[id(model.means)]
    assert float(s["exp_avg"][-2 * n_split:].abs().max()) == 0.0
    parents = (~is_small[:600]).nonzero().squeeze(-1)
    child_scales = model.get_scaling().detach()[-2 * n_split:-n_split]
    assert torch.allclose(child_scales * 1.6, scales0[parents], rtol=1e-5)
    assert float((model.means.detach()[-2 * n_split:-n_split] - means0[parents]).norm(dim=-1).max()) < 1.0
    n_before = model.means.shape[0]
    dead_now = int((model.get_opacity() < 0.005).sum())
    st.prune_gs(700)
    assert model.means.shape[0] == n_before - dead_now and dead_now > 0
    st.reset_opacity()
    assert float(model.get_opacity().max()) <= 2 * 0.005 + 1e-6
    so = st.optimizer.state[id(model.raw_opacities)]
    assert float(so["exp_avg"].abs().max()) == 0.0 and so["step_count"] == 1
    for k in strategies._PARAM_NAMES:
        assert getattr(model, k).shape[0] == model.means.shape[0]


def test_default_strategy_fused_refinement_one_host_read_same_result(lfs):
    """SURVEY.md §8f row 3 for ADC: grow + prune of a refinement step as device-side index arithmetic - the result of the reference's sequence
    (grow_gs + prune_gs) bit for bit, with ONE device->host read (the new Gaussian count) where the sequence needs seven."""
    import warnings
    from lichtfeld_studio_amd import strategies
    N = 3000
    outs, reads = [], []
    for fused in (False, True):
        model, _ = _model(N=N, dead_frac=0.05)
        p = strategies.OptimizationParameters(grow_scale3d=0.035, prune_opacity=0.005, reset_every=300)
        st = strategies.DefaultStrategy(model, p, generator=torch.Generator(device=DEV).manual_seed(2))
        _prime_optimizer(st)
        info = torch.zeros(2, N, device=DEV)
        info[0] = 4.0
        info[1, :900] = 4e-3
        rnd = torch.randn(2, N, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))
        run = (lambda: st.grow_and_prune_fused(700, info, rnd)) if fused else (lambda: (st.grow_gs(700, info, rnd), st.prune_gs(700)))
        if fused:   # warm-up on a copy of the state would change it: count on the first run, module loads do not go through the sync-debug hook
            pass
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("warn")
        try:
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                run()
        finally:
            torch.cuda.set_sync_debug_mode("default")
        reads.append(len([x for x in w if "synchroniz" in str(x.message).lower()]))
        names = strategies._PARAM_NAMES
        outs.append([n(getattr(model, k)) for k in names] + [n(st.optimizer.state[id(getattr(model, k))]["exp_avg"]) for k in names])
        assert outs[-1][0].shape[0] > N
    for a, b in zip(*outs):
        assert a.shape == b.shape and np.array_equal(a, b)
    assert reads[1] == 1 and reads[0] >= 5, reads


def test_replicas_with_the_same_seed_densify_identically(lfs):
    from lichtfeld_studio_amd import strategies
    outs = []
    for _ in range(2):
        model, _d = _model(N=2500, dead_frac=0.1)
        st = strategies.MCMC(model, strategies.OptimizationParameters(start_refine=0, refine_every=1), generator=torch.Generator(device=DEV).manual_seed(11))
        _prime_optimizer(st)
        st.post_backward(7)
        outs.append([n(x).copy() for x in model.parameters()])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_mcmc_training_grows_the_model_and_fits_the_views(lfs):
    """End to end: L1 + D-SSIM loss, regularisers, relocation / growth / noise. Starts from a random cloud with a third of the
    ground truth's Gaussians; PSNR has to improve clearly and the count has to grow to the cap."""
    import convergence_check as cc
    from lichtfeld_studio_amd import scenes, strategies
    from lichtfeld_studio_amd.trainer import GutTrainer
    gt, _ = cc.make_task(n=3000, size=96, n_views=6, sh_degree=1)
    targets = cc.render_views_hip(gt, DEV)
    g = torch.Generator().manual_seed(3)
    N0 = 1000
    init = scenes.Scene("init", 96, 96, 1, (torch.rand(N0, 3, generator=g) * 2 - 1) * 2.0, torch.randn(N0, 4, generator=g),
                        torch.full((N0, 3), math.log(0.12)), torch.full((N0,), 0.0), 0.3 * torch.randn(N0, 1, 3, generator=g), torch.zeros(N0, 3, 3),
                        gt.viewmats, gt.Ks)
    op = strategies.OptimizationParameters(iterations=1200, start_refine=100, refine_every=50, stop_refine=1000, max_cap=1500, sh_degree_interval=300)
    tr = GutTrainer(init, DEV, iterations=1200, loss="l1_ssim", strategy="mcmc", opt_params=op, seed=5)
    tr.model.active_sh_degree = 0
    psnr0 = np.mean([cc.psnr(n(a), n(b)) for a, b in zip(cc.render_views_hip(init, DEV), targets)])
    for it in range(1200):
        tr.train_step([targets[it % 6]], views=[it % 6])
    m = tr.model
    cur = scenes.Scene("cur", 96, 96, 1, m.means.detach(), m.raw_quats.detach(), m.raw_scales.detach(), m.raw_opacities.detach(),
                       m.sh0.detach(), m.shN.detach(), tr.scene.viewmats, tr.scene.Ks)
    psnr1 = np.mean([cc.psnr(n(a), n(b)) for a, b in zip(cc.render_views_hip(cur, DEV), targets)])
    assert m.means.shape[0] == 1500 and m.active_sh_degree == 1
    assert all(torch.isfinite(p).all() for p in m.parameters())
    assert psnr1 > psnr0 + 4.0, (psnr0, psnr1)


def test_mcmc_refinement_step_needs_no_host_sync(lfs):
    """at the cap (N == max_cap, the steady state of an MCMC run) a refinement step - relocation of the dead Gaussians, Adam-state
    surgery, noise - is enqueued without a single device->host synchronisation: torch's sync debug mode turns any .item() / nonzero() / blocking copy
    into an error. (The reference synchronises for the dead count, mcmc.cpp:121-146.) add_new_gs, which changes N, is not part of this step."""
    from lichtfeld_studio_amd import strategies
    model, dead = _model(N=6000, dead_frac=0.05)
    st = strategies.MCMC(model, strategies.OptimizationParameters(max_cap=6000), generator=torch.Generator(device=DEV).manual_seed(4))
    _prime_optimizer(st)
    st.post_backward(600)                      # warm-up: first use of every kernel (module loads may synchronise)
    torch.cuda.synchronize()
    before = n(model.raw_opacities).copy()
    model.raw_opacities.data[::17] = -9.0      # new dead Gaussians
    torch.cuda.set_sync_debug_mode("error")
    try:
        assert st.is_refining(700)
        st.post_backward(700)
        st.post_backward(701)                  # a plain step: noise only
    finally:
        torch.cuda.set_sync_debug_mode("default")
    after = n(model.raw_opacities)
    assert (after[::17] > -9.0).all() and np.isfinite(after).all() and not np.array_equal(after, before)


def test_mcmc_inline_shN_adam_between_refinements_is_bit_identical(lfs):
    """With the MCMC strategy the SH backward applies shN's Adam update itself on the iterations between refinements (post_backward only adds noise to
    the means there) and leaves it to the optimizer launch on refining iterations (relocation rewrites shN rows and moments first): parameters and
    moments bit-identical to the path without the inline update, across a refinement, in the deterministic accumulation mode."""
    import convergence_check as cc
    from lichtfeld_studio_amd import strategies
    from lichtfeld_studio_amd.trainer import GutTrainer
    gt, init = cc.make_task(n=3000, size=96, n_views=4, sh_degree=2)
    targets = cc.render_views_hip(gt, DEV)
    lib = lfs.load_library()
    out = []
    try:
        lib.lfs_set_debug_flags(16)
        for inline in (True, False):
            op = strategies.OptimizationParameters(iterations=3000, start_refine=100, refine_every=5, stop_refine=2500, max_cap=3000)
            tr = GutTrainer(init, DEV, iterations=3000, loss="l1_ssim", strategy="mcmc", opt_params=op, seed=5)
            tr.inline_shN_adam = inline
            tr.iteration = 1501
            tr.model.raw_opacities.data[::13] = -9.0          # something to relocate at the refining iterations (1505, 1510)
            for it in range(11):
                tr.train_step([targets[it % 4]], views=[it % 4])
            torch.cuda.synchronize()
            out.append([n(p) for p in tr.model.parameters()] + [n(tr.optimizer.state[id(p)]["exp_avg"]) for p in tr.model.parameters()]
                       + [tr.optimizer.state[id(tr.model.shN)]["step_count"]])
    finally:
        lib.lfs_set_debug_flags(0)
    assert out[0][-1] == out[1][-1] == 11
    for a, b in zip(out[0][:-1], out[1][:-1]):
        assert a.shape == b.shape and np.array_equal(a, b)
