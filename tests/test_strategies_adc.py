"""CPU: the one-sync form of the ADC refinement (DefaultStrategy.grow_and_prune_fused) against the reference's sequence duplicate -> split -> prune
(grow_gs + prune_gs, mirrors of default_strategy.cpp:47-249): parameters and Adam moments bit for bit, given the same normal deviates. Pure index
arithmetic on torch tensors - the one HIP kernel involved (quats_to_rotmats) is replaced by its formula here and runs for real in
tests/test_gpu_strategies.py."""
import torch


def _q2r(q):
    w, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)


def _model(N, seed=0):
    from lichtfeld_studio_amd import scenes
    from lichtfeld_studio_amd.rasterizer import SplatModel
    sc = scenes.syn_a(seed=seed, n=N, sh_degree=1)
    g = torch.Generator().manual_seed(seed + 1)
    raw_o = sc.raw_opacities.clone()
    raw_o[torch.rand(N, generator=g) < 0.05] = -8.0          # pruned by the opacity test
    raw_s = sc.raw_scales.clone()
    raw_s[torch.rand(N, generator=g) < 0.03] = 1.0           # pruned by the scale test (after the first opacity reset)
    raw_q = sc.raw_quats.clone()
    raw_q[torch.rand(N, generator=g) < 0.01] = 0.0           # pruned by the rotation test
    mk = lambda t: t.clone().contiguous().requires_grad_(True)
    return SplatModel(mk(sc.means), mk(sc.sh0), mk(sc.shN), mk(raw_s), mk(raw_q), mk(raw_o), 1)


def _run(fused, it, n_high, monkeypatch, inject=True):
    from lichtfeld_studio_amd import strategies
    monkeypatch.setattr(strategies.ops, "quats_to_rotmats", _q2r)
    N = 3000
    m = _model(N)
    p = strategies.OptimizationParameters(grow_scale3d=0.035, prune_opacity=0.005, reset_every=300)
    st = strategies.DefaultStrategy(m, p, generator=torch.Generator().manual_seed(2))
    g = torch.Generator().manual_seed(5)
    for prm in m.parameters():
        st.optimizer.state[id(prm)] = {"exp_avg": torch.randn(prm.shape, generator=g), "exp_avg_sq": torch.rand(prm.shape, generator=g), "step_count": 1}
    info = torch.zeros(2, N)
    info[0] = 4.0
    info[1, :n_high] = 4e-3                                   # a high average screen-space gradient: duplicated (small) or split (large)
    rnd = torch.randn(2, N, 3, generator=torch.Generator().manual_seed(9)) if inject else None   # None: each path draws from the strategy's own generator
    if fused:
        st.grow_and_prune_fused(it, info, rnd)
    else:
        st.grow_gs(it, info, rnd)
        st.prune_gs(it)
    names = strategies._PARAM_NAMES
    return ([getattr(m, k).detach() for k in names] + [st.optimizer.state[id(getattr(m, k))]["exp_avg"] for k in names]
            + [st.optimizer.state[id(getattr(m, k))]["exp_avg_sq"] for k in names])


def test_fused_refinement_equals_the_reference_sequence(monkeypatch):
    for it, n_high in ((700, 900), (200, 900), (700, 0), (700, 3000)):   # with / without the scale test; nothing grows; everything grows
        a, b = _run(False, it, n_high, monkeypatch), _run(True, it, n_high, monkeypatch)
        assert a[0].shape[0] != 3000 or n_high == 0
        for x, y in zip(a, b):
            assert x.shape == y.shape and torch.equal(x, y), (it, n_high, x.shape, y.shape)


def test_fused_refinement_consumes_the_generator_like_the_reference_sequence(monkeypatch):
    """Without injected deviates both paths draw [2, n_split, 3] from the strategy's generator - the draw of the reference's split()
    (default_strategy.cpp:100-104): the same seed gives the same model on both paths (round-2 advisor finding: the fused path drew [2, N, 3])."""
    for it, n_high in ((700, 900), (700, 3000)):
        a, b = _run(False, it, n_high, monkeypatch, inject=False), _run(True, it, n_high, monkeypatch, inject=False)
        for x, y in zip(a, b):
            assert x.shape == y.shape and torch.equal(x, y), (it, n_high)
