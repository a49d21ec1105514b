"""GPU: the product's strategy layer (strategies.MCMC / DefaultStrategy + FusedAdam + ExponentialLR over the HIP kernels) against the REFERENCE'S OWN strategy
layer run on the CPU (tests/golden/ref_strategy.npz: mcmc.cpp, default_strategy.cpp, strategy_utils.cpp, fused_adam.cpp, scheduler.cpp compiled in place against
libtorch over the reference's own kernels - oracle/make_golden_ref_strategy.py). Both sides go through the scenarios of refstrategy_util.py: same initial model,
same gradients, post_backward(iter) then step(iter) - the trainer's order, trainer.cpp:744-756 -, with every random draw the reference made replayed into the product (the MI355X generator produces other streams):
  * MCMC relocation: the reference's multinomial picks become the uniforms that make lfs_mcmc_relocate's inverse-CDF search pick the same sources;
  * MCMC growth: the multinomial indices; SGLD noise and ADC split offsets: the normal deviates.
Compared after every refining / resetting iteration and at the end: Gaussian count, SH degree, per-group learning rate and Adam step counts exactly; the 6 parameter
tensors and both Adam moments to 2e-5 relative + 1e-7 (gsplat::relocation sums ~1300 fp32 binomial terms); the MCMC means to 2e-5 relative + 1e-4 (twelve SGLD
steps of magnitude ~10 each through the fast exponential)."""
import os

import numpy as np
import pytest
import torch

import refstrategy_util as U
from gpu_util import n, t

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", U.GOLD))
PRODUCT_NAMES = ("means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opacities")


def test_parameter_defaults_are_the_reference_structs():
    """gs::param::OptimizationParameters{} as compiled from include/core/parameters.hpp vs strategies.OptimizationParameters()"""
    from lichtfeld_studio_amd import strategies
    p = strategies.OptimizationParameters()
    for k in ("iterations", "sh_degree_interval", "means_lr", "shs_lr", "opacity_lr", "scaling_lr", "rotation_lr", "lambda_dssim", "min_opacity", "refine_every",
              "start_refine", "stop_refine", "grad_threshold", "opacity_reg", "scale_reg", "max_cap", "prune_opacity", "grow_scale3d", "prune_scale3d", "reset_every",
              "pause_refine_after_reset"):
        ref = GOLD[f"defaults/{k}"].item()
        assert np.float32(getattr(p, k)) == np.float32(ref), (k, getattr(p, k), ref)
    assert bool(p.revised_opacity) == bool(GOLD["defaults/revised_opacity"].item())


class Patched:
    """torch.rand / torch.randn replaced for the duration of one post_backward"""

    def __init__(self, rand=None, randn=None):
        self.repl = {"rand": rand, "randn": randn}

    def __enter__(self):
        self.saved = {k: getattr(torch, k) for k in self.repl}
        for k, f in self.repl.items():
            if f is not None:
                setattr(torch, k, f)

    def __exit__(self, *a):
        for k, f in self.saved.items():
            setattr(torch, k, f)


def build(sc):
    from lichtfeld_studio_amd import strategies
    from lichtfeld_studio_amd.rasterizer import SplatModel
    init = U.initial(sc)
    mk = lambda a: t(a).contiguous().requires_grad_(True)
    model = SplatModel(mk(init["means"]), mk(init["sh0"]), mk(init["shN"]), mk(init["scaling"]), mk(init["rotation"]), mk(init["opacity"]), sc["sh_degree"],
                       active_sh_degree=0)
    kw = {k: (bool(v) if k == "revised_opacity" else v) for k, v in sc["params"].items()}
    params = strategies.OptimizationParameters(**kw)
    cls = strategies.MCMC if sc["kind"] == "mcmc" else strategies.DefaultStrategy
    return cls(model, params, scene_scale=sc["scene_scale"], generator=torch.Generator(device=DEV).manual_seed(0))


def draws_of(name, it):
    names = [str(x) for x in GOLD[f"{name}/it{it}/draws"] if str(x)]
    return [(nm, GOLD[f"{name}/it{it}/draw{k}"]) for k, nm in enumerate(names)]


def compare(st, name, it, full, means_atol=1e-7):
    m = st.model
    pre = f"{name}/it{it}/"
    assert m.means.shape[0] == int(GOLD[pre + "N"]), (it, m.means.shape[0], int(GOLD[pre + "N"]))
    assert m.active_sh_degree == int(GOLD[pre + "active_sh_degree"]), it
    lr = np.array([g["lr"] for g in st.optimizer.param_groups])
    np.testing.assert_allclose(lr, GOLD[pre + "lr"], rtol=1e-13, atol=0)
    worst = 0.0
    for ref_name, prod_name in zip(U.NAMES, PRODUCT_NAMES):
        p = getattr(m, prod_name)
        state = st.optimizer.state.get(id(p))
        assert (state["step_count"] if state else -1) == int(GOLD[pre + ref_name + ".step"]), (it, ref_name)
        if not full:
            continue
        for key, got in (("", p), (".exp_avg", state["exp_avg"] if state else None), (".exp_avg_sq", state["exp_avg_sq"] if state else None)):
            if (pre + ref_name + key) not in GOLD.files:
                assert got is None, (it, ref_name, key)
                continue
            ref, got = GOLD[pre + ref_name + key], n(got).reshape(-1)
            assert ref.shape == got.shape, (it, ref_name, key, ref.shape, got.shape)
            err = np.abs(got - ref) / (2e-5 * np.abs(ref) + (means_atol if ref_name == "means" and not key else 1e-7))
            worst = max(worst, float(err.max()) if err.size else 0.0)
            assert np.isfinite(got).all() and (err <= 1).all(), (it, ref_name + key, float(err.max()), int(err.argmax()))
            if key and ref.size:                       # which rows carry zeroed moments is exact
                assert np.array_equal(got == 0, ref == 0), (it, ref_name + key)
    return worst


def relocation_uniforms(st, local_picks):
    """uniforms [N] (fp64) that make lfs_mcmc_relocate draw, for the k-th dead Gaussian, the alive Gaussian the reference's multinomial drew (index into the alive list)"""
    m, p = st.model, st.params
    raw_o, q = n(m.raw_opacities).astype(np.float32), n(m.raw_quats)
    o = (np.float32(1) / (np.float32(1) + np.exp(-raw_o))).astype(np.float32)
    dead = (o <= np.float32(p.min_opacity)) | ((q * q).sum(-1) < 1e-8)
    dead_idx, alive_idx = np.nonzero(dead)[0], np.nonzero(~dead)[0]
    assert len(dead_idx) == len(local_picks), (len(dead_idx), len(local_picks))
    w = np.where(dead, 0.0, o.astype(np.float64))
    cdf = np.cumsum(w)
    u = np.full(len(o), 0.5)
    src = alive_idx[local_picks]
    u[dead_idx] = (cdf[src] - 0.5 * w[src]) / cdf[-1]
    return torch.from_numpy(u).to(DEV)


@pytest.mark.parametrize("name", ["mcmc"])
def test_mcmc_follows_the_reference_strategy(lfs, name):
    sc = U.SCENARIOS[name]
    st = build(sc)
    worst = 0.0
    for it in range(sc["it0"] + 1, sc["it0"] + sc["iters"] + 1):
        for p, g in zip(st.model.parameters(), U.grads([tuple(p.shape) for p in st.model.parameters()], it)):
            p.grad = t(g.reshape(p.shape))
        draws = draws_of(name, it)
        multis = [a for nm, a in draws if nm == "multinomial"]
        noise = [a for nm, a in draws if nm == "randn_like"][0]
        cur = st.model.means.shape[0]
        n_new = max(0, min(st.params.max_cap, int(1.05 * cur)) - cur) if st.is_refining(it) else 0
        add = multis.pop() if n_new > 0 else None
        assert add is None or len(add) == n_new
        u = relocation_uniforms(st, multis[0]) if multis else None
        assert len(multis) <= 1
        if add is not None:
            st.multinomial_sample = lambda weights, k, replacement=True, _a=add: t(_a, torch.int64)
        rand = (lambda *a, **kw: u) if u is not None else (lambda *a, **kw: torch.full((cur,), 0.5, dtype=torch.float64, device=DEV))
        with Patched(rand=rand, randn=lambda shape, **kw: t(noise).reshape(tuple(shape))):
            st.post_backward(it)
        st.step(it)          # the trainer's order (trainer.cpp:744-756): tensors replaced by the refinement have no gradient and are skipped by FusedAdam
        assert bool(GOLD[f"{name}/it{it}/refining"]) == st.is_refining(it)
        # means: every iteration adds lr * 5e5 (= 100 here) x Sigma x noise x gate(opacity) - terms of magnitude 10, whose fp32 rounding (and the kernel's fast
        # exponential in the gate) is what the two sides differ by: 1e-4 absolute after twelve iterations (measured 0.9 of 5e-5), against displacements of order 1 - 10 per iteration
        worst = max(worst, compare(st, name, it, it in sc["full_state"], means_atol=1e-4))
    print(f"{name}: worst deviation {worst:.3f} of the bar (2e-5 relative + 1e-7)")


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name", ["default", "default_revised_opacity"])
def test_default_strategy_follows_the_reference_strategy(lfs, name, fused):
    sc = U.SCENARIOS[name]
    st = build(sc)
    st.fused_refine = fused
    worst = 0.0
    for it in range(sc["it0"] + 1, sc["it0"] + sc["iters"] + 1):
        for p, g in zip(st.model.parameters(), U.grads([tuple(p.shape) for p in st.model.parameters()], it)):
            p.grad = t(g.reshape(p.shape))
        refining = bool(GOLD[f"{name}/it{it}/refining"])
        assert refining == (st.is_refining(it) and it < st.params.stop_refine)
        draws = draws_of(name, it)
        info = t(U.densification_info(st.model.means.shape[0], it)) if refining else None

        def randn(shape, **kw):
            rec = draws[0][1].reshape(2, -1, 3)
            if tuple(shape) == rec.shape:                                  # the reference's own draw: deviates for the split rows, in index order
                return t(rec)
            full = np.zeros(tuple(shape), np.float32)                      # the fused path draws per original row: place them at the rows that will be split
            full[:, GOLD[f"{name}/it{it}/split_idx"]] = rec
            return t(full)
        with Patched(randn=randn):
            st.post_backward(it, info)
        st.step(it)
        worst = max(worst, compare(st, name, it, it in sc["full_state"]))
    print(f"{name} fused={fused}: worst deviation {worst:.3f} of the bar")
