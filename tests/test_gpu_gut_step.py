"""GPU: the C++ training step (csrc/gut_step.hip, one host call per step, no host read on the critical path) against the same kernels enqueued call by
call from Python (fused.py) - which tests/test_gpu_fused.py, test_gpu_headline_parity.py and test_gpu_refk_golden.py hold to the oracle and to the
reference's own kernels. Deterministic rasterizer sums (lfs_set_debug_flags(16)) make the comparison BIT-exact: parameters and Adam moments after
several steps, also when the first attempt of a step did not fit its workspace and was run again, and on a view that sees nothing."""
import numpy as np
import pytest
import torch

from gpu_util import n, noise_allclose, noise_check, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NAMES = ["means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opacities"]


def _same_state(a, b, steps):
    for name, pa, pb in zip(NAMES, a.model.parameters(), b.model.parameters()):
        assert torch.equal(pa, pb), (name, float((pa - pb).abs().max()))
        sa, sb = a.optimizer.state[id(pa)], b.optimizer.state[id(pb)]
        assert sa["step_count"] == sb["step_count"] == steps, name
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]), name


def _pair(lfs, sc, **kw):
    from lichtfeld_studio_amd.trainer import GutTrainer
    a, b = GutTrainer(sc, DEV, iterations=7000, **kw), GutTrainer(sc, DEV, iterations=7000, **kw)
    b.cxx_step = False
    a.iteration = b.iteration = 1500   # shN is being optimised: the all-inline step
    return a, b


def test_cxx_step_is_bit_identical_to_the_python_enqueued_step(lfs):
    from lichtfeld_studio_amd import scenes
    sc = scenes.syn_a(n=7000, sh_degree=2)
    target = torch.rand(3, sc.height, sc.width, generator=torch.Generator().manual_seed(3)).to(DEV) * 0.7
    lib = lfs.load_library()
    try:
        lib.lfs_set_debug_flags(16)
        a, b = _pair(lfs, sc)
        for _ in range(4):
            la, lb = a.train_step([target], views=[0]), b.train_step([target], views=[0])
        torch.cuda.synchronize()
    finally:
        lib.lfs_set_debug_flags(0)
    assert a._gut_step is not None and b._gut_step is None
    # (the loss VALUE is a float-atomic sum of per-wavefront partials over 256 slots - its last bit depends on arrival order even in the deterministic mode, which
    #  covers the gradient accumulators; the parameters below do not depend on it)
    noise_check("C++ step vs Python step loss value", abs(float(la) - float(lb)), 1e-5 * float(lb))
    assert float(la) > 0
    assert a.last_n_isects == b.last_n_isects > 0
    assert torch.equal(a.last_visible, b.last_visible)
    _same_state(a, b, 4)


@pytest.mark.parametrize("n", [1, 63, 65, 129])
def test_cxx_step_on_ragged_gaussian_counts(lfs, n):
    """N = 1 / 63 / 65 / 129: partial last wavefronts in every per-Gaussian kernel (the finish pass hands its accumulator rows over through wave-private
    LDS in blocks of 64 rows; the SH kernels work on 64 Gaussians per workgroup) and an image of mostly empty tiles - the C++ step against the
    Python-enqueued one, bit for bit."""
    from lichtfeld_studio_amd import scenes
    sc = scenes.syn_a(n=n, sh_degree=1)
    target = torch.rand(3, sc.height, sc.width, generator=torch.Generator().manual_seed(9)).to(DEV) * 0.7
    lib = lfs.load_library()
    try:
        lib.lfs_set_debug_flags(16)
        a, b = _pair(lfs, sc)
        for _ in range(3):
            la, lb = a.train_step([target], views=[0]), b.train_step([target], views=[0])
        torch.cuda.synchronize()
    finally:
        lib.lfs_set_debug_flags(0)
    noise_check("C++ step vs Python step loss value", abs(float(la) - float(lb)), 1e-5 * float(lb))
    assert float(la) > 0
    assert a.last_n_isects == b.last_n_isects
    _same_state(a, b, 3)


@pytest.mark.parametrize("kind", ["l1_ssim", "bilateral", "mcmc", "mse_early"])
def test_cxx_split_step_is_bit_identical_for_the_other_losses_and_mcmc(lfs, kind):
    """The steps that keep gradient tensors - L1 + D-SSIM, bilateral grid (+ its TV loss and Adam), MCMC (noise every step, regularisers; relocation on the
    refining iteration 1600 lies inside the 3 steps when started at 1598), and MSE while iteration <= 1000 - through lfs_gut_view_forward / _backward_sh /
    _backward_finish (loss kernels in between, shN's Adam inside the SH backward where the Python path has it) against the call-by-call Python enqueue."""
    from lichtfeld_studio_amd import scenes, strategies
    from lichtfeld_studio_amd.trainer import GutTrainer
    sc = scenes.syn_a(n=6000, sh_degree=2)
    target = torch.rand(3, sc.height, sc.width, generator=torch.Generator().manual_seed(5)).to(DEV) * 0.7
    kw = {"l1_ssim": dict(loss="l1_ssim"), "bilateral": dict(loss="l1_ssim", use_bilateral_grid=True), "mse_early": dict(),
          "mcmc": dict(loss="l1_ssim", strategy="mcmc", opt_params=strategies.OptimizationParameters(iterations=30000, max_cap=6000))}[kind]
    lib = lfs.load_library()
    try:
        lib.lfs_set_debug_flags(16)
        a, b = GutTrainer(sc, DEV, iterations=30000, **kw), GutTrainer(sc, DEV, iterations=30000, **kw)
        b.cxx_step = False
        a.iteration = b.iteration = 10 if kind == "mse_early" else 1598
        # the bilateral grid's own gradient is a float-atomic scatter (order-dependent in the last bit, in either driver) and its Adam step feeds the next
        # render: with the grid, the FIRST step (identity grids on both sides) is compared bit for bit, later ones could only agree to rounding
        for _ in range(1 if kind == "bilateral" else 3):
            la, lb = a.train_step([target], views=[0]), b.train_step([target], views=[0])
        torch.cuda.synchronize()
    finally:
        lib.lfs_set_debug_flags(0)
    assert a._gut_step is not None and b._gut_step is None
    noise_check(f"C++ split step loss value {kind}", abs(float(la) - float(lb)), 1e-5 * abs(float(lb)))
    assert float(la) > 0
    assert a.last_n_isects == b.last_n_isects > 0
    same = torch.equal
    for name, pa, pb in zip(NAMES, a.model.parameters(), b.model.parameters()):
        pa, pb = pa.detach(), pb.detach()
        assert pa.shape == pb.shape and same(pa, pb), (kind, name, float((pa - pb).abs().max()))
    for name in NAMES:
        sa, sb = a.optimizer.state[id(getattr(a.model, name))], b.optimizer.state[id(getattr(b.model, name))]
        assert same(sa["exp_avg"], sb["exp_avg"]) and same(sa["exp_avg_sq"], sb["exp_avg_sq"]), (kind, name)
    if kind == "bilateral":
        noise_allclose("bilateral grids after one step", a.bilateral.grids, b.bilateral.grids, rtol=1e-4, atol=1e-6)
        assert not torch.equal(a.bilateral.grids, torch.zeros_like(a.bilateral.grids))


@pytest.mark.parametrize("capacity,longest", [(1500, 1024), (10 ** 7, 8)])
def test_overflowing_attempt_updates_nothing_and_is_run_again(lfs, capacity, longest):
    """capacity 1500 for ~13 000 intersections, or sort classes for tile lists of <= 1024 entries when a tile holds more: the first attempt raises
    the device flag, its Adam kernels return without touching a parameter, the host enlarges the workspace and the second attempt is the step."""
    from lichtfeld_studio_amd import scenes
    from lichtfeld_studio_amd.gut_step import GutStep
    sc = scenes.syn_a(n=4000 if longest == 1024 else 60000, sh_degree=1)
    target = torch.rand(3, sc.height, sc.width, generator=torch.Generator().manual_seed(4)).to(DEV) * 0.7
    lib = lfs.load_library()
    try:
        lib.lfs_set_debug_flags(16)
        a, b = _pair(lfs, sc)
        a._gut_step = GutStep(DEV, initial_capacity=capacity)
        a._gut_step.assumed_longest = longest
        for _ in range(2):
            la, lb = a.train_step([target], views=[0]), b.train_step([target], views=[0])
        torch.cuda.synchronize()
    finally:
        lib.lfs_set_debug_flags(0)
    gs = a._gut_step
    assert gs.retries >= 1, "the first attempt was meant to overflow"
    assert gs.n_isects == b.last_n_isects and gs.capacity >= gs.n_isects
    if longest == 8:
        assert gs.longest > 1024, gs.longest   # (otherwise this case does not test the sort-class guard)
    noise_check("overflow retry loss value", abs(float(la) - float(lb)), 1e-5 * float(lb))
    _same_state(a, b, 2)


def test_view_without_intersections_still_takes_its_adam_step(lfs):
    """A camera that looks away from every Gaussian: nothing is rendered, the loss is that of the background image, every gradient is zero (plus the
    regularisers) - and Adam still steps: moments decay, momentum moves the parameters (FusedAdam::step does not look at what the gradient holds).
    Python-enqueued path: fused.render_and_backward's adam_fallback (round-2 advisor finding: it used to skip the update altogether)."""
    from lichtfeld_studio_amd import scenes
    sc = scenes.syn_a(n=3000, sh_degree=1)
    sc.viewmats = sc.viewmats.clone()
    sc.viewmats[0, 2, 2], sc.viewmats[0, 0, 0] = -1.0, -1.0   # rotate the camera by 180 degrees about y: everything is behind it
    target = torch.rand(3, sc.height, sc.width, generator=torch.Generator().manual_seed(5)).to(DEV) * 0.7
    a, b = _pair(lfs, sc)
    good = sc.viewmats[0].clone()
    good[2, 2], good[0, 0] = 1.0, 1.0
    for tr in (a, b):   # one ordinary step first, so that the moments are not zero
        tr.scene.viewmats[0] = good.to(DEV)
    lib = lfs.load_library()
    try:
        lib.lfs_set_debug_flags(16)
        a.train_step([target], views=[0]); b.train_step([target], views=[0])
        for tr in (a, b):
            tr.scene.viewmats[0, 2, 2], tr.scene.viewmats[0, 0, 0] = -1.0, -1.0
        before = [p.detach().clone() for p in a.model.parameters()]
        la, lb = a.train_step([target], views=[0]), b.train_step([target], views=[0])
        torch.cuda.synchronize()
    finally:
        lib.lfs_set_debug_flags(0)
    assert a.last_n_isects == 0 and b.last_n_isects == 0
    assert abs(float(la) - float(lb)) <= 1e-6 * float(lb) and float(la) > 0     # (mse of the black image; summed in a different order by the two kernels)
    moved = [float((p.detach() - q).abs().max()) for p, q in zip(a.model.parameters(), before)]
    assert all(m > 0 for m in moved), moved                                      # momentum from the first step
    _same_state(a, b, 2)


def test_view_forward_backward_gradients_match_render_and_backward(lfs):
    """lfs_gut_view_forward + lfs_gut_view_backward (the gradient-tensor form: data-parallel ranks, several views per step) against
    fused.render_and_backward: same kernels, bit-identical gradients in the deterministic mode; with a caller-provided dL/d(render) as well."""
    from lichtfeld_studio_amd import fused, scenes
    from lichtfeld_studio_amd.gut_step import GutStep
    from lichtfeld_studio_amd.trainer import GutTrainer
    sc = scenes.syn_a(n=6000, sh_degree=2)
    tr = GutTrainer(sc, DEV, iterations=100)
    target = torch.rand(3, sc.height, sc.width, generator=torch.Generator().manual_seed(6)).to(DEV) * 0.7
    params = [p.detach() for p in tr.model.parameters()]
    lib = lfs.load_library()
    try:
        lib.lfs_set_debug_flags(16)
        g_ref = [torch.zeros_like(p) for p in params]
        l_ref = torch.zeros(1, device=DEV)
        out = fused.render_and_backward(tr.camera(0), tr.model, tr.bg, target, 0.5, g_ref, l_ref, accumulate=False, scale_reg=0.01, opacity_reg=0.02)
        gs = GutStep(DEV)
        deg, W, H = tr.model.get_active_sh_degree(), sc.width, sc.height
        vm, Km = tr.scene.viewmats[0], tr.scene.Ks[0]
        n_isects = gs.view_forward(params, deg, W, H, vm, Km, tr.bg)
        assert n_isects == out.n_isects
        assert torch.equal(gs.view("render", torch.float32, (1, H, W, 3)), out.image_hwc)
        assert torch.equal(gs.view("radii", torch.int32, (1, params[0].shape[0], 2)), out.radii)
        g = [torch.full_like(p, 7.0) for p in params]   # written, not added to
        l = torch.zeros(1, device=DEV)
        gs.view_backward(params, deg, W, H, vm, Km, tr.bg, g, False, target_chw=target, weight=0.5, loss_acc=l, scale_reg=0.01, opacity_reg=0.02)
        torch.cuda.synchronize()
        assert abs(float(l) - float(l_ref)) <= 1e-6 * float(l_ref)
        for name, x, y in zip(NAMES, g, g_ref):
            assert torch.equal(x, y), (name, float((x - y).abs().max()))
        # second view accumulated on top, with an explicit dL/d(render)
        v_render = torch.randn(1, H, W, 3, generator=torch.Generator().manual_seed(7)).to(DEV) * 1e-3
        gs.view_forward(params, deg, W, H, tr.scene.viewmats[0], Km, tr.bg)
        gs.view_backward(params, deg, W, H, vm, Km, tr.bg, g, True, v_render=v_render)
        torch.cuda.synchronize()
    finally:
        lib.lfs_set_debug_flags(0)
    assert all(torch.isfinite(x).all() for x in g)
    assert float((g[0] - g_ref[0]).abs().max()) > 0


# ---- round 6: lfs_gut_train_step_pipelined (two streams) against lfs_gut_train_step (one) ---------------------------------------------------------------------------
def _pipe_pair(sc):
    from lichtfeld_studio_amd.trainer import GutTrainer
    a, b = GutTrainer(sc, DEV, iterations=7000), GutTrainer(sc, DEV, iterations=7000)
    a.pipelined, b.pipelined = True, False
    a.iteration = b.iteration = 1500
    return a, b


@pytest.mark.parametrize("n_gauss,degree", [(7000, 2), (200000, 3), (65, 3)])
def test_pipelined_step_is_bit_identical_to_the_one_stream_step(lfs, n_gauss, degree):
    """The SH Adam pass of step k and the SH colours of step k + 1 run on the library's side stream, beside finish(k) / projection(k + 1) / tile lists / culling on the
    caller's: every buffer one of them writes while the other reads would show up here as a difference (200 000 Gaussians: kernels long enough to overlap for real).
    Several views in turn, deterministic rasterizer sums: parameters and moments bit for bit after 6 steps."""
    from lichtfeld_studio_amd import scenes
    sc = scenes.syn_a(n=n_gauss, sh_degree=degree)
    g = torch.Generator().manual_seed(5)
    targets = [torch.rand(3, sc.height, sc.width, generator=g).to(DEV) * 0.7 for _ in range(3)]
    n_views = sc.viewmats.shape[0]
    lib = lfs.load_library()
    try:
        lib.lfs_set_debug_flags(16)
        a, b = _pipe_pair(sc)
        for it in range(6):
            la, lb = a.train_step([targets[it % 3]], views=[it % n_views]), b.train_step([targets[it % 3]], views=[it % n_views])
        a.join_pipeline()
        torch.cuda.synchronize()
    finally:
        lib.lfs_set_debug_flags(0)
    assert a.last_plan.path == b.last_plan.path == "cxx_all"
    noise_check("pipelined vs one-stream loss value", abs(float(la) - float(lb)), 1e-5 * float(lb))
    assert a.last_n_isects == b.last_n_isects > 0
    _same_state(a, b, 6)
    assert float(a.optimizer.state[id(a.model.shN)]["exp_avg"].abs().sum()) > 0, "no gradient reached shN"


def test_pipelined_step_with_an_overflowing_attempt_and_a_change_of_step_form(lfs):
    """First attempt of a pipelined step overflows its workspace (nothing may be updated, on either stream), then the run switches to the gradient-tensor form
    (L1 + D-SSIM loss for one step: it reads sh0 / shN on the caller's stream and must wait for the side stream by itself) and back."""
    from lichtfeld_studio_amd import scenes
    from lichtfeld_studio_amd.gut_step import GutStep
    sc = scenes.syn_a(n=4000, sh_degree=1)
    target = torch.rand(3, sc.height, sc.width, generator=torch.Generator().manual_seed(4)).to(DEV) * 0.7
    lib = lfs.load_library()
    try:
        lib.lfs_set_debug_flags(16)
        a, b = _pipe_pair(sc)
        a._gut_step = GutStep(DEV, initial_capacity=1500)
        for it in range(5):
            kind = "l1_ssim" if it == 2 else "mse"
            a.loss_kind = b.loss_kind = kind
            la, lb = a.train_step([target], views=[0]), b.train_step([target], views=[0])
        a.join_pipeline()
        torch.cuda.synchronize()
    finally:
        lib.lfs_set_debug_flags(0)
    assert a._gut_step.retries >= 1, "the first attempt was meant to overflow"
    _same_state(a, b, 5)


def test_pipelined_step_float_atomics_stays_within_accumulation_noise(lfs):
    """The benchmarked mode (float atomics, no debug flag): pipelined and one-stream steps differ only by the order of the rasterizer's atomic sums."""
    from lichtfeld_studio_amd import scenes
    sc = scenes.syn_a(n=20000, sh_degree=3)
    target = torch.rand(3, sc.height, sc.width, generator=torch.Generator().manual_seed(6)).to(DEV) * 0.7
    a, b = _pipe_pair(sc)
    for it in range(4):
        la, lb = a.train_step([target], views=[it % sc.viewmats.shape[0]]), b.train_step([target], views=[it % sc.viewmats.shape[0]])
    a.join_pipeline()
    torch.cuda.synchronize()
    for name, pa, pb in zip(NAMES, a.model.parameters(), b.model.parameters()):
        assert rel_l2(n(pa), n(pb)) < 2e-4, (name, rel_l2(n(pa), n(pb)))


# ---- round 6: lfs_gut_train_step_ex (fused tail: SH backward + six Adam updates + the next view's SH colours in one launch) against lfs_gut_train_step -----------------
@pytest.mark.parametrize("n_gauss,degree", [(7000, 2), (200000, 3), (65, 3), (3000, 1)])
def test_fused_tail_step_is_bit_identical_to_the_three_pass_step(lfs, n_gauss, degree):
    """Round-robin view schedule (the trainer then knows the next view and the tail evaluates its colours: every step but the first skips its SH colour kernel), six steps,
    deterministic rasterizer sums: parameters and moments bit for bit."""
    from lichtfeld_studio_amd import scenes
    from lichtfeld_studio_amd.trainer import GutTrainer
    sc = scenes._syn_box("SYN-T", 7, n_gauss, 320, 192, 260.0, 5, sh_degree=degree)   # five orbit cameras: consecutive steps render different views
    g = torch.Generator().manual_seed(5)
    targets = [torch.rand(3, sc.height, sc.width, generator=g).to(DEV) * 0.7]
    lib = lfs.load_library()
    try:
        lib.lfs_set_debug_flags(16)
        a, b = GutTrainer(sc, DEV, iterations=7000), GutTrainer(sc, DEV, iterations=7000)
        a.fused_tail, b.fused_tail = True, False
        a.iteration = b.iteration = 1500
        for it in range(6):
            la, lb = a.train_step(targets), b.train_step(targets)
        torch.cuda.synchronize()
    finally:
        lib.lfs_set_debug_flags(0)
    assert a.last_plan.path == b.last_plan.path == "cxx_all"
    assert a._gut_step.colour_launches_saved == 5 and b._gut_step.colour_launches_saved == 0
    noise_check("fused tail vs three passes loss value", abs(float(la) - float(lb)), 1e-5 * float(lb))
    assert a.last_n_isects == b.last_n_isects > 0
    _same_state(a, b, 6)


def test_fused_tail_colours_are_dropped_when_the_next_step_is_another_one(lfs):
    """The tail evaluated colours for the scheduled view; the caller then renders a DIFFERENT view, raises the SH degree, or runs another step form in between: the
    workspace's colours must not be used (GutStep's book), and the results stay those of the three-pass step."""
    from lichtfeld_studio_amd import scenes
    from lichtfeld_studio_amd.trainer import GutTrainer
    sc = scenes._syn_box("SYN-T", 8, 5000, 320, 192, 260.0, 5, sh_degree=3)
    target = torch.rand(3, sc.height, sc.width, generator=torch.Generator().manual_seed(8)).to(DEV) * 0.7
    nv = sc.viewmats.shape[0]
    lib = lfs.load_library()
    try:
        lib.lfs_set_debug_flags(16)
        a, b = GutTrainer(sc, DEV, iterations=7000), GutTrainer(sc, DEV, iterations=7000)
        a.fused_tail, b.fused_tail = True, False
        a.iteration = b.iteration = 1997   # the SH degree rises after iteration 2000 (sh_degree_interval 1000): colours evaluated at the old degree are void
        a.model.active_sh_degree = b.model.active_sh_degree = 2
        for it in range(7):
            if it == 4:      # an explicit view that is not the scheduled one, announced by nobody
                kw = dict(views=[(it + 3) % nv])
            elif it == 5:    # explicit view + announcement of the next
                kw = dict(views=[1 % nv], next_views=[2 % nv])
            elif it == 6:
                kw = dict(views=[2 % nv])
            else:
                kw = {}
            la, lb = a.train_step([target], **kw), b.train_step([target], **kw)
        torch.cuda.synchronize()
    finally:
        lib.lfs_set_debug_flags(0)
    assert a.model.active_sh_degree == 3
    assert 1 <= a._gut_step.colour_launches_saved <= 4
    _same_state(a, b, 7)
