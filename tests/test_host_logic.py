"""CPU: host-side logic that needs no GPU — UT parameter round trip, scene determinism, view
sharding, the flat gradient bucket, FusedAdam's bookkeeping (bias-correction reciprocals, the
shN skip, per-parameter step counts) with the kernel call intercepted, ExponentialLR."""
import math

import numpy as np
import torch


def test_ut_params_roundtrip(lfs):
    ut = lfs.UnscentedTransformParameters(0.2, 1.5, 0.1, 0.3, False)
    back = lfs.UnscentedTransformParameters.from_tensor(ut.to_tensor())
    assert abs(back.alpha - 0.2) < 1e-7 and abs(back.beta - 1.5) < 1e-7 and back.require_all_sigma_points_valid is False
    d = lfs.UnscentedTransformParameters()
    assert (d.alpha, d.beta, d.kappa, d.in_image_margin_factor, d.require_all_sigma_points_valid) == (0.1, 2.0, 0.0, 0.1, True)


def test_enums_match_reference_values(lfs):
    assert int(lfs.CameraModelType.PINHOLE) == 0 and int(lfs.CameraModelType.ORTHO) == 1 and int(lfs.CameraModelType.FISHEYE) == 2
    assert int(lfs.ShutterType.ROLLING_TOP_TO_BOTTOM) == 0 and int(lfs.ShutterType.GLOBAL) == 4


def test_scenes_are_deterministic_and_shaped(lfs):
    from lichtfeld_studio_amd import scenes
    a, b = scenes.syn_a(), scenes.syn_a()
    assert torch.equal(a.means, b.means) and a.N == 10000 and a.width == 256
    s = scenes.syn_b(n=2000, n_views=6)
    assert s.sh0.shape == (2000, 1, 3) and s.shN.shape == (2000, 15, 3) and s.viewmats.shape == (6, 4, 4)
    R = s.viewmats[:, :3, :3]
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand(6, 3, 3), atol=1e-5)
    # every camera sits on the radius-10 orbit (heights -1/0/1) and looks at the origin
    cam_pos = -(R.transpose(1, 2) @ s.viewmats[:, :3, 3:]).squeeze(-1)
    assert torch.allclose(cam_pos.norm(dim=-1), torch.sqrt(100 + cam_pos[:, 1] ** 2), atol=1e-4)


def test_view_sharding_is_disjoint_and_covering(lfs):
    from lichtfeld_studio_amd.dist import views_for_step
    world, vpr, n_views = 4, 2, 64
    seen = []
    for step in range(8):
        batch = [v for r in range(world) for v in views_for_step(step, r, world, n_views, vpr)]
        assert len(set(batch)) == world * vpr
        seen += batch
    assert sorted(seen) == list(range(64))


def test_grad_bucket_views_and_gather(lfs):
    from lichtfeld_studio_amd.dist import GradBucket
    ps = [torch.zeros(5, 3), torch.zeros(5, 1, 3), torch.zeros(5)]
    b = GradBucket(ps)
    assert b.flat.numel() == 15 + 15 + 5
    gs = [torch.arange(15.).view(5, 3), None, torch.ones(5)]
    b.gather(gs)
    assert torch.equal(b.views[0], gs[0]) and torch.count_nonzero(b.views[1]) == 0 and torch.equal(b.flat[-5:], torch.ones(5))
    b.all_reduce()  # no process group: a no-op


def test_fused_adam_bookkeeping(lfs, monkeypatch):
    """fused_adam.cpp:22-95: per-group options, lazy state, step_count++ even when the shN group is
    skipped (iteration <= 1000), reciprocals in double then float."""
    from lichtfeld_studio_amd import fused_adam, ops
    calls = []
    monkeypatch.setattr(ops, "adam_step_multi", lambda entries: calls.append(list(entries)))
    ps = [torch.nn.Parameter(torch.zeros(4, 3)) for _ in range(6)]
    groups = [{"params": [p], "lr": 0.1 * (i + 1)} for i, p in enumerate(ps)]
    opt = fused_adam.FusedAdam(groups, fused=True)
    for p in ps:
        p.grad = torch.ones_like(p)
    opt.step(iteration=1)
    assert len(calls[-1]) == 5 and all(e[0] is not ps[2] for e in calls[-1])  # group 3 (shN) skipped
    assert opt.state[id(ps[2])]["step_count"] == 1
    e = calls[-1][0]
    assert abs(e[4] - 0.1) < 1e-12 and abs(e[8] - 1 / (1 - 0.9)) < 1e-6 and abs(e[9] - 1 / math.sqrt(1 - 0.999)) < 1e-3
    opt.step(iteration=1001)
    assert len(calls[-1]) == 6
    shn = [x for x in calls[-1] if x[0] is ps[2]][0]
    assert abs(shn[8] - 1 / (1 - 0.9 ** 2)) < 1e-6  # its step_count advanced while skipped
    ps[0].grad = None
    opt.step(iteration=1002)
    assert len(calls[-1]) == 5  # parameters without grad are skipped
    opt.zero_grad()
    assert all(p.grad is None for p in ps)


def test_exponential_lr_only_touches_means_group(lfs):
    from lichtfeld_studio_amd import fused_adam
    ps = [torch.nn.Parameter(torch.zeros(2)) for _ in range(2)]
    opt = fused_adam.FusedAdam([{"params": [ps[0]], "lr": 1.0}, {"params": [ps[1]], "lr": 1.0}])
    sch = fused_adam.ExponentialLR(opt, gamma=0.01 ** (1 / 100))
    for _ in range(100):
        sch.step()
    assert abs(opt.param_groups[0]["lr"] - 0.01) < 1e-9 and opt.param_groups[1]["lr"] == 1.0


def test_ssim_reference_is_self_consistent():
    """tests/ssim_reference.py (the checker of the GPU SSIM tests): identical images -> 1, symmetric in its arguments, window sums to 1,
    analytic gradient == finite differences (fp64 gradcheck)."""
    import torch
    import ssim_reference as ref
    g = torch.Generator().manual_seed(0)
    a = torch.rand(1, 2, 14, 15, generator=g, dtype=torch.float64)     # (11 x 11 window: a 4 x 5 interior; gradcheck makes two evaluations per input element)
    b = torch.rand(1, 2, 14, 15, generator=g, dtype=torch.float64)
    assert abs(sum(ref.GAUSS) - 1.0) < 1e-6
    assert torch.allclose(ref.ssim_map(a, a)[:, :, 5:-5, 5:-5], torch.ones(1, 2, 4, 5, dtype=torch.float64), atol=1e-12)
    assert torch.allclose(ref.ssim_map(a, b), ref.ssim_map(b, a), atol=1e-12)
    a.requires_grad_(True)
    nt = torch.get_num_threads()
    torch.set_num_threads(1)          # 840 evaluations of a 14 x 15 convolution: the thread pool's hand-over costs more than the arithmetic (88 s -> seconds on a busy host)
    try:
        assert torch.autograd.gradcheck(lambda x: ref.photometric_loss(x, b, 0.2), (a,), eps=1e-6, atol=1e-6)
    finally:
        torch.set_num_threads(nt)


def test_warmup_exponential_lr_follows_reference_schedule():
    """scheduler.cpp:27-63: linear warm-up from start factor to 1 over warmup_steps, then gamma^(step - warmup_steps)."""
    import lichtfeld_studio_amd  # noqa: F401
    from lichtfeld_studio_amd.fused_adam import WarmupExponentialLR

    class Opt:
        param_groups = [{"lr": 2e-3}, {"lr": 1.0}]
    o = Opt()
    s = WarmupExponentialLR(o, gamma=0.5, warmup_steps=4, warmup_start_factor=0.01)
    seen = []
    for _ in range(7):
        s.step()
        seen.append(o.param_groups[0]["lr"] / 2e-3)
    expect = [0.01 + 0.99 * 0.25, 0.01 + 0.99 * 0.5, 0.01 + 0.99 * 0.75, 1.0, 0.5, 0.25, 0.125]
    assert all(abs(a - b) < 1e-12 for a, b in zip(seen, expect)), seen
    assert abs(o.param_groups[1]["lr"] - 0.125) < 1e-12
    o2 = Opt(); o2.param_groups = [{"lr": 1.0}, {"lr": 1.0}]
    s2 = WarmupExponentialLR(o2, gamma=0.5, param_group_index=1)
    s2.step()
    assert o2.param_groups[0]["lr"] == 1.0 and o2.param_groups[1]["lr"] == 0.5


def test_step_plan_table():
    """trainer.plan_step - the pure function that picks the training-step form - over EVERY combination of its inputs: the invariants that make a skipped or
    doubled update impossible, and the rows the documented configurations must land on."""
    import itertools

    import lichtfeld_studio_amd  # noqa: F401
    from lichtfeld_studio_amd.trainer import plan_step
    base = dict(rasterizer="gut", fused_l2=True, world=1, force_collectives=False, sh_sharded=False, shard_rows=0, n_views=1, loss="mse", strategy=None, refining=False,
                iteration=3000, has_shN=True, optimizer_fused=True, bilateral=False)
    # documented rows
    assert plan_step(**base).path == "cxx_all"                                                    # the benchmarked step: one C++ call, no gradient tensors
    assert plan_step(**dict(base, iteration=500)).path == "cxx_views"                             # shN not optimised yet: gradient tensors, shN skipped
    assert plan_step(**dict(base, iteration=500)).skip_deferred
    assert plan_step(**dict(base, loss="l1_ssim")) == plan_step(**dict(base, loss="l1_ssim"))    # (frozen dataclass: comparable)
    p = plan_step(**dict(base, loss="l1_ssim"))
    assert p.path == "cxx_views" and p.inline_shN and not p.inline_all                            # L1 + D-SSIM: shN's update inline, the rest through the optimizer
    p = plan_step(**dict(base, strategy="mcmc", loss="l1_ssim", refining=True))
    assert p.path == "cxx_views" and not p.inline_shN                                             # relocation rewrites shN rows before the optimizer step
    p = plan_step(**dict(base, world=8))
    assert p.path == "cxx_views" and p.multi and not p.inline_shN and not p.skip_deferred         # data-parallel, replicated: every gradient through the all-reduce
    p = plan_step(**dict(base, world=8, n_views=8))
    assert p.path == "cxx_views"                                                                  # BASELINE config 4: 8 views per rank
    p = plan_step(**dict(base, n_views=8))
    assert p.path == "batch_views" and p.inline_shN                                               # several views on one rank: one SH pass over all of them
    p = plan_step(**dict(base, world=4, sh_sharded=True, shard_rows=1000))
    assert p.path == "py_views" and p.inline_shard and p.skip_deferred                            # SH-sharded: owners update their shard, shN never all-reduced
    assert plan_step(**dict(base, cxx_step=False)).path == "py_views" and plan_step(**dict(base, cxx_step=False)).inline_all
    assert plan_step(**dict(base, rasterizer="fastgs")).path == "fastgs" and plan_step(**dict(base, fused_l2=False)).path == "autograd"
    assert plan_step(**dict(base, world=1, force_collectives=True)).path == "cxx_views"           # one GPU running the multi-rank code path
    # the factored SH exchange (round 4): a variant of the replicated multi-rank layout only
    p = plan_step(**dict(base, world=8, factored_sh=True))
    assert p.path == "cxx_factored" and p.inline_shN and p.skip_deferred and p.multi               # shN updated inside the multi-view SH backward, never all-reduced
    assert plan_step(**dict(base, world=8, factored_sh=True, iteration=500)).inline_shN is False    # Adam does not read shN yet
    assert plan_step(**dict(base, world=8, factored_sh=True, strategy="mcmc", loss="l1_ssim", refining=True)).inline_shN is False
    assert plan_step(**dict(base, world=1, factored_sh=True)).path == "cxx_all"                     # one rank: nothing to exchange
    assert plan_step(**dict(base, world=1, force_collectives=True, factored_sh=True)).path == "cxx_factored"
    # invariants over the whole table
    n = 0
    for (world, force, sharded, n_views, loss, strategy, refining, iteration, has_shN, fused, bilateral, i_shN, i_all, cxx, batch) in itertools.product(
            (1, 2, 8), (False, True), (False, True), (1, 8), ("mse", "l1_ssim"), (None, "mcmc", "default"), (False, True), (500, 1000, 1001, 7000), (False, True),
            (False, True), (False, True), (False, True), (False, True), (False, True), (False, True)):
        if refining and strategy is None:
            continue
        p = plan_step(rasterizer="gut", fused_l2=True, world=world, force_collectives=force, sh_sharded=sharded, shard_rows=1000 if sharded else 0, n_views=n_views,
                      loss=loss, strategy=strategy, refining=refining, iteration=iteration, has_shN=has_shN, optimizer_fused=fused, bilateral=bilateral,
                      inline_shN_adam=i_shN, inline_all_adam=i_all, cxx_step=cxx, batch_views=batch)
        n += 1
        assert p.path in ("cxx_all", "cxx_views", "batch_views", "py_views")
        assert p.multi == (world > 1 or force)
        assert p.skip_deferred == (iteration <= 1000 or sharded)
        if p.inline_shN or p.inline_all or p.inline_shard:   # an inline update only when Adam would have read the tensor, and never around a refinement
            assert iteration > 1000 and has_shN and fused and i_shN and not refining
        if p.inline_shN or p.inline_all:                     # replicated layout: only without a strategy, or MCMC between refinements (post_backward = noise on the means)
            assert strategy in (None, "mcmc")
        if p.inline_all:
            assert p.inline_shN and loss == "mse" and strategy is None and not bilateral and not p.multi and not sharded and n_views == 1 and i_all
        if p.inline_shN and p.path != "batch_views":
            assert not p.multi and not sharded and n_views == 1        # a single view's gradient, no all-reduce: nobody else needs the shN gradient tensor
        if p.inline_shN and p.path == "batch_views":
            assert not p.multi and not sharded
        if p.inline_shard:
            assert sharded and n_views == 1 and p.path == "py_views"
        assert not (p.inline_shard and p.inline_shN)
        if p.path == "cxx_all":
            assert cxx and p.inline_all
        if p.path == "cxx_views":
            assert cxx and not sharded and (p.multi or n_views == 1) and not p.inline_all
        if p.path == "batch_views":
            assert n_views > 1 and not p.multi and not sharded and batch
        if sharded:
            assert p.path == "py_views"
    assert n > 20000
