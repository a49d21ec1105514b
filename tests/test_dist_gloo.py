"""CPU, world_size 2 over gloo: the N>1 path — disjoint view shards, ONE all-reduce of the flat
gradient bucket, identical Adam step on every rank => parameters stay bit-identical across ranks and
equal the single-process result on the concatenated view batch (up to fp32 summation order).
The HIP kernels cannot run here; the per-view gradient is a deterministic stand-in and the Adam
arithmetic comes from the oracle (the checker), which is exactly what is being verified: the
data-parallel plumbing of lichtfeld-studio_amd/dist.py."""
import os
import sys

import numpy as np
import torch

from port_util import free_port
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_grads(params, view):
    g = torch.Generator().manual_seed(1000 + view)
    return [torch.randn(p.shape, generator=g) * (1 + 0.1 * i) for i, p in enumerate(params)]


def _adam_all(oracle, params, states, grads, t):
    import math
    out = []
    for p, (m, v), g in zip(params, states, grads):
        pn, mn, vn = oracle.adam_step(p.numpy(), m, v, g.numpy().astype(np.float32), 1e-2, 0.9, 0.999, 1e-15,
                                      1 / (1 - 0.9 ** t), 1 / math.sqrt(1 - 0.999 ** t))
        out.append((torch.from_numpy(pn), (mn, vn)))
    return out


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import oracle
    import lichtfeld_studio_amd  # noqa: F401
    from lichtfeld_studio_amd import dist as ld
    r, w, _ = ld.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    params = [torch.randn(50, 3, generator=g), torch.randn(50, 15, 3, generator=g), torch.randn(50, generator=g)]
    states = [(np.zeros(p.shape, np.float32), np.zeros(p.shape, np.float32)) for p in params]
    bucket = ld.GradBucket(params)
    for step in range(3):
        views = ld.views_for_step(step, rank, world, 8, views_per_rank=2)
        grads = [torch.zeros_like(p) for p in params]
        for v in views:
            for a, b in zip(grads, _fake_grads(params, v)):
                a += b
        bucket.gather(grads)
        bucket.all_reduce()
        res = _adam_all(oracle, params, states, [x.clone() for x in bucket.views], step + 1)
        params = [x[0] for x in res]
        states = [x[1] for x in res]
    t = ld.max_over_ranks(float(rank), torch.device("cpu"))
    ld.barrier()
    q.put((rank, [p.numpy() for p in params], t))
    dist.destroy_process_group()


def test_dp2_matches_single_process_and_ranks_stay_identical():
    import oracle
    oracle.lib()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, p0, t0), (_, p1, t1) = results
    assert t0 == 1.0 and t1 == 1.0  # max over ranks
    for a, b in zip(p0, p1):
        assert np.array_equal(a, b), "replicas diverged"
    # single process on the concatenated batch
    sys.path.insert(0, ROOT)
    from lichtfeld_studio_amd import dist as ld
    g = torch.Generator().manual_seed(0)
    params = [torch.randn(50, 3, generator=g), torch.randn(50, 15, 3, generator=g), torch.randn(50, generator=g)]
    states = [(np.zeros(p.shape, np.float32), np.zeros(p.shape, np.float32)) for p in params]
    for step in range(3):
        views = [v for r in range(2) for v in ld.views_for_step(step, r, 2, 8, 2)]
        grads = [torch.zeros_like(p) for p in params]
        for v in views:
            for a, b in zip(grads, _fake_grads(params, v)):
                a += b
        res = _adam_all(oracle, params, states, grads, step + 1)
        params, states = [x[0] for x in res], [x[1] for x in res]
    for a, b in zip(p0, params):
        np.testing.assert_allclose(a, b.numpy(), rtol=1e-5, atol=1e-6)


def _worker_deferred(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import lichtfeld_studio_amd  # noqa: F401
    from lichtfeld_studio_amd import dist as ld
    ld.init_distributed(backend="gloo")
    params = [torch.zeros(7, 3), torch.zeros(7, 15, 3), torch.zeros(7), torch.zeros(7, 4)]
    bucket = ld.GradBucket(params, deferred=[1])
    assert bucket.active_numel == 7 * 3 + 7 + 7 * 4 and bucket.flat.numel() == sum(p.numel() for p in params)
    assert all(v.shape == p.shape for v, p in zip(bucket.views, params))
    grads = [torch.full_like(p, float(rank + 1) * (i + 1)) for i, p in enumerate(params)]
    bucket.gather(grads)
    bucket.all_reduce(skip_deferred=True)
    part = [v.clone() for v in bucket.views]
    bucket.gather(grads)
    bucket.all_reduce(skip_deferred=False)
    full = [v.clone() for v in bucket.views]
    q.put((rank, [x.numpy() for x in part], [x.numpy() for x in full]))
    dist.destroy_process_group()


def test_deferred_bucket_segment_is_left_out_of_the_collective():
    """The higher-degree SH gradient (not read by Adam while iteration <= 1000) sits at the end of the flat bucket and is
    skipped by all_reduce(skip_deferred=True); everything else is summed over the ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker_deferred, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, part, full in results:
        for i in range(4):
            total = 3.0 * (i + 1)               # (1 + 2) * (i + 1)
            local = float(rank + 1) * (i + 1)
            assert np.all(full[i] == total)
            assert np.all(part[i] == (local if i == 1 else total))


# ---------------------------------------------------------------------------------------------------------------------
# SH-sharded data parallelism (dist.ShExchange): the exchange logic with the oracle's SH as the stand-in kernels
# ---------------------------------------------------------------------------------------------------------------------
def _sh_standins(oracle):
    """CPU stand-ins with the signatures of fused.sh_model_fwd / sh_model_bwd (what lfs_sh_model_fwd / _bwd compute)."""
    def campos(viewmat):
        m = viewmat.reshape(4, 4).double().numpy()
        return -(m[:3, :3].T @ m[:3, 3])

    def fwd(deg, means, viewmat, sh0, shN, radii):
        dirs = means.double().numpy() - campos(viewmat)
        mask = (radii.reshape(-1, 2).numpy() > 0).all(-1)
        c = oracle.spherical_harmonics_fwd(deg, dirs, torch.cat([sh0, shN], 1).double().numpy(), mask, dtype=np.float64)
        return torch.from_numpy(np.maximum(c + 0.5, 0.0)).float()

    def bwd(deg, means, viewmat, sh0, shN, radii, colors, v_colors, v_sh0, v_shN, v_means, accumulate):
        dirs = means.double().numpy() - campos(viewmat)
        mask = (radii.reshape(-1, 2).numpy() > 0).all(-1)
        vc = v_colors.double().numpy() * (colors.numpy() > 0)          # clamp_min backward
        v_coeffs, v_dirs = oracle.spherical_harmonics_bwd(deg, dirs, torch.cat([sh0, shN], 1).double().numpy(), mask, vc, True, dtype=np.float64)
        v_coeffs, v_dirs = torch.from_numpy(v_coeffs).float(), torch.from_numpy(v_dirs).float()
        if accumulate:
            v_sh0 += v_coeffs[:, :1]; v_shN += v_coeffs[:, 1:]
        else:
            v_sh0.copy_(v_coeffs[:, :1]); v_shN.copy_(v_coeffs[:, 1:])
        v_means += v_dirs
    def fwd_views(deg, means, viewmats, sh0, shN, radii_views):
        V, S, n = radii_views.shape[0], radii_views.shape[1], means.shape[0]
        out = torch.zeros(V, S, 3)
        for v in range(V):
            out[v, :n] = fwd(deg, means, viewmats[v], sh0, shN, radii_views[v, :n])
        return out

    def bwd_views(deg, means, viewmats, sh0, shN, radii_views, colors_views, v_colors_views, v_sh0, v_shN, v_means, accumulate, adam=None):
        assert adam is None
        n = means.shape[0]
        for v in range(radii_views.shape[0]):
            bwd(deg, means, viewmats[v], sh0, shN, radii_views[v, :n], colors_views[v, :n], v_colors_views[v, :n], v_sh0, v_shN, v_means, accumulate or v > 0)
    return fwd, bwd, fwd_views, bwd_views


def _sh_problem(world, N=203):
    g = torch.Generator().manual_seed(5)
    K, deg = 9, 2                                          # N not divisible by the world size: the last shard is short
    means = torch.randn(N, 3, generator=g)
    sh0, shN = torch.randn(N, 1, 3, generator=g), torch.randn(N, K - 1, 3, generator=g)
    viewmats, radii, v_colors = [], [], []
    for j in range(world):
        m = torch.eye(4)
        m[:3, 3] = torch.randn(3, generator=g) * 3
        viewmats.append(m[None])
        radii.append((torch.rand(N, 2, generator=g) > 0.3).int() * 5)      # ~half of the Gaussians visible per view
        v_colors.append(torch.randn(N, 3, generator=g))
    return N, K, deg, means, sh0, shN, viewmats, radii, v_colors


def _sh_worker(rank, world, port, q, N=203):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import oracle
    import lichtfeld_studio_amd  # noqa: F401
    from lichtfeld_studio_amd import dist as ld
    ld.init_distributed(backend="gloo")
    _, _, fwd, bwd = _sh_standins(oracle)
    N, K, deg, means, sh0, shN, viewmats, radii, v_colors = _sh_problem(world, N)
    ex = ld.ShExchange(N, world, rank)
    shN_shard = ex.shard(shN).clone()
    # two sub-steps (views_per_rank = 2): the second accumulates
    g_sh0, g_shN, g_means = torch.zeros(N, 1, 3), torch.zeros(ex.n, K - 1, 3), torch.zeros(N, 3)
    colors_seen = []
    for k in range(2):
        vm = [viewmats[(j + k) % world] for j in range(world)]
        my = (rank + k) % world
        colors, ctx = ex.forward(deg, means, sh0, shN_shard, radii[my], vm, fwd)
        colors_seen.append(colors)
        g_means += 0.0 if k else 1.0                        # stands for the rasterizer's own dL/dmeans (copy_ / add_ happen before the SH backward)
        ex.backward(ctx, deg, means, sh0, shN_shard, vm, v_colors[my] * (k + 1), g_sh0, g_shN, g_means, k > 0, bwd)
    bucket = ld.GradBucket([g_means, g_sh0])
    bucket.gather([g_means, g_sh0])
    bucket.all_reduce()
    full = ex.gather_rows(g_shN)
    q.put((rank, [c.numpy() for c in colors_seen], bucket.views[0].numpy().copy(), bucket.views[1].numpy().copy(), full.numpy(), (ex.r0, ex.r1)))
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world,N", [(2, 203), (3, 5), (4, 3), (8, 29)])      # uneven shards; more ranks than rows per shard; a rank that owns nothing; 8 ranks
def test_sh_sharded_exchange_matches_replicated_computation(world, N):
    import oracle
    oracle.lib()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_sh_worker, args=(r, world, port, q, N)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # replicated reference: every rank evaluates SH for its own views on the full tensors, gradients summed over ranks
    fwd, bwd, _, _ = _sh_standins(oracle)
    N, K, deg, means, sh0, shN, viewmats, radii, v_colors = _sh_problem(world, N)
    ref_sh0, ref_shN, ref_means = torch.zeros(N, 1, 3), torch.zeros(N, K - 1, 3), torch.full((N, 3), float(world))
    for rank in range(world):
        for k in range(2):
            my = (rank + k) % world
            c = fwd(deg, means, viewmats[my], sh0, shN, radii[my])
            vis = (radii[my] > 0).all(-1)
            assert np.allclose(results[rank][1][k][vis.numpy()], c.numpy()[vis.numpy()], atol=1e-6)      # colours of the visible Gaussians
            bwd(deg, means, viewmats[my], sh0, shN, radii[my], c, v_colors[my] * (k + 1), ref_sh0, ref_shN, ref_means, True)
    S = (N + world - 1) // world
    assert [r[5] for r in results] == [(min(j * S, N), min(min(j * S, N) + S, N)) for j in range(world)]
    if (world, N) == (4, 3):
        assert results[3][5] == (3, 3)                       # the fourth rank owns no rows and still takes part in every collective
    for rank in range(world):
        assert np.allclose(results[rank][2], ref_means.numpy(), atol=2e-5), "means gradient (rasterizer part + SH view-direction part)"
        assert np.allclose(results[rank][3], ref_sh0.numpy(), atol=2e-5)
        assert np.allclose(results[rank][4], ref_shN.numpy(), atol=2e-5), "all-gathered shard gradients == replicated shN gradient"
        assert np.array_equal(results[0][4], results[rank][4])


def _worker_early(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import lichtfeld_studio_amd  # noqa: F401
    from lichtfeld_studio_amd import dist as ld
    ld.init_distributed(backend="gloo")
    ld.stats_enable(False)
    N = 11
    params = [torch.zeros(N, 3), torch.zeros(N, 1, 3), torch.zeros(N, 15, 3), torch.zeros(N, 3), torch.zeros(N, 4), torch.zeros(N)]   # the model's parameter order
    g = torch.Generator().manual_seed(5 + rank)
    grads = [torch.randn(p.shape, generator=g) for p in params]
    out = {}
    for skip in (True, False):
        a = ld.GradBucket(params, deferred=[2]); a.gather(grads); a.all_reduce(skip_deferred=skip)
        b = ld.GradBucket(params, deferred=[2]); b.gather(grads)
        b.all_reduce_early([3, 4, 5])          # scales, quats, opacities: contiguous [6N, 14N) of the flat buffer
        b.all_reduce(skip_deferred=skip)
        out[skip] = ([v.clone().numpy() for v in a.views], [v.clone().numpy() for v in b.views])
    # the replicated data-parallel step of round 3: the shN segment [14N, 59N) first, in 4 chunks (45 N = 495 floats -> 124 + 124 + 124 + 123), then the rest
    c = ld.GradBucket(params, deferred=[2]); c.gather(grads)
    c.all_reduce_early([2], chunks=4)
    c.all_reduce(skip_deferred=False)
    out["shN_first"] = [v.clone().numpy() for v in c.views]
    st = ld.stats_collect()
    q.put((rank, out, st, ld.ranks_seen(torch.device("cpu"))))
    dist.destroy_process_group()


def test_early_segment_all_reduce_equals_the_single_collective():
    """GradBucket.all_reduce_early([scales, quats, opacities]) + all_reduce() (what the trainer does so that the geometry gradients travel
    while the SH backward runs) gives exactly the sums of one all_reduce() over the same bucket; the deferred shN segment stays local when
    skipped; the per-collective accounting counts three collectives and the right payload."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker_early, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
    for rank, out, st, seen in res:
        assert seen == 2
        for skip in (True, False):
            single, split = out[skip]
            for x, y in zip(single, split):
                assert np.array_equal(x, y)
        for x, y in zip(out[False][0], out["shN_first"]):
            assert np.array_equal(x, y)
        assert st["all_reduce_early"]["calls"] == 2 + 4 and st["all_reduce_early"]["bytes"] == 2 * 4 * 11 * 8 + 4 * 11 * 45
        assert st["all_reduce"]["calls"] == 2 + 1 + 2 + 1   # two single collectives; the split ones: [0,6N) (+ [14N, 59N) when not skipped); [0,14N) behind the shN chunks
    assert np.array_equal(res[0][1][False][1][2], res[1][1][False][1][2])            # shN summed over both ranks
    assert not np.array_equal(res[0][1][True][1][2], res[1][1][True][1][2])          # ... and left local when deferred


# ---- the factored exchange of the replicated layout (dist.ColorGradExchange, round 4) ------------------------------------------------------------------------
def _factored_worker(rank, world, port, q, N, vpr):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import oracle
    import lichtfeld_studio_amd  # noqa: F401
    from lichtfeld_studio_amd import dist as ld
    ld.init_distributed(backend="gloo")
    fwd, bwd, _, bwd_views = _sh_standins(oracle)
    N, K, deg, means, sh0, shN, viewmats, radii, v_colors = _sh_problem(world * vpr, N)
    ex = ld.ColorGradExchange(N, world, rank, vpr, "cpu")
    # what trainer._step_cxx_factored does per view: the rendering rank masks its dL/dcolour rows with visibility (rows of invisible Gaussians are zero: the
    # rasterizer never touched them) and with the clamp of ITS colours; the rasterizer's own dL/dmeans goes into the bucket
    g_means, g_other = torch.zeros(N, 3), torch.zeros(N, 8)
    for k in range(vpr):
        my = rank * vpr + k
        c = fwd(deg, means, viewmats[my], sh0, shN, radii[my])
        vis = (radii[my] > 0).all(-1, keepdim=True)
        ex.send[k] = v_colors[my] * vis * (c > 0)
        g_means += 1.0 + rank          # stands for the rasterizer part of dL/dmeans of this view
        g_other += float(my + 1)       # ... and for scales / quaternions / opacities
    bucket = ld.GradBucket([g_means, torch.zeros(N, 1, 3), torch.zeros(N, K - 1, 3), g_other], deferred=[1, 2])
    bucket.gather([g_means, None, None, g_other])
    rows = ex.gather()
    bucket.all_reduce_early([0, 3])
    vms = torch.cat([viewmats[j] for j in range(world * vpr)])            # rank-major: view k of rank r at r * vpr + k
    ones = torch.ones(world * vpr, N, 2, dtype=torch.int32)              # the pre-masked form needs neither radii nor colours: all-visible, all-positive stand-ins
    ex.v_dirs.zero_()
    bwd_views(deg, means, vms, sh0, shN, ones, torch.ones(world * vpr, N, 3), rows, bucket.views[1], bucket.views[2], ex.v_dirs, False)
    bucket.all_reduce(skip_deferred=True)
    bucket.views[0].add_(ex.v_dirs)
    q.put((rank, [v.numpy().copy() for v in bucket.views]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,N,vpr", [(2, 203, 1), (3, 57, 2), (4, 31, 1), (8, 41, 1)])   # (8 ranks x 1 view: what `bench.py --gpus 8` runs)
def test_factored_color_gradient_exchange_matches_replicated_computation(world, N, vpr):
    """Replicated layout, two ways to get the SH gradients onto every rank: (reference) each rank runs the SH backward of its own views and the [N,K,3] gradients
    are summed over the ranks - what the flat all-reduce does; (factored) the ranks all-gather clamp-masked dL/dcolour rows and each runs the multi-view backward
    over ALL views. Same sh0 / shN gradients and the same dL/dmeans (rasterizer part through the all-reduce + direction term from the multi-view pass) up to
    summation order, and BIT-identical across the ranks."""
    import oracle
    oracle.lib()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_factored_worker, args=(r, world, port, q, N, vpr)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    fwd, bwd, _, _ = _sh_standins(oracle)
    N, K, deg, means, sh0, shN, viewmats, radii, v_colors = _sh_problem(world * vpr, N)
    ref_sh0, ref_shN, ref_means, ref_other = torch.zeros(N, 1, 3), torch.zeros(N, K - 1, 3), torch.zeros(N, 3), torch.zeros(N, 8)
    for rank in range(world):
        for k in range(vpr):
            my = rank * vpr + k
            c = fwd(deg, means, viewmats[my], sh0, shN, radii[my])
            bwd(deg, means, viewmats[my], sh0, shN, radii[my], c, v_colors[my], ref_sh0, ref_shN, ref_means, True)
            ref_means += 1.0 + rank
            ref_other += float(my + 1)
    for rank in range(world):
        g_means, g_sh0, g_shN, g_other = results[rank][1]
        assert np.allclose(g_means, ref_means.numpy(), atol=2e-5) and np.allclose(g_sh0, ref_sh0.numpy(), atol=2e-5) and np.allclose(g_shN, ref_shN.numpy(), atol=2e-5)
        assert np.array_equal(g_other, ref_other.numpy())
        for a, b in zip(results[0][1], results[rank][1]):
            assert np.array_equal(a, b), "replicas must stay bit-identical"
    assert float(np.abs(ref_shN.numpy()).max()) > 0


# ---------------------------------------------------------------------------------------------------------------------
# dist.agree_all: the bench's layout agreement travels over its own gloo side group
# ---------------------------------------------------------------------------------------------------------------------
def _worker_agree(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LFS_DIST_TIMEOUT_S="120")
    import lichtfeld_studio_amd  # noqa: F401
    from lichtfeld_studio_amd import dist as ld
    ld.init_distributed(backend="gloo")
    out = [ld.agree_all(True), ld.agree_all(rank != 1), ld.agree_all(rank == 1), ld.agree_all(True)]
    q.put((rank, out, ld._AGREE_GROUP is not None and ld._AGREE_GROUP is not dist.group.WORLD))
    dist.destroy_process_group()


def test_layout_agreement_runs_over_its_own_group_and_is_the_minimum_over_ranks():
    """One rank saying no is everybody's no; the answer is the same on every rank; the side group exists after the first call and is not the default group."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker_agree, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in range(3)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, own_group in results:
        assert out == [True, False, False, True], (rank, out)
        assert own_group
