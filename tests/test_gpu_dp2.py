"""GPU, two processes on ONE device over gloo (device tensors staged through the host in dist.py; RCCL refuses two ranks on
one GPU, and the driver's multi-GPU run is the only place with more than one): the data-parallel step end to end with the real HIP
kernels. Both layouts - SH-sharded (default) and fully replicated - must reproduce a single process that renders the same
global batch (views_per_rank = 2) up to fp32 summation order, and keep the replicated parameters bit-identical across ranks. Third layout (round 4):
replicated with the FACTORED exchange of the SH gradients (dist.ColorGradExchange: dL/dcolour rows all-gathered, multi-view SH backward on every rank)."""
import os
import sys

import numpy as np
import pytest
import torch

from port_util import free_port
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 4


def _scene():
    from lichtfeld_studio_amd import scenes
    sc = scenes._syn_box("SYN-DP", 3, 6001, 192, 128, 150.0, 8, sh_degree=2)     # odd N: uneven shards
    sc.raw_scales += float(np.log(6.0))
    return sc


def _worker(rank, world, port, sharded, q, backend="gloo"):
    sys.path.insert(0, ROOT)
    rccl = backend == "nccl"       # one GPU per rank over RCCL (needs >= world devices); gloo: all ranks share cuda:0, collectives staged through the host
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank if rccl else 0))
    import lichtfeld_studio_amd  # noqa: F401
    from lichtfeld_studio_amd import dist as ld, scenes
    from lichtfeld_studio_amd.trainer import GutTrainer
    if rccl:
        torch.cuda.set_device(rank)
    ld.init_distributed(backend=backend)
    dev = torch.device("cuda", rank if rccl else 0)
    sc = _scene()
    # sharded: True = SH-sharded layout (dist.ShExchange), False = replicated with the flat all-reduce, "factored" = replicated with the factored SH exchange
    tr = GutTrainer(sc, dev, iterations=100, world=world, rank=rank, sh_sharded=sharded is True, factored_sh=sharded == "factored")
    tr.iteration = 1000           # past the shN warm-up: Adam updates shN, the replicated layout all-reduces it
    target = scenes.target_image(sc.height, sc.width).to(dev) * 0.6
    losses = [float(tr.train_step([target])) for _ in range(STEPS)]
    params = [p.detach().cpu().numpy() for p in tr.model.parameters()]
    params[2] = tr.full_shN().cpu().numpy()
    full = tr.export_model()
    assert full.shN.shape[0] == full.means.shape[0] and np.array_equal(full.shN.cpu().numpy(), params[2])
    q.put((rank, params, losses, tuple(tr.model.shN.shape)))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("sharded,world,backend", [(True, 2, "gloo"), (False, 2, "gloo"), (True, 4, "gloo"), ("factored", 2, "gloo"), ("factored", 3, "gloo"),
                                                   (False, 2, "nccl"), (True, 2, "nccl"), ("factored", 2, "nccl")])
def test_multi_rank_step_matches_single_process(lfs, sharded, world, backend):
    """backend "nccl" = RCCL with one GPU per rank: self-skips on a box with fewer GPUs than ranks (every box so far), so that the first multi-GPU box
    that runs this suite exercises the real collectives of both layouts - the replicated one with its chunked early all-reduce of the SH gradients."""
    from lichtfeld_studio_amd import scenes
    from lichtfeld_studio_amd.trainer import GutTrainer
    if backend == "nccl" and torch.cuda.device_count() < world:
        pytest.skip(f"RCCL needs one GPU per rank: {torch.cuda.device_count()} device(s) here, {world} ranks")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, sharded, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=600) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single process, same global batch
    dev = torch.device("cuda:0")
    sc = _scene()
    tr = GutTrainer(sc, dev, iterations=100, views_per_rank=world)
    tr.iteration = 1000
    target = scenes.target_image(sc.height, sc.width).to(dev) * 0.6
    ref_losses = [float(tr.train_step([target])) for _ in range(STEPS)]
    ref = [p.detach().cpu().numpy() for p in tr.model.parameters()]
    p0, l0 = results[0][1], results[0][2]
    N = sc.means.shape[0]
    S = (N + world - 1) // world
    assert [r[3][0] for r in results] == ([min(S, N - j * S) for j in range(world)] if sharded is True else [N] * world)     # 6001 rows: 3001 + 3000, or 1501 x 3 + 1498
    for name, a, r in zip(["means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opacities"], p0, ref):
        for other in results[1:]:
            assert np.array_equal(a, other[1][["means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opacities"].index(name)]), f"{name}: ranks diverged"
        # Adam normalises the step: a gradient that differs in the last bits moves a parameter by the same lr - compare the update
        # (elements whose gradient is pure rounding noise may step the other way: allow 0.03 % of them)
        tol = 2e-3 * np.abs(r - _scene_param(sc, name)).max() + 1e-7
        # (measured: 0 .. 1.1e-4 of the elements - three float-atomic trajectories of four Adam steps each; the bar is 3e-4)
        assert (np.abs(a - r) > tol).mean() < 3e-4, (name, np.abs(a - r).max(), tol, float((np.abs(a - r) > tol).mean()))
    assert np.allclose(np.sum([r[2] for r in results], 0), ref_losses, rtol=1e-5)      # each rank reports its share of the loss
    assert float(np.abs(ref[2] - sc.shN.numpy()).max()) > 0                     # shN did train


def _scene_param(sc, name):
    return {"means": sc.means, "sh0": sc.sh0, "shN": sc.shN, "raw_scales": sc.raw_scales, "raw_quats": sc.raw_quats, "raw_opacities": sc.raw_opacities}[name].numpy()


def _mcmc_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import lichtfeld_studio_amd  # noqa: F401
    from lichtfeld_studio_amd import dist as ld, scenes, strategies
    from lichtfeld_studio_amd.trainer import GutTrainer
    ld.init_distributed(backend="gloo")
    dev = torch.device("cuda:0")
    sc = _scene()
    op = strategies.OptimizationParameters(iterations=400, start_refine=1002, refine_every=4, stop_refine=2000, max_cap=7000)
    tr = GutTrainer(sc, dev, iterations=400, world=world, rank=rank, strategy="mcmc", opt_params=op, sh_sharded=True)
    assert tr.sh_exchange is not None, "MCMC supports the SH-sharded layout"
    tr.iteration = 1000
    target = scenes.target_image(sc.height, sc.width).to(dev) * 0.6
    n_seen, losses = [], []
    for _ in range(14):
        losses.append(float(tr.train_step([target])))
        n_seen.append((tr.model.means.shape[0], tr.model.shN.shape[0], tr.sh_exchange.r0, tr.sh_exchange.r1))
    st = tr.optimizer._state(tr.model.shN)
    params = [p.detach().cpu().numpy() for p in tr.model.parameters()]
    params[2] = tr.full_shN().cpu().numpy()
    moments = tr.sh_exchange.gather_rows(st["exp_avg"]).cpu().numpy()
    q.put((rank, params, losses, n_seen, moments, st["step_count"]))
    torch.distributed.destroy_process_group()


def test_two_rank_mcmc_keeps_sh_sharded_layout(lfs):
    """MCMC (relocation + growth + noise) on two SH-sharded ranks: refinement steps gather shN and its moments, run the replicated strategy code and
    re-shard for the new N. Replicated parameters, the gathered shN and its gathered Adam moments stay bit-identical across ranks; shards follow N."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_mcmc_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=600) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, p0, l0, n0, m0, c0), (_, p1, l1, n1, m1, c1) = results
    for name, a, b in zip(["means", "sh0", "shN (gathered)", "raw_scales", "raw_quats", "raw_opacities"], p0, p1):
        assert a.shape == b.shape and np.array_equal(a, b), f"{name}: ranks diverged"
    assert np.array_equal(m0, m1) and c0 == c1 == 14 - 3      # the three refinement steps replace the tensors: no gradient, no optimizer update (as in the reference)
    N0, N_end = 6001, n0[-1][0]
    assert N_end > N0 and N_end <= 7000, "MCMC growth (5 % per refinement, capped)"
    for (N, rows0, a0, b0), (N_, rows1, a1, b1) in zip(n0, n1):
        S = (N + 1) // 2
        assert N == N_ and (a0, b0) == (0, S) and (a1, b1) == (S, N) and rows0 == S and rows1 == N - S      # shards re-cut when N changes
    assert p0[2].shape[0] == N_end and np.isfinite(l0).all() and np.isfinite(l1).all()
    assert float(np.abs(m0).max()) > 0
