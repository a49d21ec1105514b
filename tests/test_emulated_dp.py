"""The data-parallel training step with the REAL kernels' source on the CPU: the bodies of tests/test_gpu_dp2.py (ranks spawned as processes, gloo collectives, all three
layouts - replicated with the flat all-reduce, replicated with the factored SH exchange, SH-sharded - against a single process that renders the same global batch, replicas
bit-identical across ranks) executed with the product library compiled as host code on the wavefront emulator (tests/emul_util.py) in every rank. tests/test_dist_gloo.py
checks the exchanges with oracle stand-ins for the kernels; this file puts the product's own step (csrc/gut_step.hip's split forms, the multi-view SH kernels with Adam inside,
dist.py's staging) under them. What it cannot show is RCCL itself: no collective has run between two GPUs in this repository."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import emul_util  # noqa: E402
import test_gpu_dp2 as dp2  # noqa: E402


def _emulated_worker(*args):
    """a rank: the same worker as on the GPU box, inside the emulation context (spawned process: nothing is inherited from the parent's patches)"""
    sys.path.insert(0, HERE)
    with emul_util.installed(), emul_util.cuda_requests_served_by_the_cpu():
        dp2._worker(*args)


def _emulated_mcmc_worker(*args):
    sys.path.insert(0, HERE)
    with emul_util.installed(), emul_util.cuda_requests_served_by_the_cpu():
        dp2._real_mcmc_worker(*args)


@pytest.fixture(scope="module", autouse=True)
def _emulated():
    if not emul_util.available():
        pytest.skip("no clang++ to build the emulated library")
    emul_util.build()   # once, before the ranks race for it
    saved = dp2._worker, dp2._mcmc_worker
    dp2._real_mcmc_worker = dp2._mcmc_worker
    dp2._worker, dp2._mcmc_worker = _emulated_worker, _emulated_mcmc_worker
    try:
        with emul_util.installed(), emul_util.cuda_requests_served_by_the_cpu():
            yield
    finally:
        dp2._worker, dp2._mcmc_worker = saved


# default: the north-star layout (replicated Gaussians, flat all-reduce with the early SH chunks) on two ranks - 50 s; LFS_EMUL_DP_ALL=1 adds the factored exchange on two
# and three ranks and the SH-sharded layout (all four: 3.5 minutes, green at the commit that added this file and at the one that set this default)
_CASES = [(False, 2)] + ([("factored", 2), (True, 2), ("factored", 3)] if os.environ.get("LFS_EMUL_DP_ALL") else [])


@pytest.mark.parametrize("sharded,world", _CASES)
def test_emulated_multi_rank_step_matches_single_process(lfs, sharded, world):
    dp2.test_multi_rank_step_matches_single_process(lfs, sharded, world, "gloo")
