"""GPU: short version of tests/convergence_check.py — the HIP path and the CPU oracle (restatement of the reference kernels),
trained with the same recipe from the same start, reach the same PSNR (north star: within 0.05 dB)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_hip_and_oracle_training_reach_the_same_psnr(lfs, oracle_mod):
    import convergence_check as cc
    dev = torch.device("cuda:0")
    gt, init = cc.make_task(n=1500, size=96, n_views=4, sh_degree=1)
    targets = cc.render_views_hip(gt, dev)
    start = float(np.mean([cc.psnr(a.cpu().numpy(), b.cpu().numpy()) for a, b in zip(cc.render_views_hip(init, dev), targets)]))
    iters = 150
    hip, _ = cc.train_hip(init, targets, iters, 7000, dev, iters)
    ora = cc.train_oracle(init, [t.cpu().numpy() for t in targets], iters, 7000, iters)
    p_hip, p_ora = hip[iters], ora[iters]
    assert p_hip > start + 0.5, (start, p_hip)            # training works
    assert abs(p_hip - p_ora) < 0.05, (p_hip, p_ora)      # and lands where the reference algorithm lands
