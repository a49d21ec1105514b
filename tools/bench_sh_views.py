#!/usr/bin/env python3
"""GPU: time of the multi-view SH backward with shN's Adam update inside (lfs_sh_model_bwd_views - what every rank of the factored data-parallel layout runs over the
views of ALL ranks) at SYN-B size for 1 / 2 / 4 / 8 views, and a checksum of its outputs (two builds of the kernel must agree bit for bit: LFS_GSPLAT_LIB selects one).
    python tools/bench_sh_views.py"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lichtfeld_studio_amd as lfs  # noqa: E402
from lichtfeld_studio_amd import fused, scenes  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    sc = scenes.syn_b(n_views=8).to(dev)
    N, K = sc.N, 16
    g = torch.Generator(device=dev).manual_seed(3)
    print("library:", lfs.load_library().lfs_version().decode())
    for V in (1, 2, 4, 8):
        rows = torch.randn(V, N, 3, device=dev, generator=g) * 1e-3
        rows[:, ::17] = 0.0                                       # invisible / clamped rows
        vms = sc.viewmats[:V].contiguous()
        state = lambda: (sc.shN.clone(), torch.zeros_like(sc.shN), torch.zeros_like(sc.shN), torch.zeros(N, 1, 3, device=dev), torch.zeros(N, 3, device=dev))
        adam = lambda m, v: dict(exp_avg=m, exp_avg_sq=v, lr=1.25e-4, beta1=0.9, beta2=0.999, eps=1e-15, bc1_rcp=1.0 / (1 - 0.9 ** 7), bc2_sqrt_rcp=1.0 / (1 - 0.999 ** 7) ** 0.5)
        shN, m, v, v_sh0, v_means = state()
        fused.sh_model_bwd_views(3, sc.means, vms, sc.sh0, shN, None, None, rows, v_sh0, None, v_means, False, adam=adam(m, v))
        torch.cuda.synchronize()
        h = hashlib.sha1()
        for t in (shN, m, v, v_sh0, v_means):
            h.update(t.cpu().numpy().tobytes())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for rep in range(6):
            shN, m, v, v_sh0, v_means = state()
            torch.cuda.synchronize()
            e0.record()
            fused.sh_model_bwd_views(3, sc.means, vms, sc.sh0, shN, None, None, rows, v_sh0, None, v_means, False, adam=adam(m, v))
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"views {V}: {np.median(ts[1:]):.4f} ms (median of 5), outputs sha1 {h.hexdigest()[:16]}")


if __name__ == "__main__":
    main()
