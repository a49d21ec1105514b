#!/usr/bin/env python3
"""The op-level spherical-harmonics kernels (gsplat::spherical_harmonics_fwd / _bwd: explicit dirs, cat(sh0, shN), bool masks) against the model-form kernels of
the fused step on the same 1M-Gaussian scene: HIP-event time per launch. Run on the GPU box: python tools/bench_sh_ops.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lichtfeld_studio_amd as lfs  # noqa: F401
from lichtfeld_studio_amd import fused, ops, scenes

dev = torch.device("cuda:0")
sc = scenes.syn_b().to(dev)
N = sc.means.shape[0]
vm, K = sc.viewmats[0:1].contiguous(), sc.Ks[0:1].contiguous()
quats, scales, opac, radii, m2, d = fused.activations_project(sc.means, sc.raw_quats, sc.raw_scales, sc.raw_opacities, vm, K, sc.width, sc.height, None)
campos = torch.inverse(vm)[:, :3, 3]
dirs = (sc.means - campos).contiguous()
coeffs = torch.cat([sc.sh0, sc.shN], 1).contiguous()
masks = (radii[0] > 0).all(-1).contiguous()
vcol = torch.randn(N, 3, device=dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


res = {"N": N, "visible": int(masks.sum())}
for deg in (3, 0):
    res[f"op_fwd_deg{deg}_ms"] = timed(lambda: ops.spherical_harmonics_fwd(deg, dirs, coeffs, masks))
    res[f"op_bwd_deg{deg}_ms"] = timed(lambda: ops.spherical_harmonics_bwd(16, deg, dirs, coeffs, masks, vcol, True))
    res[f"model_fwd_deg{deg}_ms"] = timed(lambda: fused.sh_model_fwd(deg, sc.means, vm, sc.sh0, sc.shN, radii))
print(json.dumps(res))
