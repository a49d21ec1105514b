"""CPU: a short fixed-seed pass of the two differential shape fuzzers (tools/fuzz_emulated.py --oracle, tools/fuzz_emulated_fastgs.py) - random small problems through the
3DGUT operator chain and through the fastgs rasterizer on the EMULATED product library (tests/emul_util.py), every stage compared with the CPU oracle at the bars of
the GPU parity tests. The long runs are recorded in profiles/r04/fuzz_emulated.txt; this keeps the tools themselves under test."""
import importlib.util
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import emul_util  # noqa: E402


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def emulated():
    if not emul_util.available():
        pytest.skip("no clang++ to build the emulated library")
    with emul_util.installed() as lib:
        yield lib


def test_gut_chain_fuzz_cases_match_the_oracle(emulated, oracle_mod):
    import lichtfeld_studio_amd as lfs
    from lichtfeld_studio_amd import ops
    fz = _tool("fuzz_emulated")
    rng = np.random.default_rng(2026)
    walked = 0
    for i in range(25):
        _, k = fz.one_case(rng, lfs, ops, emulated, i, oracle_mod)
        walked += k
    assert walked > 1000, walked                                   # not vacuous: tile lists were walked
    assert fz.STATS["intersect_tile / intersect_offset: bit-exact cases"][0] >= 10
    assert any(k.startswith("raster bwd: v_means rel-L2") for k in fz.STATS)


def test_fastgs_fuzz_cases_match_the_oracle(emulated, oracle_mod):
    from lichtfeld_studio_amd import fastgs
    fz = _tool("fuzz_emulated_fastgs")
    rng = np.random.default_rng(2027)
    inst = 0
    for i in range(30):
        _, k = fz.one_case(rng, fastgs, emulated, oracle_mod, i)
        inst += k
    assert inst > 1000, inst
    assert any(k.startswith("backward: means rel-L2") for k in fz.STATS)


def test_loss_kernels_on_random_image_shapes_match_the_fp64_reference(emulated):
    """fused SSIM (both paddings) and the fused photometric loss on random image shapes from 1 x 1 upwards (narrower than the 11-tap window, not a multiple of the
    16 x 16 workgroup tile, 1 - 4 channels, batches) against tests/ssim_reference.py in fp64 - the comparison and the bars of tests/test_gpu_loss.py"""
    import torch
    import ssim_reference as ref
    from lichtfeld_studio_amd import losses
    rng = np.random.default_rng(2028)
    with emul_util.cuda_requests_served_by_the_cpu():
        for i in range(24):
            B, C = int(rng.integers(1, 3)), int(rng.integers(1, 5))
            H, W = (int(rng.choice([1, 2, 10, 11, 12, 16, 17, int(rng.integers(1, 70))])) for _ in range(2))
            padding = "same" if rng.random() < 0.5 else "valid"
            g = torch.Generator().manual_seed(i)
            a = torch.rand(B, C, H, W, generator=g)
            b = (a + 0.15 * torch.randn(B, C, H, W, generator=g)).clamp(0, 1)
            x = a.to("cuda:0").requires_grad_(True)
            val = losses.fused_ssim(x, b.to("cuda:0"), padding, True)
            val.backward()
            xr = a.double().requires_grad_(True)
            croppable = H > 10 and W > 10
            if padding == "valid" and not croppable:
                # reference quirk (fused_ssim.cuh:88-98): too small to crop -> mean over the un-cropped map, zeros back
                assert float(x.grad.abs().max()) == 0.0, (B, C, H, W, padding)
                continue
            vr = ref.fused_ssim(xr, b.double(), padding)
            vr.backward()
            assert abs(float(val.detach()) - float(vr.detach())) < 2e-6, (B, C, H, W, padding, float(val.detach()), float(vr.detach()))
            assert float((x.grad.double() - xr.grad).abs().max()) < 1e-5 * float(xr.grad.abs().max()) + 1e-12, (B, C, H, W, padding)
        for i in range(12):
            H, W = (int(rng.choice([1, 10, 11, 17, int(rng.integers(1, 70))])) for _ in range(2))
            lam = float(rng.choice([0.0, 0.2, 1.0]))
            g = torch.Generator().manual_seed(100 + i)
            render = torch.rand(1, H, W, 3, generator=g) * 1.4 - 0.2
            target = torch.rand(3, H, W, generator=g)
            loss = torch.zeros(1, device="cuda:0")
            v = losses.photometric_loss_fwd_bwd(render.to("cuda:0"), target.to("cuda:0"), lam, 0.5, loss)
            r = render.double().requires_grad_(True)
            img = torch.clamp(r[0].permute(2, 0, 1), 0, 1).unsqueeze(0)
            lr = 0.5 * ref.photometric_loss(img, target.double().unsqueeze(0), lam)
            if H > 10 and W > 10:
                lr.backward()
            else:
                (0.5 * (1 - lam) * (img - target.double().unsqueeze(0)).abs().mean()).backward()
            assert abs(float(loss) - float(lr.detach())) < 2e-6, (H, W, lam, float(loss), float(lr.detach()))
            assert float((v.double() - r.grad).abs().max()) < 1e-5 * float(r.grad.abs().max()) + 1e-12, (H, W, lam)
