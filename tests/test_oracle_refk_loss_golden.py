"""CPU: the restatements used as oracle for the loss-side kernels - tests/ssim_reference.py (torch conv2d) and oracle/bilateral.py (numpy) - against golden
vectors generated from the REFERENCE'S OWN kernels run on the CPU (tests/golden/refk_loss.npz; oracle/make_golden_refk_loss.py runs the kernels of
src/training/kernels/ssim.cu and bilateral_grid_{forward,backward,tv}.cu, compiled in place as host code under oracle/ref_emul/). This pins SURVEY.md §8f
row 2 to the reference itself; the same file checks the HIP kernels in tests/test_gpu_refk_golden.py.
Tolerances (fp32 on the reference side, fp64 restatement unless noted): SSIM map 2e-6 absolute, derivative maps and dL/dimg1 2e-5 of their maximum;
bilateral slice 2e-6, its gradients 2e-5 of their maximum (grad_grid sums thousands of atomics in pixel order); TV loss 1e-5 relative."""
import os

import numpy as np
import pytest
import torch

import ssim_reference as ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refk_loss.npz")
C1, C2 = 0.01 ** 2, 0.03 ** 2


def cases():
    z = np.load(GOLD)
    out = {}
    for k in z.files:
        name, key = k.split("/", 1)
        out.setdefault(name, {})[key] = z[k]
    return out


CASES = cases()
SSIM = sorted(k for k in CASES if k.startswith("ssim_"))
SLICE = sorted(k for k in CASES if k.startswith("slice_"))
TV = sorted(k for k in CASES if k.startswith("tv_"))


def close(got, want, rel, what):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape and np.isfinite(got).all(), what
    err, scale = np.abs(got - want).max(), max(np.abs(want).max(), 1e-30)
    assert err <= rel * scale, (what, err, scale)
    return err / scale


def ssim_derivative_maps(a, b):
    """dm/dmu1, dm/dsigma1_sq, dm/dsigma12 per pixel from the closed form (ssim.cu:200-221 states the same expressions), in fp64."""
    mu1, mu2 = ref._blur(a), ref._blur(b)
    s1, s2, s12 = ref._blur(a * a) - mu1 * mu1, ref._blur(b * b) - mu2 * mu2, ref._blur(a * b) - mu1 * mu2
    A, B, Cc, D = mu1 * mu1 + mu2 * mu2 + C1, s1 + s2 + C2, 2 * mu1 * mu2 + C1, 2 * s12 + C2
    d_mu1 = (mu2 * 2 * D) / (A * B) - (mu2 * 2 * Cc) / (A * B) - (mu1 * 2 * Cc * D) / (A * A * B) + (mu1 * 2 * Cc * D) / (A * B * B)
    return d_mu1, -(Cc * D) / (A * B * B), (2 * Cc) / (A * B)


@pytest.mark.parametrize("name", SSIM)
def test_ssim_restatement_matches_the_reference_kernel(name):
    d = CASES[name]
    a, b = torch.from_numpy(d["img1"]).double(), torch.from_numpy(d["img2"]).double()
    close(ref.ssim_map(a, b).numpy(), d["ssim_map"], 2e-6, "ssim_map")
    for got, key in zip(ssim_derivative_maps(a, b), ("dm_dmu1", "dm_dsigma1_sq", "dm_dsigma12")):
        close(got.numpy(), d[key], 2e-5, key)
    x = a.clone().requires_grad_(True)
    (ref.ssim_map(x, b) * torch.from_numpy(d["dL_dmap"]).double()).sum().backward()
    close(x.grad.numpy(), d["dL_dimg1"], 2e-5, "dL_dimg1")


@pytest.mark.parametrize("name", SLICE)
def test_bilateral_slice_restatement_matches_the_reference_kernel(name):
    from oracle import bilateral
    d = CASES[name]
    close(bilateral.slice_forward(d["grid"], d["rgb"], np.float64), d["output"], 2e-6, "slice output")
    gg, gr = bilateral.slice_backward(d["grid"], d["rgb"], d["grad_output"], np.float64)
    close(gg, d["grad_grid"], 2e-5, "grad_grid")
    close(gr, d["grad_rgb"], 2e-5, "grad_rgb")


@pytest.mark.parametrize("name", TV)
def test_bilateral_tv_restatement_matches_the_reference_kernel(name):
    from oracle import bilateral
    d = CASES[name]
    assert abs(float(bilateral.tv_forward(d["grids"], np.float64)) - float(d["tv_loss"])) <= 1e-5 * float(d["tv_loss"])
    close(bilateral.tv_backward(d["grids"], float(d["grad_output"]), np.float64), d["grad_grids"], 2e-6, "tv grad")


def test_golden_file_regenerates_from_the_reference_kernels():
    """Where oracle/_ref/libref_loss.so exists (the build container), the reference's kernels reproduce the committed file bit for bit."""
    import oracle
    if oracle.refk_loss_lib() is None:
        pytest.skip("oracle/_ref/libref_loss.so not built (needs /root/reference)")
    for name in SSIM:
        d = CASES[name]
        m, d1, d2, d3 = oracle.refk_fusedssim(C1, C2, d["img1"], d["img2"], True)
        g = oracle.refk_fusedssim_backward(C1, C2, d["img1"], d["img2"], d["dL_dmap"], d1, d2, d3)
        for got, key in zip((m, d1, d2, d3, g), ("ssim_map", "dm_dmu1", "dm_dsigma1_sq", "dm_dsigma12", "dL_dimg1")):
            assert np.array_equal(got, d[key]), (name, key)
    for name in SLICE:
        d = CASES[name]
        o, gg, gr = oracle.refk_bilateral_slice(d["grid"], d["rgb"], d["grad_output"])
        assert np.array_equal(o, d["output"]) and np.array_equal(gr, d["grad_rgb"])
        close(gg, d["grad_grid"], 1e-6, "grad_grid")      # atomics: the fibers' order is fixed, but keep the bar meaningful if it ever is not
    for name in TV:
        d = CASES[name]
        loss, gg = oracle.refk_bilateral_tv(d["grids"], float(d["grad_output"]))
        assert np.float32(loss) == d["tv_loss"] and np.array_equal(gg, d["grad_grids"])
