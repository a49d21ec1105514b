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


# ---- the host code around the kernels: fused_ssim.cuh's autograd wrapper, compute_photometric_loss, BilateralGrid, WarmupExponentialLR ("host/..." entries,
# oracle/_ref/libref_loss_host.so) ----------------------------------------------------------------------------------------------------------------------------


def host(key):
    return np.load(GOLD)["host/" + key]


@pytest.mark.parametrize("name", ["photometric_40x52", "photometric_small_9x40"])
def test_photometric_loss_restatement_matches_the_reference_host_code(name):
    img, gt = torch.from_numpy(host(f"{name}/rendered")).double(), torch.from_numpy(host(f"{name}/gt")).double()
    x = img.clone().requires_grad_(True)
    loss = ref.photometric_loss(x[None], gt[None], 0.2)
    assert abs(float(loss.detach()) - float(host(f"{name}/loss"))) <= 2e-6
    assert abs(float(ref.fused_ssim(img[None], gt[None], "valid")) - float(host(f"{name}/ssim_valid"))) <= 2e-6
    assert abs(float(ref.fused_ssim(img[None], gt[None], "same")) - float(host(f"{name}/ssim_same"))) <= 2e-6
    loss.backward()
    if img.shape[1] > 10:
        close(x.grad.numpy(), host(f"{name}/grad"), 2e-5, "d loss / d rendered")
    else:
        # fused_ssim.cuh:67-70 vs :92-97: with <= 10 rows the forward keeps the whole map but the backward scatters nothing - the SSIM term has NO gradient there,
        # only the L1 term does. The product mirrors this (tests/test_gpu_loss.py); the torch restatement differentiates the true function.
        assert np.abs(host(f"{name}/ssim_valid_grad")).max() == 0 and np.abs(host(f"{name}/ssim_same_grad")).max() > 0
        l1 = (0.8 * torch.sign(img - gt) / img.numel()).numpy()
        close(l1, host(f"{name}/grad"), 1e-6, "L1 part only")


def test_bilateral_grid_component_restatement_matches_the_reference_host_code():
    from oracle import bilateral
    delta, rgb, v = host("bilateral/delta"), host("bilateral/rgb"), host("bilateral/v_out")
    eye = np.eye(4, dtype=np.float32)[:3].reshape(12)
    assert np.array_equal(host("bilateral/identity"), np.broadcast_to(eye[None, :, None, None, None], delta.shape))    # bilateral_grid.cpp:88-95
    grids = host("bilateral/identity") + delta
    x = np.clip(rgb, 0, 1).transpose(1, 2, 0)                                                                          # apply(): clamp, CHW -> HWC (:115-117)
    close(bilateral.slice_forward(grids[1], x, np.float64).transpose(2, 0, 1), host("bilateral/out"), 2e-6, "apply")
    gg, gr = bilateral.slice_backward(grids[1], x, v.transpose(1, 2, 0), np.float64)
    tvg = bilateral.tv_backward(grids, 10.0, np.float64)
    want = tvg.copy()
    want[1] += gg
    close(want, host("bilateral/g_grids"), 2e-5, "d / d grids (slice of image 1 + TV of all)")
    inside = ((rgb >= 0) & (rgb <= 1))
    close(gr.transpose(2, 0, 1) * inside, host("bilateral/g_rgb"), 2e-5, "d / d rgb through the clamp")
    assert abs(float(bilateral.tv_forward(grids, np.float64)) - float(host("bilateral/tv"))) <= 1e-5 * float(host("bilateral/tv"))


def test_product_warmup_schedule_is_the_reference_schedule_bit_for_bit():
    import lichtfeld_studio_amd  # noqa: F401
    from lichtfeld_studio_amd.fused_adam import WarmupExponentialLR

    class Opt:
        def __init__(self, lr):
            self.param_groups = [{"lr": lr}]
    lr0, gamma, warm, start, n_steps = host("warmup/args")
    opt = Opt(float(lr0))
    sched = WarmupExponentialLR(opt, float(gamma), int(warm), float(start), -1)
    got = []
    for _ in range(int(n_steps)):
        sched.step()
        got.append(opt.param_groups[0]["lr"])
    assert np.array_equal(np.array(got), host("warmup/lrs"))


def test_host_golden_entries_regenerate_from_the_reference_host_code():
    import oracle
    if oracle.ref_loss_host_lib() is None:
        pytest.skip("oracle/_ref/libref_loss_host.so not built (needs /root/reference)")
    for name in ("photometric_40x52", "photometric_small_9x40"):
        loss, g = oracle.ref_photometric_loss(host(f"{name}/rendered"), host(f"{name}/gt"), 0.2)
        assert loss == host(f"{name}/loss") and np.array_equal(g, host(f"{name}/grad"))
    r = oracle.ref_bilateral_grid(3, 6, 5, 4, 1, host("bilateral/delta"), host("bilateral/rgb"), host("bilateral/v_out"), 10.0)
    assert np.array_equal(r["out"], host("bilateral/out")) and np.array_equal(r["g_rgb"], host("bilateral/g_rgb"))
    close(r["g_grids"], host("bilateral/g_grids"), 1e-6, "g_grids")
    lr0, gamma, warm, start, n_steps = host("warmup/args")
    assert np.array_equal(oracle.ref_warmup_schedule(lr0, gamma, int(warm), start, int(n_steps)), host("warmup/lrs"))
