"""Test BODIES of the `-m gpu` parity suite executed on the CPU: the product library compiled as host code on the wavefront emulator (tests/emul_util.py), the
tests' "cuda:0" requests served by CPU tensors. Same inputs, same oracle / golden comparisons, same assertions and tolerances as on the MI355X - what differs is
the arithmetic underneath (host IEEE float, exp2f / division instead of v_exp_f32 / v_rcp_f32), so this checks the kernels' logic, not the ISA. The selection below
is every GPU test that (a) goes through the C ABI only (no HIP streams / events / libtorch extension) and (b) finishes in seconds under emulation; the
GPU run of the same functions stays the parity test proper."""
import importlib
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import emul_util  # noqa: E402

# module -> the tests of it that are NOT taken, each with the reason
SELECTION = {
    "test_gpu_small_ops": {},
    "test_gpu_projection_sh": {},
    "test_gpu_intersect": {"test_intersect_full_size_properties": "1 M Gaussians: minutes under emulation",
                           "test_two_pass_scatter_equals_one_pass_over_random_shapes": "40 random shapes up to 150 k Gaussians: half a minute; the two-pass scatter is covered by tests/test_emulated_intersect.py"},
    "test_gpu_loss": {"test_photometric_loss_full_size_properties": "1080p images: minutes under emulation",
                      "test_l1_ssim_trainers_agree_and_train": "trainer on a HIP stream"},
    "test_gpu_bilateral": {"test_gut_trainer_with_bilateral_grid_matches_autograd_composition": "trainer on a HIP stream"},
}
if os.environ.get("LFS_EMUL_SUITE_TRY"):   # development: LFS_EMUL_SUITE_TRY=test_gpu_loss,test_gpu_fastgs runs whole modules to see what the emulator can take
    SELECTION = {m: {} for m in os.environ["LFS_EMUL_SUITE_TRY"].split(",")}


@pytest.fixture(scope="module", autouse=True)
def _emulated_library_and_cpu_tensors():
    if not emul_util.available():
        pytest.skip("no clang++ to build the emulated library")
    with emul_util.installed(), emul_util.cuda_requests_served_by_the_cpu():
        yield


_MODULES = []
for _modname, _excluded in SELECTION.items():
    _mod = importlib.import_module(_modname)
    _MODULES.append(_modname)
    for _n in dir(_mod):
        _obj = getattr(_mod, _n)
        if _n.startswith("test_") and callable(_obj) and _n not in _excluded:
            globals()[f"test_emulated__{_modname[9:]}__{_n[5:]}"] = _obj
        elif hasattr(_obj, "_pytestfixturefunction") or type(_obj).__name__ == "FixtureFunctionDefinition":
            globals().setdefault(_n, _obj)   # the module's own fixtures
