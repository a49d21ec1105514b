"""Test BODIES of the `-m gpu` parity suite executed on the CPU: the product library compiled as host code on the wavefront emulator (tests/emul_util.py), the
tests' "cuda:0" requests served by CPU tensors. Same inputs, same oracle / golden comparisons, same assertions and tolerances as on the MI355X - what differs is
the arithmetic underneath (host IEEE float, exp2f / division instead of v_exp_f32 / v_rcp_f32), so this checks the kernels' logic, not the ISA. The selection below
is every GPU test that (a) goes through the C ABI only (no HIP streams / events / libtorch extension) and (b) finishes in seconds under emulation; the
GPU run of the same functions stays the parity test proper."""
import importlib
import os
import sys
import types

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import emul_util  # noqa: E402

# module -> the tests of it that are NOT taken, each with the reason (dict), or the list of the tests that are
SELECTION = {
    "test_gpu_small_ops": {},
    "test_gpu_zz_edge_sizes": {},
    "test_gpu_projection_sh": {},
    "test_gpu_intersect": {"test_intersect_full_size_properties": "1 M Gaussians: minutes under emulation",
                           "test_two_pass_scatter_equals_one_pass_over_random_shapes": "40 random shapes up to 150 k Gaussians: half a minute; the two-pass scatter is covered by tests/test_emulated_intersect.py"},
    "test_gpu_loss": {"test_photometric_loss_full_size_properties": "1080p images: minutes under emulation",
                      "test_l1_ssim_trainers_agree_and_train": "trainer on a HIP stream"},
    "test_gpu_bilateral": {"test_gut_trainer_with_bilateral_grid_matches_autograd_composition": "trainer on a HIP stream"},
    "test_gpu_fastgs": {"test_fast_rasterize_autograd_and_full_size": "1 M primitives at 1080p",
                        "test_fastgs_trainer_step_matches_autograd": "whole training steps: more than a minute each under emulation",
                        "test_fastgs_trainer_reference_default_configuration": "whole training steps", "test_fastgs_inline_shN_adam_matches_separate_optimizer": "whole training steps"},
    "test_gpu_dataprep": {"test_colmap_directory_to_training_and_ply": "trainer on a HIP stream", "test_train_colmap_tool_end_to_end": "a subprocess that needs the GPU",
                          "test_mean_neighbor_distances_large_against_kdtree": "a minute under emulation (the small sizes of the same kernel are taken)"},
    # every HIP-vs-reference-kernel golden comparison (tests/golden/refk_*.npz, ref_*.npz: outputs of the reference's own kernels / host code run on the CPU)
    "test_gpu_refk_golden": {"test_hip_rasterization_matches_reference_kernel[syn_a]": "10 000 Gaussians at 256 x 256: half a minute under emulation (nine other cases are taken)"},
    "test_gpu_strategy_reference": {},
    # the benchmarked path: the whole training step as ONE C++ call (csrc/gut_step.hip: lfs_gut_train_step, speculative workspace, pinned counts) against the
    # Python-enqueued op sequence, bit for bit; its split forms against render_and_backward
    "test_gpu_gut_step": {"test_cxx_split_step_is_bit_identical_for_the_other_losses_and_mcmc": "40 s per variant (trainer steps with SSIM / bilateral grid / MCMC around the C++ calls)",
                          "test_overflowing_attempt_updates_nothing_and_is_run_again": "20 - 130 s (the guarded attempt is covered by tests/test_emulated_step_pack.py)",
                          "test_fused_tail_step_is_bit_identical_to_the_three_pass_step[200000-3]": "twelve training steps on 200 000 Gaussians: minutes under emulation (three smaller cases are taken)",
                          "test_pipelined_step_is_bit_identical_to_the_one_stream_step[200000-3]": "as above",
                          "test_pipelined_step_is_bit_identical_to_the_one_stream_step[7000-2]": "a minute (the optional two-stream form: [65-3] and [3000-1] are taken; tests/test_emulated_step_pack.py holds it bit for bit on its own scenes)",
                          "test_pipelined_step_float_atomics_stays_within_accumulation_noise": "two minutes (the optional two-stream form, float atomics: a GPU property)"},
    # one whole training image through the Python mirror (projection -> SH -> intersection -> rasterize -> loss -> backward) against the oracle's pipeline
    "test_gpu_pipeline": ["test_train_step_gradients_match_oracle", "test_render_modes_and_background_gradient"],
    # the fused front half / backward / finish / inline-Adam entry points against the op-by-op path, torch autograd and the oracle
    "test_gpu_fused": {"test_batched_views_step_matches_the_view_by_view_step": "two minutes under emulation", "test_inline_shN_adam_trainer_path_trains": "100 s: dozens of training steps",
                       "test_all_inline_adam_step_is_bit_identical_to_the_separate_kernels": "a minute", "test_fused_finish_grads_matches_finish_plus_activation_backward": "half a minute",
                       "test_fused_and_autograd_trainers_take_the_same_steps": "25 s of training steps"},
    # the Python mirrors of rasterize() / fast_rasterize() and the fused training steps against what the reference's own rasterize() + backward() + train step produced
    "test_gpu_raster_reference": {},
    "test_gpu_raster": {"test_cell_culling_full_size_bit_identical": "1 M Gaussians at 1080p", "test_raster_bwd_is_linear_in_output_gradients_full_size": "1 M Gaussians at 1080p",
                        "test_deterministic_backward_mode_is_bit_reproducible": "the integer-atomic accumulation mode is compiled out of the emulated build (no 64-bit atomics to model)",
                        "test_lds_reduction_asm_block_agrees_with_the_compiler_generated_stores": "compares two GPU builds in a subprocess (ISA-level by definition)",
                        "test_instruction_properties_the_backward_relies_on": "runs a hipcc-built probe on the device (v_rcp_f32 at 1.0, v_mul_legacy_f32): hardware by definition",
                        "test_cell_culling_is_conservative[huge_and_near]": "half a minute (culling-off walks of every tile list); [small] and [low_opacity] are taken",
                        "test_cell_culling_is_conservative[needles]": "as above", "test_cell_culling_is_conservative[flat_disks]": "as above (tools/fuzz_emulated.py --flat checks culling on / off on flat disks)"},
    "test_gpu_strategies": {"test_default_strategy_fused_refinement_one_host_read_same_result": "counts host synchronisations of a HIP stream",
                            "test_mcmc_refinement_step_needs_no_host_sync": "counts host synchronisations of a HIP stream",
                            "test_mcmc_inline_shN_adam_between_refinements_is_bit_identical": "trainer on a HIP stream",
                            "test_mcmc_training_grows_the_model_and_fits_the_views": "trainer on a HIP stream"},
}
if os.environ.get("LFS_EMUL_SUITE_TRY"):   # development: LFS_EMUL_SUITE_TRY=test_gpu_loss,test_gpu_fastgs runs whole modules to see what the emulator can take
    SELECTION = {m: {} for m in os.environ["LFS_EMUL_SUITE_TRY"].split(",")}


@pytest.fixture(scope="module", autouse=True)
def _emulated_library_and_cpu_tensors():
    if not emul_util.available():
        pytest.skip("no clang++ to build the emulated library")
    with emul_util.installed(), emul_util.cuda_requests_served_by_the_cpu():
        yield


def _without_cases(fn, dropped):
    """a copy of the test function whose parametrize marks lack the values in `dropped` (exclusions written as name[case])"""
    new = types.FunctionType(fn.__code__, fn.__globals__, fn.__name__, fn.__defaults__, fn.__closure__)
    new.__dict__.update(fn.__dict__)
    marks = []
    for m in getattr(fn, "pytestmark", []):
        if m.name == "parametrize" and isinstance(m.args[0], str):   # (several arguments: the case id is their values joined with "-", as pytest writes it)
            case = (lambda v: "-".join(str(x) for x in v)) if "," in m.args[0] else str
            m = pytest.mark.parametrize(m.args[0], [v for v in m.args[1] if case(v) not in dropped], **m.kwargs).mark
        marks.append(m)
    new.pytestmark = marks
    return new


_MODULES = []
for _modname, _excluded in SELECTION.items():
    _mod = importlib.import_module(_modname)
    _MODULES.append(_modname)
    for _n in dir(_mod):
        _obj = getattr(_mod, _n)
        if _n.startswith("test_") and callable(_obj) and (_n in _excluded if isinstance(_excluded, list) else _n not in _excluded):
            _drop = {k[len(_n) + 1:-1] for k in _excluded if k.startswith(_n + "[")}
            globals()[f"test_emulated__{_modname[9:]}__{_n[5:]}"] = _without_cases(_obj, _drop) if _drop else _obj
        elif hasattr(_obj, "_pytestfixturefunction") or type(_obj).__name__ == "FixtureFunctionDefinition":
            globals().setdefault(_n, _obj)   # the module's own fixtures


def test_every_module_of_the_package_talks_to_the_emulated_library():
    """guards the runs above against being vacuous: a module that kept its reference to the real loader would call the gfx950 library (which fails without a
    device, or - worse - would not be what these tests claim to exercise)"""
    import lichtfeld_studio_amd as lfs
    pkg = lfs.__name__
    lib = emul_util.library()
    seen = 0
    for name, mod in list(sys.modules.items()):
        if mod is not None and (name == pkg or name.startswith(pkg + ".")) and hasattr(mod, "load_library"):
            assert mod.load_library() is lib, name
            seen += 1
    assert seen >= 8, seen
    assert lib.lfs_version().decode().endswith("src-unknown")      # the host build (csrc/version.hip without the build's hash), not liblfs_gsplat.so
    lib.lfs_set_debug_flags(16 | 1)
    try:
        assert int(lib.lfs_get_debug_flags()) == 1                  # the integer-atomic mode is not part of an emulated build: its bit never arrives
    finally:
        lib.lfs_set_debug_flags(0)
