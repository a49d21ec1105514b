"""Static check of the shipped gfx950 code objects (no GPU): register allocation, scratch and LDS of the kernels of the training step, read from the AMDGPU
metadata inside lichtfeld-studio_amd/liblfs_gsplat.so (tools/kernel_resources.py). A source or compiler change that makes a hot kernel spill, or costs the
rasterizer its occupancy, fails HERE instead of showing up as a slower bench on the GPU box. The numbers DESIGN.md section 4 / 6 quote are the ones asserted."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "lichtfeld-studio_amd", "liblfs_gsplat.so")


@pytest.fixture(scope="module")
def table():
    if not os.path.exists(LIB) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("library not built / no llvm-readelf")
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    t = mod.kernels(LIB)
    assert len(t) > 100, len(t)
    return t


# the kernels one SYN-B training step launches (profiles/r04/step_launch_sequence.txt)
STEP = ["sh_fwd_kernel<16, true>", "projection_ut_kernel<true, true, true>", "isect_count_kernel<true>", "tile_scan_kernel", "isect_rows_kernel", "isect_tiles_kernel",
        "tile_sort_bins_kernel<256, 256, true, 32u>", "tile_sort_bins_kernel<1024, 1024, false, 64u>", "raster_cull_kernel<true>", "raster_fwd_kernel<3, 1>",
        "raster_bwd_kernel<3, 1, true, 0>", "sh_bwd_kernel<16, true, false, true>", "raster_finish_adam_kernel<true>"]


def test_only_the_kd_tree_kernel_uses_scratch(table):
    spilling = {k: v["scratch_bytes"] for k, v in table.items() if v["scratch_bytes"]}
    assert set(spilling) <= {"dataprep::kdtree_mean_distance_kernel"}, spilling   # (its traversal stack; initialisation only, not on the training path)


def test_step_kernels_exist_without_scratch_and_within_the_lds_a_plain_launch_may_use(table):
    for k in STEP:
        assert k in table, k
        assert table[k]["scratch_bytes"] == 0 and table[k]["agprs"] == 0, (k, table[k])
        assert table[k]["lds_bytes"] <= 64 * 1024, (k, table[k])


def test_rasterizer_pair_keeps_seven_wavefronts_per_simd(table):
    """7 wavefronts per SIMD, set by the ~100 SGPRs of the two record groups (not by the vector registers); the backward's LDS is the
    one value-major transpose block of lfs_raster_common.cuh (16 x 72 floats per wavefront)."""
    for k in ("raster_fwd_kernel<3, 1>", "raster_bwd_kernel<3, 1, true, 0>", "raster_bwd_kernel<3, 1, true, 2>"):
        assert table[k]["vgprs"] <= 48 and table[k]["waves_per_simd"] >= 7, (k, table[k])
    assert table["raster_bwd_kernel<3, 1, true, 0>"]["lds_bytes"] == 16 * 80 * 4   # (round 6: the quad layout's row stride, lfs_raster_common.cuh RED_QROW)
    assert table["raster_fwd_kernel<3, 1>"]["lds_bytes"] == 0


def test_streaming_kernels_register_budgets(table):
    """The optimizer-carrying SH backward and the Adam kernels run at full occupancy (they reach 5.1 - 5.8 TB/s); sh_fwd / projection / finish_adam sit at ~100
    VGPRs = 4 - 5 wavefronts per SIMD (3.4 - 4.2 TB/s: the lever DESIGN.md section 8 names for the next round) - a regression beyond that is flagged."""
    for k in ("sh_bwd_kernel<16, true, false, true>", "adam_kernel", "adam_multi_kernel", "raster_cull_kernel<true>"):
        assert table[k]["waves_per_simd"] == 8, (k, table[k])
    for k in ("sh_fwd_kernel<16, true>", "projection_ut_kernel<true, true, true>", "raster_finish_adam_kernel<true>"):
        assert table[k]["vgprs"] <= 104 and table[k]["waves_per_simd"] >= 4, (k, table[k])


def test_backward_rasterizer_writes_m0_exactly_once():
    """LFS_RED_M0_ONCE (lfs_raster_common.cuh): the add-TID stores of the LDS reduction take their base from M0, which raster_bwd_kernel's prologue sets once. The
    compiler is not told that the inline asm reads M0, so the statement "nothing else in the kernel touches it" is held here, against the disassembly of every
    shipped instantiation: one instruction mentions m0 (the prologue's s_mov_b32), and none of the forms that use it implicitly (LDS-direct loads, s_movrel /
    v_movrel, s_sendmsg with a payload, GDS) occurs."""
    if not os.path.exists(LIB) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("library not built / no llvm-objdump")
    spec = importlib.util.spec_from_file_location("isa", os.path.join(ROOT, "tools", "isa.py"))
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    isa = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(isa)
    kernels = {k: v for k, v in isa.disassemble(LIB).items() if k.startswith("raster_bwd_kernel<")}
    assert len(kernels) >= 16, sorted(kernels)
    for k, lines in kernels.items():
        ops = [l.split("//")[0].strip() for l in lines]
        m0 = [o for o in ops if "m0" in o.replace(",", " ").split()]
        addtid = [o for o in ops if o.startswith("ds_write_addtid_b32")]
        if not addtid:      # (a build with the asm stores switched off has nothing to hold)
            continue
        assert len(m0) == 1 and m0[0].startswith("s_mov_b32 m0,"), (k, m0)
        assert ops.index(m0[0]) < ops.index(addtid[0]), k
        implicit = [o for o in ops if o.split()[0].startswith(("s_movrel", "v_movrel", "s_sendmsg", "ds_gws", "ds_ordered")) or " lds" in o or " gds" in o]
        assert not implicit, (k, implicit[:4])
