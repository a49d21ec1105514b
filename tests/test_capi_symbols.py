"""CPU: the C-ABI library loads here (no GPU) and exports every symbol include/lfs_gsplat.h declares.
No compute entry point is called."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "lfs_gsplat.h")).read()
    return sorted(set(re.findall(r"LFS_API\s+[\w\s\*]+?\b(lfs_\w+)\s*\(", text)))


def test_header_declares_expected_surface():
    syms = _declared_symbols()
    for must in ["lfs_projection_ut_3dgs_fused", "lfs_spherical_harmonics_fwd", "lfs_spherical_harmonics_bwd",
                 "lfs_intersect_tile_count", "lfs_intersect_tile_emit", "lfs_intersect_offset",
                 "lfs_rasterize_to_pixels_from_world_3dgs_fwd", "lfs_rasterize_to_pixels_from_world_3dgs_bwd",
                 "lfs_quats_to_rotmats", "lfs_relocation", "lfs_add_noise", "lfs_adam_step"]:
        assert must in syms


def test_library_exports_every_declared_symbol(lfs):
    lib = lfs.load_library()
    raw = ctypes.CDLL(lfs.library_path())
    for s in _declared_symbols():
        assert hasattr(raw, s), f"liblfs_gsplat.so does not export {s}"
    assert lib.lfs_version().decode().startswith("lfs_gsplat gfx950")
    from lichtfeld_studio_amd import capi
    assert sorted(capi.EXPORTS) == _declared_symbols()


def test_workspace_size_queries_are_pure_host_functions(lfs):
    lib = lfs.load_library()
    a = lib.lfs_intersect_tile_workspace_bytes(ctypes.c_uint32(1), ctypes.c_uint32(1000000), ctypes.c_uint32(120), ctypes.c_uint32(68))
    b = lib.lfs_rasterize_workspace_bytes(ctypes.c_uint32(1), ctypes.c_uint32(1000000), ctypes.c_uint32(3), ctypes.c_uint32(1920), ctypes.c_uint32(1080), ctypes.c_uint32(16), ctypes.c_int64(5000000))
    assert a > 3 * 8160 * 4 and a % 256 == 0
    assert b >= 1000000 * 128 and b % 256 == 0


def test_ops_refuse_cpu_tensors_loudly(lfs):
    """No CPU fallback: CHECK_INPUT semantics (gsplat/Common.h:12-17)."""
    import pytest
    import torch
    from lichtfeld_studio_amd import ops
    with pytest.raises(RuntimeError):
        ops.quats_to_rotmats(torch.randn(4, 4))
    with pytest.raises(RuntimeError):
        ops.spherical_harmonics_fwd(0, torch.randn(4, 3), torch.randn(4, 1, 3), None)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under lichtfeld-studio_amd/ may import, include or load it."""
    pkg = os.path.join(ROOT, "lichtfeld-studio_amd")
    for dirpath, dirs, files in os.walk(pkg):
        dirs[:] = [d for d in dirs if d not in ("build", "__pycache__")]
        for f in files:
            if not f.endswith((".py", ".hip", ".cuh", ".h", ".cpp")):
                continue
            src = open(os.path.join(dirpath, f), errors="ignore").read()
            assert not re.search(r"^\s*(import|from)\s+oracle\b", src, re.M), f
            assert not re.search(r'#include\s+"[^"]*oracle', src), f
            assert "liboracle" not in src and "oracle_ops" not in src, f


def test_io_library_exports_every_symbol_of_its_header():
    """include/lfs_io.h <-> liblfs_io.so <-> loader.IO_EXPORTS (host-only library: loads without ROCm)."""
    text = open(os.path.join(ROOT, "include", "lfs_io.h")).read()
    declared = sorted(set(re.findall(r"LFS_IO_API\s+[\w\s\*]+?\b(lfs_\w+)\s*\(", text)))
    assert len(declared) >= 25 and "lfs_colmap_open" in declared and "lfs_ply_write_splat" in declared and "lfs_image_load_rgb8" in declared
    import lichtfeld_studio_amd  # noqa: F401
    from lichtfeld_studio_amd import loader
    raw = ctypes.CDLL(loader.io_library_path())
    for s in declared:
        assert hasattr(raw, s), f"liblfs_io.so does not export {s}"
    assert sorted(loader.IO_EXPORTS) == declared
    loader.io_library().lfs_io_version.restype = ctypes.c_char_p
    assert loader.io_library().lfs_io_version().decode().startswith("lfs_io")
    # nothing but libc / libstdc++ / zlib underneath: no HIP, no torch
    import subprocess
    needed = subprocess.run(["readelf", "-d", loader.io_library_path()], capture_output=True, text=True).stdout
    assert "amdhip" not in needed and "torch" not in needed and "libz" in needed


def test_step_driver_host_functions(lfs):
    """csrc/gut_step.hip, host side only (no launch): the workspace layout of the speculative training step grows with the capacity and keeps every region
    256-byte aligned and disjoint; lfs_gut_step_fits = "count within capacity AND longest tile list within the sort classes launched for the assumption";
    argument validation happens before anything is enqueued; lfs_gut_step_wait reads the stamp protocol from plain host memory."""
    from lichtfeld_studio_amd.gut_step import StepArgs, StepLayout
    lib = lfs.load_library()
    lay = {}
    for cap in (1 << 16, 5_000_000, 20_000_000):
        l = StepLayout()
        assert lib.lfs_gut_step_layout_for(ctypes.c_uint32(1_000_000), ctypes.c_uint32(1920), ctypes.c_uint32(1080), ctypes.c_uint32(16), ctypes.c_int64(cap), ctypes.byref(l)) == 0
        lay[cap] = l
        offs = sorted((getattr(l, k), k) for k, _ in StepLayout._fields_ if k not in ("bytes", "tile_offsets"))
        assert all(o % 256 == 0 for o, _ in offs) and len({o for o, _ in offs}) == len(offs), offs
        assert offs[-1][0] < l.bytes and l.tile_offsets < l.bytes
    assert lay[1 << 16].bytes < lay[5_000_000].bytes < lay[20_000_000].bytes
    # per intersection: 8 (keys) + 4 (ids) + 8 (binning scratch) + 4 cells x 8 (cell lists) = 52 bytes
    per = (lay[20_000_000].bytes - lay[5_000_000].bytes) / 15_000_000
    assert 51.9 < per < 52.1, per
    assert lib.lfs_gut_step_layout_for(ctypes.c_uint32(1000), ctypes.c_uint32(64), ctypes.c_uint32(64), ctypes.c_uint32(16), ctypes.c_int64(0), ctypes.byref(StepLayout())) != 0
    fits = lambda n, lg, cap, assumed: lib.lfs_gut_step_fits(ctypes.c_int64(n), ctypes.c_int64(lg), ctypes.c_int64(cap), ctypes.c_int64(assumed))
    assert fits(100, 10, 100, 1024) == 1 and fits(101, 10, 100, 1024) == 0
    assert fits(100, 1024, 1000, 1024) == 1 and fits(100, 1025, 1000, 1024) == 0      # first sort class: lists of <= 1024 entries
    assert fits(100, 4096, 1000, 1025) == 1 and fits(100, 4097, 1000, 4096) == 0
    assert fits(100, 16384, 1000, 5000) == 1 and fits(100, 16385, 1000, 16384) == 0
    assert fits(100, 10 ** 9, 1000, 16385) == 1                                          # the global-memory class sorts any length
    a = StepArgs()                                                                         # all-null arguments: refused before any launch
    assert lib.lfs_gut_train_step(ctypes.byref(a), ctypes.c_int64(1000), ctypes.c_int64(1024), None, ctypes.c_size_t(0), None, ctypes.c_int64(1), None) != 0
    counts = (ctypes.c_int64 * 3)(123, 45, 7)
    n, lg = ctypes.c_int64(0), ctypes.c_int64(0)
    assert lib.lfs_gut_step_wait(counts, ctypes.c_int64(7), ctypes.c_double(0.1), ctypes.byref(n), ctypes.byref(lg)) == 0 and (n.value, lg.value) == (123, 45)
    assert lib.lfs_gut_step_wait(counts, ctypes.c_int64(8), ctypes.c_double(0.05), ctypes.byref(n), ctypes.byref(lg)) != 0     # a stamp that never arrives: time-out, not a hang
