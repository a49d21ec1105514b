"""The training step's front end with the rasterizer's records written by the PROJECTION kernel (round 4: csrc/projection_ut.hip PACK - the SH colours are evaluated
first, for every Gaussian, and the projection kernel stores the 64-byte record and the 32-byte culling record of each visible Gaussian itself) compiled as HOST
code on the wavefront emulator (tests/emul) and driven through lfs_gut_view_forward, against the SAME entry point with debug bit 6 set - the round-3 order:
projection, SH colours of the visible Gaussians, raster_pack_kernel's second pass over the Gaussians. pack_gaussian is contraction-free (lfs_raster_pack.cuh),
so both translation units produce the same record bits: render, alpha, radii, last_ids, tile offsets and the counts must be IDENTICAL.
(The masked tile lists this file was first written for - per-tile cell masks computed at the Gaussian, commit dfa4319 - were measured and removed:
profiles/r04/masked_tile_lists_negative.txt.)"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
CSRC = os.path.join(ROOT, "lichtfeld-studio_amd", "csrc")


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ to build the emulated kernels")
    out = str(tmp_path_factory.mktemp("emul") / "liblfs_step_emul.so")
    srcs = [os.path.join(CSRC, f) for f in ("intersect.hip", "raster.hip", "projection_ut.hip", "sh.hip", "gut_step.hip")]
    cmd = [CLANG, "-x", "c++", "-std=c++17", "-O1", "-DLFS_EMULATE", "-fPIC", "-shared", "-ffp-contract=on", "-I" + os.path.join(HERE, "emul"), "-Wno-unused-value",
           "-Wno-unknown-attributes", *srcs, os.path.join(HERE, "emul", "emul_stubs.cpp"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lib = C.CDLL(out)
    lib.lfs_rasterize_workspace_bytes.restype = C.c_size_t
    lib.lfs_intersect_tile_workspace_bytes.restype = C.c_size_t
    return lib


def _scene(seed, N, W, H, smin, smax, spread=1.0, K=4, degree=1):
    sys.path.insert(0, HERE)
    from gpu_util import make_gaussians, pinhole_K, small_rotation_viewmat
    rng = np.random.default_rng(seed)
    means, quats, scales, opac = make_gaussians(rng, N, spread=spread, smin=smin, smax=smax)
    raw_scales = np.log(scales).astype(np.float32)
    raw_opac = np.log(opac / (1 - opac)).astype(np.float32)
    opac_low = rng.random(N) < 0.1
    raw_opac[opac_low] = -6.0                                    # some below 1/255: never listed by the projection
    sh0 = (rng.standard_normal((N, 1, 3)) * 0.5).astype(np.float32)
    shN = (rng.standard_normal((N, K - 1, 3)) * 0.2).astype(np.float32)
    vm = small_rotation_viewmat(rng, 0.08, 0.15)
    Kmat = pinhole_K(0.8 * W, W, H, 1)[0]
    return dict(means=means, raw_quats=quats, raw_scales=raw_scales, raw_opac=raw_opac, sh0=sh0, shN=shN, vm=vm, K=Kmat, bg=rng.random(3).astype(np.float32), Kn=K, degree=degree)


def _forward(lib, sc, W, H, tile, capacity, flags, assumed_longest=1024):
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("lfs_gut_step_defs", os.path.join(ROOT, "lichtfeld-studio_amd", "gut_step.py"))
    src = open(spec.origin).read()
    ns = {}
    # only the two ctypes structures are needed (the module itself imports the GPU library loader)
    start, end = src.index("class StepArgs"), src.index("class GutStep")
    exec("import ctypes as C\n" + src[start:end], ns)
    StepArgs, StepLayout = ns["StepArgs"], ns["StepLayout"]
    N = sc["means"].shape[0]
    lib.lfs_set_debug_flags(C.c_uint32(flags))
    try:
        lay = StepLayout()
        assert lib.lfs_gut_step_layout_for(C.c_uint32(N), C.c_uint32(W), C.c_uint32(H), C.c_uint32(tile), C.c_int64(capacity), C.byref(lay)) == 0
        ws = np.full(int(lay.bytes) + 64, 0xA5, np.uint8)             # garbage: the step must not rely on a clean workspace
        keep = [np.ascontiguousarray(sc[k], np.float32) for k in ("means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opac", "vm", "K", "bg")]
        a = StepArgs()
        a.N, a.K, a.sh_degree, a.image_width, a.image_height, a.tile_size = N, sc["Kn"], sc["degree"], W, H, tile
        a.means, a.sh0, a.shN, a.raw_scales, a.raw_quats, a.raw_opacities, a.viewmat, a.Kmat, a.background = [x.ctypes.data for x in keep]
        counts = np.zeros(3, np.int64)
        rc = lib.lfs_gut_view_forward(C.byref(a), C.c_int64(capacity), C.c_int64(assumed_longest), C.c_void_p(ws.ctypes.data), C.c_size_t(int(lay.bytes)),
                                      C.c_void_p(counts.ctypes.data), C.c_int64(7), None)
        assert rc == 0, rc
        assert counts[2] == 7
        view = lambda name, dtype, shape: ws[getattr(lay, name):getattr(lay, name) + int(np.prod(shape)) * np.dtype(dtype).itemsize].view(dtype).reshape(shape).copy()
        tw, th = (W + tile - 1) // tile, (H + tile - 1) // tile
        out = dict(render=view("render", np.float32, (H, W, 3)), alpha=view("alpha", np.float32, (H, W)), radii=view("radii", np.int32, (N, 2)),
                   last_ids=view("last_ids", np.int32, (H, W)), offsets=view("tile_offsets", np.int32, (tw * th + 1,)), counts=counts.copy(),
                   abort=int(view("abort_flag", np.int32, (1,))[0]), ws=ws, lay=lay)
        return out
    finally:
        lib.lfs_set_debug_flags(C.c_uint32(0))


CASES = {
    # name: (N, W, H, tile, smin, smax, spread)
    "syn_a_like": (1500, 160, 112, 16, 0.01, 0.06, 1.0),
    "dense_long_lists": (5000, 48, 48, 16, 0.03, 0.12, 0.25),
    "large_gaussians": (300, 208, 144, 16, 0.2, 0.9, 1.0),
    "tile_8_ragged": (900, 91, 67, 8, 0.01, 0.08, 1.0),
}


@pytest.mark.parametrize("case", list(CASES))
def test_records_packed_by_the_projection_kernel_render_bit_identically(emu, case):
    N, W, H, tile, smin, smax, spread = CASES[case]
    sc = _scene(sum(map(ord, case)), N, W, H, smin, smax, spread)
    cap = 64 * N
    ref = _forward(emu, sc, W, H, tile, cap, flags=64, assumed_longest=1 << 20)     # round-3 order: separate pack kernel
    new = _forward(emu, sc, W, H, tile, cap, flags=0, assumed_longest=1 << 20)      # SH first, records written by the projection kernel
    assert ref["abort"] == 0 and new["abort"] == 0 and int(ref["counts"][0]) > 0
    assert np.array_equal(ref["counts"][:2], new["counts"][:2]) and np.array_equal(ref["offsets"], new["offsets"])
    assert np.array_equal(ref["radii"], new["radii"])
    assert (ref["alpha"] > 0.05).mean() > 0.02, "degenerate scene"
    assert np.array_equal(ref["render"], new["render"]) and np.array_equal(ref["alpha"], new["alpha"]) and np.array_equal(ref["last_ids"], new["last_ids"])
    # and with culling off (debug bit 0) on top: the cell lists are the tile lists, same image
    off = _forward(emu, sc, W, H, tile, cap, flags=1, assumed_longest=1 << 20)
    assert np.array_equal(off["render"], ref["render"]) and np.array_equal(off["alpha"], ref["alpha"]) and np.array_equal(off["last_ids"], ref["last_ids"])


def test_guarded_attempt_that_does_not_fit_reports_counts_and_renders_nothing(emu):
    N, W, H, tile, smin, smax, spread = CASES["syn_a_like"]
    sc = _scene(5, N, W, H, smin, smax, spread)
    ok = _forward(emu, sc, W, H, tile, 64 * N, flags=0)
    small = _forward(emu, sc, W, H, tile, max(int(ok["counts"][0]) // 3, 1), flags=0)
    assert small["abort"] == 1 and np.array_equal(small["counts"][:2], ok["counts"][:2])
    assert small["offsets"].max() == 0 and float(small["alpha"].max()) == 0.0


# ---- round 6: the pipelined step (lfs_gut_train_step_pipelined) against lfs_gut_train_step on the emulator -----------------------------------------------------------
# Streams are synchronous here, so this pins the DATA FLOW of the two-stream form - records packed without colours + the colour kernel writing their rgb slots, the SH
# backward split into a direction pass and an Adam pass fed by 32-byte hand-over rows, the abort-flag snapshot - not its ordering (tests/test_gpu_gut_step.py does that
# on the GPU, in the deterministic accumulation mode).
def _train(lib, sc, W, H, tile, capacity, steps, pipelined, assumed_longest=1 << 20, shrink_on_step=None, tail=False):
    import importlib.util
    spec = importlib.util.spec_from_file_location("lfs_gut_step_defs", os.path.join(ROOT, "lichtfeld-studio_amd", "gut_step.py"))
    src = open(spec.origin).read()
    ns = {}
    exec("import ctypes as C\n" + src[src.index("class StepArgs"):src.index("class GutStep")], ns)
    StepArgs, StepLayout = ns["StepArgs"], ns["StepLayout"]
    N = sc["means"].shape[0]
    lay = StepLayout()
    assert lib.lfs_gut_step_layout_for(C.c_uint32(N), C.c_uint32(W), C.c_uint32(H), C.c_uint32(tile), C.c_int64(capacity), C.byref(lay)) == 0
    ws = np.full(int(lay.bytes) + 64, 0xA5, np.uint8)
    names = ("means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opac")
    params = [np.ascontiguousarray(sc[k], np.float32).copy() for k in names]
    m = [np.zeros_like(p) for p in params]
    v = [np.zeros_like(p) for p in params]
    rng = np.random.default_rng(11)
    target = rng.random((3, H, W)).astype(np.float32)
    vm, Km, bg = [np.ascontiguousarray(sc[k], np.float32) for k in ("vm", "K", "bg")]
    from gpu_util import small_rotation_viewmat
    vms = [vm, np.ascontiguousarray(small_rotation_viewmat(np.random.default_rng(77), 0.2, 0.3), np.float32), vm]   # the views take turns: 0, 1, 2 (= 0 again), 0, ...
    loss = np.zeros(1, np.float32)
    losses, fitted = [], []
    fn = lib.lfs_gut_train_step_pipelined if pipelined else lib.lfs_gut_train_step
    colours_ready = False
    for it in range(steps):
        vm = vms[it % 3]
        a = StepArgs()
        a.N, a.K, a.sh_degree, a.image_width, a.image_height, a.tile_size = N, sc["Kn"], sc["degree"], W, H, tile
        a.means, a.sh0, a.shN, a.raw_scales, a.raw_quats, a.raw_opacities = [p.ctypes.data for p in params]
        for k in range(6):
            a.exp_avg[k], a.exp_avg_sq[k] = m[k].ctypes.data, v[k].ctypes.data
            t = it + 1
            for j, val in enumerate((1e-2 if k else 1e-3, 0.9, 0.999, 1e-15, 1.0 / (1.0 - 0.9 ** t), 1.0 / np.sqrt(1.0 - 0.999 ** t))):
                a.adam[k][j] = val
        a.viewmat, a.Kmat, a.background, a.target_chw = vm.ctypes.data, Km.ctypes.data, bg.ctypes.data, target.ctypes.data
        a.loss_weight, a.scale_reg, a.opacity_reg, a.loss = 1.0, 0.01, 0.01, loss.ctypes.data
        counts = np.zeros(3, np.int64)
        cap = capacity if shrink_on_step != it else 8          # an attempt that cannot fit: nothing may be updated by it
        if tail:   # lfs_gut_train_step_ex: the fused tail, with the next view named on every step but the last
            nxt = vms[(it + 1) % 3] if it + 1 < steps else None
            rc = lib.lfs_gut_train_step_ex(C.byref(a), C.c_void_p(nxt.ctypes.data) if nxt is not None else None, C.c_int(int(colours_ready)), C.c_int64(cap),
                                           C.c_int64(assumed_longest), C.c_void_p(ws.ctypes.data), C.c_size_t(int(lay.bytes)), C.c_void_p(counts.ctypes.data), C.c_int64(it + 1), None)
        else:
            rc = fn(C.byref(a), C.c_int64(cap), C.c_int64(assumed_longest), C.c_void_p(ws.ctypes.data), C.c_size_t(int(lay.bytes)), C.c_void_p(counts.ctypes.data),
                    C.c_int64(it + 1), None)
        assert rc == 0, rc
        fitted.append(bool(lib.lfs_gut_step_fits(C.c_int64(int(counts[0])), C.c_int64(int(counts[1])), C.c_int64(cap), C.c_int64(assumed_longest))))
        if tail:   # (an attempt that did not fit wrote no colours; what was ready before it still is - the NEXT call in this helper renders another view, though)
            colours_ready = fitted[-1] and it + 1 < steps
        losses.append(float(loss[0]))
    assert lib.lfs_gut_pipeline_join(None) in (0, 1)
    return dict(params=params, m=m, v=v, losses=losses, fitted=fitted)


@pytest.mark.parametrize("case", ["syn_a_like", "large_gaussians"])
def test_pipelined_step_matches_the_serial_step_bit_for_bit(emu, case):
    N, W, H, tile, smin, smax, spread = CASES[case]
    N = min(N, 600)
    sc = _scene(3 + sum(map(ord, case)), N, W, H, smin, smax, spread, K=16, degree=3)
    ref = _train(emu, sc, W, H, tile, 64 * N, 3, pipelined=False)
    new = _train(emu, sc, W, H, tile, 64 * N, 3, pipelined=True)
    assert all(ref["fitted"]) and all(new["fitted"]) and ref["losses"][0] > 0
    assert ref["losses"] == new["losses"]
    moved = 0
    for k in range(6):
        assert np.array_equal(ref["params"][k], new["params"][k]), k
        assert np.array_equal(ref["m"][k], new["m"][k]) and np.array_equal(ref["v"][k], new["v"][k]), k
        moved += int((ref["m"][k] != 0).sum())
    assert moved > 100, "no gradient reached the parameters"


def test_pipelined_attempt_that_does_not_fit_updates_nothing(emu):
    N, W, H, tile, smin, smax, spread = CASES["syn_a_like"]
    N = 500
    sc = _scene(21, N, W, H, smin, smax, spread, K=16, degree=3)
    ref = _train(emu, sc, W, H, tile, 64 * N, 2, pipelined=True)
    # the same two steps with a hopeless attempt in between (step index 1 of 3 runs with capacity 8): its kernels must leave parameters and moments alone
    new = _train(emu, sc, W, H, tile, 64 * N, 3, pipelined=True, shrink_on_step=1)
    assert new["fitted"] == [True, False, True]
    # (Adam's bias corrections follow the call index in _train, so only the FIRST step and the untouched state after the aborted attempt are compared)
    one = _train(emu, sc, W, H, tile, 64 * N, 1, pipelined=True)
    two = _train(emu, sc, W, H, tile, 64 * N, 2, pipelined=True, shrink_on_step=1)
    for k in range(6):
        assert np.array_equal(one["params"][k], two["params"][k]) and np.array_equal(one["m"][k], two["m"][k]) and np.array_equal(one["v"][k], two["v"][k]), k
    assert ref["fitted"] == [True, True]


# ---- round 6: the fused tail (lfs_gut_train_step_ex: SH backward + six Adam updates + the next view's SH colours in one launch) ----------------------------------------
@pytest.mark.parametrize("case,K,degree", [("syn_a_like", 16, 3), ("large_gaussians", 16, 2), ("tile_8_ragged", 4, 1)])
def test_fused_tail_step_matches_the_three_pass_step_bit_for_bit(emu, case, K, degree):
    """Three views in turn, 4 steps: every step after the first renders with the colours the previous step's tail evaluated for it (colors_ready), from the coefficient
    rows in registers and the mean it had just updated - parameters, moments and losses must be those of lfs_gut_train_step."""
    N, W, H, tile, smin, smax, spread = CASES[case]
    N = min(N, 600)
    sc = _scene(9 + sum(map(ord, case)), N, W, H, smin, smax, spread, K=K, degree=degree)
    ref = _train(emu, sc, W, H, tile, 64 * N, 4, pipelined=False)
    new = _train(emu, sc, W, H, tile, 64 * N, 4, pipelined=False, tail=True)
    assert all(ref["fitted"]) and all(new["fitted"]) and ref["losses"][0] > 0
    assert ref["losses"] == new["losses"]
    moved = 0
    for k in range(6):
        assert np.array_equal(ref["params"][k], new["params"][k]), k
        assert np.array_equal(ref["m"][k], new["m"][k]) and np.array_equal(ref["v"][k], new["v"][k]), k
        moved += int((ref["m"][k] != 0).sum())
    assert moved > 100, "no gradient reached the parameters"


def test_fused_tail_attempt_that_does_not_fit_updates_nothing(emu):
    N, W, H, tile, smin, smax, spread = CASES["syn_a_like"]
    N = 500
    sc = _scene(21, N, W, H, smin, smax, spread, K=16, degree=3)
    one = _train(emu, sc, W, H, tile, 64 * N, 1, pipelined=False, tail=True)
    two = _train(emu, sc, W, H, tile, 64 * N, 2, pipelined=False, tail=True, shrink_on_step=1)
    assert two["fitted"] == [True, False]
    for k in range(6):
        assert np.array_equal(one["params"][k], two["params"][k]) and np.array_equal(one["m"][k], two["m"][k]) and np.array_equal(one["v"][k], two["v"][k]), k
