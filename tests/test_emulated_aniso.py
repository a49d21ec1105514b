"""CPU (emulated product library): the backward rasterizer's gradients for FLAT Gaussians - one scale 4 .. 80 x below the other two, the shape most Gaussians of a trained scene
have. Found by the differential fuzzer (tools/fuzz_emulated.py --oracle, seed 12 case 1106; profiles/r04/fuzz_emulated.txt): the foot-vector form w = gro - t q of K8
(csrc/raster.hip ray_eval) subtracts two vectors of length |o - mu| / s_min, the component of w along q keeps an absolute rounding error of ulp(|gro|), and the finish pass multiplies
dL/dgro by 1 / s_min again - without a correction dL/dmeans is off by 0.3 % at aspect 10 and by tens of percent at aspect 80, dL/dquats by 2 - 4 % at aspect 40 - 80, where the
reference's cross-product form evaluated in fp32 (the oracle's float instantiation) stays at 1e-4. The forward image is not affected (|w|^2 is insensitive to the error).
LFS_BWD_REORTH (one Gram-Schmidt step on w, seven FMA-class instructions per evaluation, alpha untouched) removes it: the DEFAULT since round 5 (timed and re-verified on
the MI355X: DESIGN.md 6). Round 6: the global-shutter kernels evaluate in a per-Gaussian ROTATED frame in which g lies on one axis (LFS_REC_ROT, lfs_raster_common.cuh): the
foot vector is formed without a difference, the step has nothing left to remove there (the frame is built in double precision - in single precision its own rounding
puts the offset back: v_means 5e-3 at aspect 80); -DLFS_REC_ROT=0 is the round-5 form, which the rolling-shutter kernels keep. All three builds are run here: the default must hold the oracle's own fp32 accuracy at every aspect ratio; the -DLFS_BWD_REORTH=0 build documents
what the step is there for (so that nobody removes it as dead arithmetic). The same probe runs on the hardware in tests/test_gpu_aniso.py."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import emul_util  # noqa: E402


def _probe(tmp_path, defines):
    if not emul_util.available():
        pytest.skip("no clang++ to build the emulated library")
    out = os.path.join(tmp_path, "aniso.json")
    env = dict(os.environ, LFS_EMUL_DEFINES=defines)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "aniso_probe.py"), "--json", out], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    print(r.stdout)
    return json.load(open(out))


def _holds_the_oracles_fp32_accuracy(rows, min_aspect=0.0):
    for r in rows:
        if r["aspect"] < 2 or r["aspect"] < min_aspect:   # (aspect 1: dL/dquats is zero in exact arithmetic, nothing to be relative to)
            continue
        for k in ("v_quats", "v_means", "v_scales"):
            assert r[k + "_hip"] <= max(4 * r[k + "_o32"], 3e-4), (k, r)
        assert r["fwd"] < 2e-5, r


def test_flat_gaussians_default_build_holds_fp32_accuracy(tmp_path):
    rows = _probe(str(tmp_path), "")
    _holds_the_oracles_fp32_accuracy(rows)
    # round 6 (LFS_ACC_SYM): dL/dscales and dL/dquats come from sums of like-signed second moments, not from dL/dA . M + dL/dg . g (1e6 times their sum on the thin axis of a
    # flat Gaussian): an order of magnitude inside the reference's own fp32 arithmetic at every aspect ratio (measured 6e-6 .. 3e-5 against 5e-5 .. 6e-4)
    for r in rows:
        if r["aspect"] >= 2:
            assert r["v_scales_hip"] < 1e-4 and r["v_quats_hip"] < 1e-4, r


def test_round5_records_with_the_gram_schmidt_step_hold_fp32_accuracy(tmp_path):
    """-DLFS_REC_ROT=0: the round-5 form (g in the record, w = g - t q, one Gram-Schmidt step in the backward) - what the rolling-shutter kernels still run."""
    _holds_the_oracles_fp32_accuracy(_probe(str(tmp_path), "-DLFS_REC_ROT=0"))


def test_rotated_records_with_the_dA_accumulators_hold_fp32_accuracy(tmp_path):
    """-DLFS_ACC_SYM=0: the rotated records with the round-5 accumulator row (dL/dA, dL/dg) and the double-precision frame rebuilt by the finish pass - the A/B form
    between the two round-6 changes (794 -> 807 img/s for the symmetric sums on top of it)."""
    _holds_the_oracles_fp32_accuracy(_probe(str(tmp_path), "-DLFS_ACC_SYM=0"))


def test_without_the_reorthogonalisation_flat_gaussians_lose_their_position_gradient(tmp_path):
    rows = _probe(str(tmp_path), "-DLFS_REC_ROT=0 -DLFS_BWD_REORTH=0")
    for r in rows:
        if 2 <= r["aspect"] <= 4:
            assert r["v_quats_hip"] < 5e-4 and r["v_means_hip"] < 5e-4 and r["v_scales_hip"] < 5e-4, r
        assert r["fwd"] < 2e-5, r                                  # the forward holds at every aspect ratio
    # the defect itself, as numbers: aspect 40 - 80 is off by more than 1 % in dL/dmeans without the step
    assert max(r["v_means_hip"] for r in rows if r["aspect"] >= 40) > 1e-2
