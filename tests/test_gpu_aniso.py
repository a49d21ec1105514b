"""GPU: the backward rasterizer (K8) on FLAT Gaussians - the probe of tests/test_emulated_aniso.py (tools/aniso_probe.py: one Gaussian, two scales 0.04, the third down to
5e-4 = aspect 80, six orientations, 48 x 48 pinhole image) on the MI355X through the shipped library: dL/dmeans, dL/dquats, dL/dscales against the fp64 oracle must stay
within 4 x the error of the oracle's own fp32 evaluation of the reference's cross-product form (Bwd.cu:263-333) or 3e-4, whichever is larger, at every aspect ratio; the forward
image within 2e-5. Rounds 1 - 4 shipped a backward that was off by up to 37 % in dL/dmeans at aspect 80; no GPU test had a Gaussian flatter than 6 : 1."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flat_gaussians_hold_the_oracles_fp32_accuracy_on_the_gpu(lfs, tmp_path):
    out = os.path.join(str(tmp_path), "aniso.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "aniso_probe.py"), "--gpu", "--json", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    print(r.stdout)
    rows = json.load(open(out))
    assert len(rows) == 6 and max(x["aspect"] for x in rows) >= 80
    for x in rows:
        assert x["fwd"] < 2e-5, x
        if x["aspect"] < 2:
            continue
        for k in ("v_quats", "v_means", "v_scales"):
            assert x[k + "_hip"] <= max(4 * x[k + "_o32"], 3e-4), (k, x)
        # round 6 (LFS_ACC_SYM): the scale and rotation gradients are sums of like-signed second moments - 8e-6 .. 3e-5 on the MI355X at every aspect ratio, float atomics
        # included (the dL/dA form moved by 1e-4 .. 4e-3 between runs at aspect 80: profiles/r06/lease23_aniso_probe_noise.txt)
        assert x["v_scales_hip"] < 1e-4 and x["v_quats_hip"] < 1e-4, x
