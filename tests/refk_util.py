"""Loading the golden vectors generated from the reference's own kernels (tests/golden/refk_*.npz, oracle/make_golden_refk.py)."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RASTER_CASES = sorted(os.path.basename(f)[len("refk_raster_"):-4] for f in glob.glob(os.path.join(GOLDEN, "refk_raster_*.npz")))


def raster_case(name):
    d = dict(np.load(os.path.join(GOLDEN, f"refk_raster_{name}.npz")))
    for k in ("viewmats1", "backgrounds", "masks", "radial", "tangential", "thin_prism"):
        d.setdefault(k, None)
    d["W"], d["H"], d["tile"] = int(d["meta_W"]), int(d["meta_H"]), int(d["meta_tile"])
    d["camera_model"], d["rs_type"] = int(d["meta_camera_model"]), int(d["meta_rs_type"])
    return d


def projection_cases():
    z = np.load(os.path.join(GOLDEN, "refk_projection.npz"))
    cases = {}
    for key in z.files:
        name, field = key.split("/", 1)
        cases.setdefault(name, {})[field] = z[key]
    for d in cases.values():
        for k in ("viewmats1", "radial", "tangential", "thin_prism", "compensations"):
            d.setdefault(k, None)
        d["W"], d["H"] = int(d["meta_W"]), int(d["meta_H"])
        d["camera_model"], d["rs_type"] = int(d["meta_camera_model"]), int(d["meta_rs_type"])
        d["eps2d"], d["radius_clip"] = float(d["meta_eps2d"]), float(d["meta_radius_clip"])
        d["calc_compensations"], d["no_opacity"] = bool(d["meta_calc_compensations"]), bool(d["meta_no_opacity"])
        d["ut_params"] = d["meta_ut_params"]
    return cases


def small_ops():
    return dict(np.load(os.path.join(GOLDEN, "refk_small_ops.npz")))


def oracle_raster_args(d):
    return (d["means"], d["quats"], d["scales"], d["colors"], d["opacities_cn"], d["backgrounds"], d["masks"], d["W"], d["H"], d["tile"], d["viewmats0"],
            d["viewmats1"], d["Ks"], d["camera_model"], d["rs_type"], d["radial"], d["tangential"], d["thin_prism"], d["offsets"], d["flatten_ids"])
