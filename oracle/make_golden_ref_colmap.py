"""Generate tests/golden/colmap_scene/ (a small synthetic COLMAP dataset: binary + text sparse model, one camera per supported model, un-normalised
quaternions, CRLF text files) and tests/golden/ref_colmap.npz = what the REFERENCE'S OWN reader returns for it (oracle/_ref/libref_colmap.so:
src/loader/formats/colmap.cpp compiled in place against libtorch, `make -C oracle refcolmap`). Run in the build container, where /root/reference exists:
    python oracle/make_golden_ref_colmap.py
tests/test_loader_reference.py holds liblfs_io.so (the product's reader) and oracle/colmap_io.py (its restatement) to this file - SURVEY.md §8f row 4."""
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import test_loader_io as T  # noqa: E402  (the dataset writer the loader tests use)

FIELDS = ("camera_id", "colmap_model", "camera_model_type", "width", "height", "focal_x", "focal_y", "center_x", "center_y")


def pack(prefix, views, center, out):
    for k in FIELDS:
        out[f"{prefix}/{k}"] = np.array([v[k] for v in views])
    for k, width in (("R", 9), ("T", 3), ("radial", 6), ("tangential", 2), ("params", 12)):
        a = np.full((len(views), width), np.nan, np.float32)
        for i, v in enumerate(views):
            a[i, :v[k].size] = v[k].reshape(-1)
        out[f"{prefix}/{k}"] = a
    out[f"{prefix}/names"] = np.array([v["name"] for v in views])
    out[f"{prefix}/center"] = center


if __name__ == "__main__":
    assert oracle.ref_colmap_lib() is not None, "build oracle/_ref/libref_colmap.so first (make -C oracle refcolmap)"
    base = os.path.join(ROOT, "tests", "golden", "colmap_scene")
    shutil.rmtree(base, ignore_errors=True)
    cams, images, xyz, rgb = T._dataset(np.random.default_rng(2024), n_images=11)
    T._write(base, cams, images, xyz, rgb, "sparse/0", txt=False, images_folder="images_2")
    T._write(base, cams, images, xyz, rgb, "sparse/0", txt=True, images_folder="images_2")
    open(os.path.join(base, "images_2", ".keep"), "w").close()
    out = {}
    for folder in ("images_2",):
        for text in (False, True):
            views, center = oracle.ref_colmap_cameras(base, folder, text)
            pack(f"{folder}/{'txt' if text else 'bin'}", views, center, out)
    for text in (False, True):
        p, c = oracle.ref_colmap_points(base, text)
        out[f"points/{'txt' if text else 'bin'}/means"], out[f"points/{'txt' if text else 'bin'}/colors"] = p, c
    path = os.path.join(ROOT, "tests", "golden", "ref_colmap.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(base) for f in fs), "bytes of scene")
