"""Generates tests/golden/jpeg/: a handful of small JPEG files written with Pillow and the RGB arrays Pillow (libjpeg-turbo, default decode
pipeline: islow IDCT, fancy upsampling) decodes them to. The native decoder of liblfs_io must reproduce these arrays bit for bit
(tests/test_loader_io.py::test_native_jpeg_decoder_matches_committed_golden) - independent of whether Pillow is installed where the tests run.

    python tests/golden/make_jpeg_golden.py      # Pillow 12.2.0 / libjpeg-turbo (jpeglib 6.2 ABI) in the build container
"""
import os

import numpy as np
from PIL import Image, features

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jpeg")
os.makedirs(HERE, exist_ok=True)
rng = np.random.default_rng(20260922)


def picture(h, w):
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(x / 7.0 + y / 13.0), 128 + 90 * np.cos(x / 11.0 - y / 5.0), 128 + 80 * np.sin((x + y) / 9.0)], -1)
    return np.clip(img + rng.normal(0, 12, img.shape), 0, 255).astype(np.uint8)


CASES = {
    "baseline_444_q90": dict(size=(37, 53), quality=90, subsampling=0),
    "baseline_422_q75": dict(size=(40, 61), quality=75, subsampling=1),
    "baseline_420_q85_opt": dict(size=(67, 90), quality=85, subsampling=2, optimize=True),
    "progressive_420_q80": dict(size=(70, 99), quality=80, subsampling=2, progressive=True),
    "progressive_444_q95": dict(size=(33, 35), quality=95, subsampling=0, progressive=True),
    "grey_q85": dict(size=(40, 56), quality=85, grey=True),
    "restart_420_q80": dict(size=(48, 80), quality=80, subsampling=2, restart_marker_blocks=4),
    "tiny_2x2_420": dict(size=(2, 2), quality=90, subsampling=2),
}
expected = {}
for name, c in CASES.items():
    c = dict(c)
    h, w = c.pop("size")
    img = picture(h, w)
    if c.pop("grey", False):
        img = img[..., 0]
    path = os.path.join(HERE, name + ".jpg")
    Image.fromarray(img).save(path, **c)
    expected[name] = np.asarray(Image.open(path).convert("RGB"))
np.savez_compressed(os.path.join(HERE, "expected_rgb.npz"), **expected)
print("wrote", len(expected), "cases with Pillow", Image.__version__, "jpeg", features.version("jpg"), "turbo", features.check_feature("libjpeg_turbo"))
