"""CPU: liblfs_io.so (COLMAP bin + txt, splat PLY, PNG / PNM, image sizes) through lichtfeld_studio_amd.loader against the
restatement in oracle/colmap_io.py on synthetic datasets written in the documented COLMAP layouts; PNG decoding against Pillow.
Integers / strings / bytes bit-exact; float32 camera quantities to 2e-7 relative (same float32 expressions, libm sqrt)."""
import os
import struct

import numpy as np
import pytest

from oracle import colmap_io as oc


@pytest.fixture(scope="module")
def ld():
    import lichtfeld_studio_amd  # noqa: F401
    from lichtfeld_studio_amd import loader
    loader.io_library()
    return loader


def _dataset(rng, n_images=13):
    # one camera per supported model (ids deliberately unordered), images cycling through them
    params = {0: [800.5, 320.25, 240.75], 1: [810.0, 790.5, 321.0, 239.0], 2: [805.0, 320.0, 240.0, 0.0], 3: [600.0, 300.0, 200.0, 0.05, -0.01],
              4: [700.0, 710.0, 330.0, 250.0, 0.1, -0.05, 0.001, -0.002], 5: [400.0, 401.0, 320.0, 240.0, 0.01, 0.02, -0.03, 0.004],
              6: [650.0, 655.0, 322.0, 242.0, 0.1, 0.2, 0.003, 0.004, 0.5, 0.6, 0.7, 0.8], 8: [380.0, 320.0, 240.0, 0.07], 9: [390.0, 321.0, 241.0, 0.02, -0.004]}
    params[20] = [805.0, 320.0, 240.0, 0.125]   # SIMPLE_RADIAL with k1 != 0 (kept as radial distortion)
    cams = [(cid * 3 + 1, (2 if cid == 20 else cid), 640, 480, p) for cid, p in params.items()]
    ids = [c[0] for c in cams]
    images = []
    for i in range(n_images):
        q = rng.standard_normal(4) * (0.5 + i)         # un-normalised on purpose
        t = rng.standard_normal(3) * 3
        images.append((100 + i, list(q), list(t), ids[i % len(ids)], f"img_{i:03d}.png" if i % 2 else f"sub dir_{i}.JPG".replace(" ", "_")))
    xyz = rng.standard_normal((57, 3)) * 2
    rgb = rng.integers(0, 256, (57, 3))
    return cams, images, xyz, rgb


def _write(base, cams, images, xyz, rgb, layout="sparse/0", txt=False, images_folder="images", upper=False):
    sp = os.path.join(base, layout) if layout else base
    os.makedirs(sp, exist_ok=True)
    os.makedirs(os.path.join(base, images_folder), exist_ok=True)
    name = (lambda s: s.upper()) if upper else (lambda s: s)
    if txt:
        oc.write_cameras_txt(os.path.join(sp, name("cameras.txt")), cams, crlf=True)
        oc.write_images_txt(os.path.join(sp, name("images.txt")), images, crlf=True)
        oc.write_points3d_txt(os.path.join(sp, name("points3D.txt")), xyz, rgb)
    else:
        oc.write_cameras_bin(os.path.join(sp, name("cameras.bin")), cams)
        oc.write_images_bin(os.path.join(sp, name("images.bin")), images)
        oc.write_points3d_bin(os.path.join(sp, name("points3D.bin")), xyz, rgb)
    return sp


def _compare(views, center, ref_views, ref_center, base, folder):
    assert len(views) == len(ref_views)
    for v, r in zip(views, ref_views):
        assert (v.camera_id, v.colmap_model, v.camera_model_type, v.width, v.height, v.image_name) == \
               (r["camera_id"], r["colmap_model"], r["camera_model_type"], r["width"], r["height"], r["name"])
        assert v.image_path == os.path.join(base, folder, r["name"])
        for a, b in [(v.focal_x, r["focal_x"]), (v.focal_y, r["focal_y"]), (v.center_x, r["center_x"]), (v.center_y, r["center_y"])]:
            assert abs(a - b) <= 2e-7 * abs(b), (a, b)
        np.testing.assert_allclose(v.R, r["R"], rtol=0, atol=2e-7)
        assert np.array_equal(v.T, r["T"]) and np.array_equal(v.params, r["params"])
        assert np.array_equal(v.radial_distortion, r["radial"]) and np.array_equal(v.tangential_distortion, r["tangential"])
    np.testing.assert_allclose(center, ref_center, rtol=0, atol=2e-6)


@pytest.mark.parametrize("layout,upper", [("sparse/0", False), ("sparse", True), ("", False)])
def test_colmap_binary_and_text_match_oracle(ld, tmp_path, layout, upper):
    rng = np.random.default_rng(11)
    cams, images, xyz, rgb = _dataset(rng)
    base = str(tmp_path / "scene")
    sp = _write(base, cams, images, xyz, rgb, layout, txt=False, upper=upper)
    _write(base, cams, images, xyz, rgb, layout, txt=True, upper=upper)
    n = (lambda s: s.upper()) if upper else (lambda s: s)
    ref_cams = oc.read_cameras_bin(os.path.join(sp, n("cameras.bin")))
    ref_views, ref_center = oc.assemble(ref_cams, oc.read_images_bin(os.path.join(sp, n("images.bin"))))
    views, center = ld.read_colmap_cameras_and_images(base, "images")
    _compare(views, center, ref_views, ref_center, base, "images")
    assert views[0].R.dtype == np.float32 and abs(np.linalg.det(views[0].R.astype(np.float64)) - 1) < 1e-5
    # the model table: what ends up where
    by_model = {v.colmap_model: v for v in views}
    assert by_model[5].camera_model_type == 2 and len(by_model[5].radial_distortion) == 4 and len(by_model[5].tangential_distortion) == 0
    assert len(by_model[6].radial_distortion) == 6 and list(by_model[6].tangential_distortion) == [np.float32(0.003), np.float32(0.004)]
    assert by_model[0].focal_x == by_model[0].focal_y == np.float32(800.5)
    k1_zero = [v for v in views if v.colmap_model == 2 and v.params[3] == 0][0]
    k1_set = [v for v in views if v.colmap_model == 2 and v.params[3] != 0][0]
    assert len(k1_zero.radial_distortion) == 0 and list(k1_set.radial_distortion) == [np.float32(0.125)]
    # text files: float32 parsing of the decimal strings (std::stof), otherwise the same assembly
    tviews, tcenter = ld.read_colmap_cameras_and_images_text(base, "images")
    tref_views, tref_center = oc.assemble(oc.read_cameras_txt(os.path.join(sp, n("cameras.txt"))), oc.read_images_txt(os.path.join(sp, n("images.txt"))))
    _compare(tviews, tcenter, tref_views, tref_center, base, "images")
    # points
    pc = ld.read_colmap_point_cloud(base)
    assert np.array_equal(pc.means, xyz.astype(np.float32)) and np.array_equal(pc.colors, rgb.astype(np.uint8))
    pct = ld.read_colmap_point_cloud_text(base)
    assert np.array_equal(pct.means, xyz.astype(np.float32)) and np.array_equal(pct.colors, rgb.astype(np.uint8))


def test_colmap_scale_factor_and_image_size_correction(ld, tmp_path):
    rng = np.random.default_rng(5)
    cams, images, xyz, rgb = _dataset(rng, n_images=4)
    base = str(tmp_path / "scene")
    sp = _write(base, cams, images, xyz, rgb, images_folder="images_4")
    views, _ = ld.read_colmap_cameras_and_images(base, "images_4")
    ref_views, _ = oc.assemble(oc.read_cameras_bin(os.path.join(sp, "cameras.bin"), 4.0), oc.read_images_bin(os.path.join(sp, "images.bin")))
    for v, r in zip(views, ref_views):
        assert (v.width, v.height) == (160, 120) == (r["width"], r["height"])
        assert np.array_equal(v.params, r["params"]) and v.focal_x == r["focal_x"] and v.center_y == r["center_y"]
    assert views[0].focal_x == np.float32(views[0].params[0]) and abs(views[0].params[0] * 4 - cams[[c[0] for c in cams].index(views[0].camera_id)][4][0]) < 1e-3
    assert [oc.folder_scale(s) for s in ("images", "images_2", "images_x", "images_32", "a_b_8")] == [1.0, 2.0, 1.0, 1.0, 8.0]
    # the first image exists with a different size: every view is rescaled to it (colmap.cpp:836-865)
    ld.write_png(os.path.join(base, "images_4", images[0][4]), np.zeros((90, 200, 3), np.uint8))
    views2, _ = ld.read_colmap_cameras_and_images(base, "images_4")
    ref2, _ = oc.assemble(oc.read_cameras_bin(os.path.join(sp, "cameras.bin"), 4.0), oc.read_images_bin(os.path.join(sp, "images.bin")), first_image_size=(200, 90))
    for v, r in zip(views2, ref2):
        assert (v.width, v.height) == (200, 90)
        assert v.focal_x == r["focal_x"] and v.focal_y == r["focal_y"] and v.center_x == r["center_x"] and v.center_y == r["center_y"]


def test_colmap_error_behaviour(ld, tmp_path):
    rng = np.random.default_rng(7)
    cams, images, xyz, rgb = _dataset(rng, n_images=3)
    base = str(tmp_path / "scene")
    sp = _write(base, cams, images, xyz, rgb)
    with pytest.raises(ld.LoaderError, match="Images folder does not exist"):
        ld.read_colmap_cameras_and_images(base, "images_8")
    with pytest.raises(ld.LoaderError, match="Cannot find 'cameras.txt'"):
        ld.read_colmap_cameras_and_images_text(base, "images")
    for fname, what in (("cameras.bin", "cameras.bin"), ("images.bin", "images.bin"), ("points3D.bin", "points3D.bin")):
        path = os.path.join(sp, fname)
        good = open(path, "rb").read()
        open(path, "wb").write(good + b"\0")
        with pytest.raises(ld.LoaderError, match=f"{what}: trailing bytes"):
            ld.read_colmap_point_cloud(base) if fname.startswith("points") else ld.read_colmap_cameras_and_images(base, "images")
        open(path, "wb").write(good[:-9])           # truncated: an error, not a read past the buffer
        with pytest.raises(ld.LoaderError, match="unexpected end of file|trailing bytes|unterminated"):
            ld.read_colmap_point_cloud(base) if fname.startswith("points") else ld.read_colmap_cameras_and_images(base, "images")
        open(path, "wb").write(good)
    # rejected models and ids
    for model, n_params, msg in ((7, 5, "FOV camera model is not supported"), (10, 12, "THIN_PRISM_FISHEYE camera model is not supported")):
        oc.write_cameras_bin(os.path.join(sp, "cameras.bin"), [(cams[0][0], model, 640, 480, [1.0] * n_params)])
        oc.write_images_bin(os.path.join(sp, "images.bin"), [(1, [1, 0, 0, 0], [0, 0, 0], cams[0][0], "a.png")])
        with pytest.raises(ld.LoaderError, match=msg):
            ld.read_colmap_cameras_and_images(base, "images")
    with open(os.path.join(sp, "cameras.bin"), "wb") as f:
        f.write(struct.pack("<QIiQQ", 1, 1, 11, 640, 480))
    with pytest.raises(ld.LoaderError, match="Unsupported camera-model id 11"):
        ld.read_colmap_cameras_and_images(base, "images")
    oc.write_cameras_bin(os.path.join(sp, "cameras.bin"), [(5, 1, 640, 480, [1.0, 1.0, 2.0, 2.0])])
    oc.write_images_bin(os.path.join(sp, "images.bin"), [(1, [1, 0, 0, 0], [0, 0, 0], 6, "a.png")])
    with pytest.raises(ld.LoaderError, match="Camera ID 6 not found"):
        ld.read_colmap_cameras_and_images(base, "images")
    open(os.path.join(sp, "images.txt"), "w").write("# c\n1 1 0 0 0 0 0 0 5 a.png\n\n2 1 0 0 0 0 0 0 5\n1 2 3\n")
    open(os.path.join(sp, "cameras.txt"), "w").write("5 PINHOLE 640 480 1 1 2 2\n")
    with pytest.raises(ld.LoaderError, match="Invalid format in images.txt line 3"):
        ld.read_colmap_cameras_and_images_text(base, "images")
    open(os.path.join(sp, "cameras.txt"), "w").write("5 PINHOLEX 640 480 1 1 2 2\n")
    with pytest.raises(ld.LoaderError, match="Invalid format in cameras.txt"):
        ld.read_colmap_cameras_and_images_text(base, "images")


def test_ply_write_matches_reference_layout_and_reads_back(ld, tmp_path):
    import torch
    from lichtfeld_studio_amd.rasterizer import SplatModel
    rng = np.random.default_rng(3)
    N, K = 101, 9
    arr = dict(means=rng.standard_normal((N, 3)), sh0=rng.standard_normal((N, 1, 3)), shN=rng.standard_normal((N, K - 1, 3)),
               scales=rng.standard_normal((N, 3)), quats=rng.standard_normal((N, 4)) * 3, opac=rng.standard_normal(N))
    t = {k: torch.tensor(v, dtype=torch.float32) for k, v in arr.items()}
    model = SplatModel(t["means"], t["sh0"], t["shN"], t["scales"], t["quats"], t["opac"], 2)
    path = str(tmp_path / "out" / "splat_7000.ply")
    ld.save_ply(model, path)
    f32 = {k: v.numpy() for k, v in t.items()}
    ref = oc.ply_bytes(f32["means"], f32["sh0"], f32["shN"], f32["opac"], f32["scales"], f32["quats"])
    got = open(path, "rb").read()
    hdr_len = got.index(b"end_header\n") + 11
    assert got[:hdr_len] == ref[:hdr_len]
    a, b = np.frombuffer(got[hdr_len:], "<f4").reshape(N, -1), np.frombuffer(ref[hdr_len:], "<f4").reshape(N, -1)
    assert a.shape[1] == 6 + 3 + 3 * (K - 1) + 1 + 3 + 4
    assert np.array_equal(a[:, :-4], b[:, :-4]) and np.allclose(a[:, -4:], b[:, -4:], atol=1e-7)   # rotation normalised by torch vs numpy
    names, data = ld.read_ply(path)
    assert names == oc.ply_attribute_names(3, 3 * (K - 1)) and np.array_equal(data, a)
    back = ld.load_ply(path, device="cpu")
    for name, x, y in zip(["means", "sh0", "shN", "scales", "quats", "opac"], back.parameters(), [t["means"], t["sh0"], t["shN"], t["scales"],
                                                                                                 torch.nn.functional.normalize(t["quats"], dim=-1), t["opac"]]):
        assert x.shape == y.shape and torch.allclose(x.detach(), y, atol=1e-7), name
    assert back.get_active_sh_degree() == 2 and back.max_sh_degree == 2          # a LOADED model evaluates every degree it holds (evaluation / rendering of a trained file)
    assert ld.load_ply(path, device="cpu", active_sh_degree=0).get_active_sh_degree() == 0     # ... a resume path asks for the reference's SplatData start (splat_data.cpp:211)
    # generic reader: ascii, mixed types, an element after the vertices, doubles
    p2 = str(tmp_path / "mixed.ply")
    open(p2, "w").write("ply\nformat ascii 1.0\ncomment hi\nelement vertex 2\nproperty double x\nproperty float y\nproperty uchar red\nelement face 1\n"
                        "property list uchar int vertex_indices\nend_header\n0.5 1.5 255\n-2 3e-1 7\n3 0 1 1\n")
    names, data = ld.read_ply(p2)
    assert names == ["x", "y", "red"] and np.array_equal(data, np.array([[0.5, 1.5, 255], [-2, 0.3, 7]], np.float32))
    p3 = str(tmp_path / "bin.ply")
    with open(p3, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 2\nproperty double x\nproperty short s\nproperty uchar c\nend_header\n")
        f.write(struct.pack("<dhBdhB", 1.25, -3, 200, -7.5, 12, 1))
    names, data = ld.read_ply(p3)
    assert np.array_equal(data, np.array([[1.25, -3, 200], [-7.5, 12, 1]], np.float32))
    for bad, msg in ((b"plx\n", "File too small|missing PLY header"), (b"ply\nformat binary_big_endian 1.0\nelement vertex 0\nend_header\n", "not supported"),
                     (b"ply\nformat binary_little_endian 1.0\nelement vertex 5\nproperty float x\nend_header\nabcd", "truncated"),
                     (b"ply\nformat binary_little_endian 1.0\nelement vertex 5\nproperty float x\n", "No end_header")):
        open(p3, "wb").write(bad + b" " * 8)
        with pytest.raises(ld.LoaderError, match=msg):
            ld.read_ply(p3)


def test_image_headers_decoders_and_sizes(ld, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    cases = {"rgb.png": Image.fromarray(rgb), "rgba.png": Image.fromarray(np.dstack([rgb, rgb[..., :1]]), "RGBA"), "grey.png": Image.fromarray(rgb[..., 0], "L"),
             "la.png": Image.fromarray(rgb[..., :2], "LA"), "pal.png": Image.fromarray(rgb).quantize(17), "rgb.jpg": Image.fromarray(rgb),
             "grey16.png": Image.fromarray((rgb[..., 0].astype(np.uint16) * 257))}
    for name, im in cases.items():
        p = str(tmp_path / name)
        im.save(p, **({"quality": 95} if name.endswith("jpg") else {"optimize": name == "rgb.png"}))
        w, h, c = ld.get_image_info(p)
        assert (w, h) == (53, 37), name
        assert c == {"rgb.png": 3, "rgba.png": 4, "grey.png": 1, "la.png": 2, "pal.png": 3, "rgb.jpg": 3, "grey16.png": 1}[name], name
        got = ld.decode_rgb8(p)
        assert got.shape == (37, 53, 3) and got.dtype == np.uint8
        if name in ("rgb.png", "rgba.png"):
            assert np.array_equal(got, rgb), name
        elif name in ("grey.png", "grey16.png"):
            assert np.array_equal(got, np.repeat(rgb[..., :1], 3, -1)), name
        elif name == "la.png":      # 2 channels -> (r, g, (r + g) / 2), image_io.cpp:235-247
            assert np.array_equal(got[..., :2], rgb[..., :2]) and np.array_equal(got[..., 2], ((rgb[..., 0].astype(int) + rgb[..., 1]) // 2).astype(np.uint8))
        elif name == "pal.png":
            assert np.array_equal(got, np.asarray(Image.open(p).convert("RGB")))
        else:                        # JPEG: decoded natively, bit-identical to libjpeg-turbo (Pillow)
            assert np.array_equal(got, np.asarray(Image.open(p).convert("RGB")))
    # PNM, and our own PNG writer through Pillow
    pp = str(tmp_path / "a.ppm")
    open(pp, "wb").write(b"P6\n# c\n53 37\n255\n" + rgb.tobytes())
    assert ld.get_image_info(pp) == (53, 37, 3) and np.array_equal(ld.decode_rgb8(pp), rgb)
    pg = str(tmp_path / "a.pgm")
    open(pg, "wb").write(b"P5 53 37 255\n" + rgb[..., 1].tobytes())
    assert ld.get_image_info(pg) == (53, 37, 1) and np.array_equal(ld.decode_rgb8(pg)[..., 2], rgb[..., 1])
    pw = str(tmp_path / "w.png")
    ld.write_png(pw, rgb)
    assert np.array_equal(np.asarray(Image.open(pw)), rgb) and np.array_equal(ld.decode_rgb8(pw), rgb)
    with pytest.raises(ld.LoaderError, match="Failed to open"):
        ld.get_image_info(str(tmp_path / "missing.png"))
    head = open(pw, "rb").read()[:60]
    open(pw, "wb").write(head)
    with pytest.raises(ld.LoaderError, match="PNG"):
        ld.decode_rgb8(pw)
    # load_image's output sizes (image_io.cpp:112-270)
    for w, h, div, mw in [(1920, 1080, -1, 0), (1920, 1080, 1, 0), (1920, 1080, 2, 0), (4946, 3286, 4, 0), (4946, 3286, 8, 500), (1000, 3000, -1, 1600),
                          (3000, 1000, 2, 1000), (5, 3, 8, 0), (1600, 1600, -1, 1600), (1601, 1600, -1, 1600)]:
        assert ld.image_target_size(w, h, div, mw) == oc.target_size(w, h, div, mw), (w, h, div, mw)
    with pytest.raises(ld.LoaderError, match="unsupported resize factor 3"):
        ld.image_target_size(100, 100, 3, 0)


def test_dataset_split_and_camera_matrices(ld):
    cams = [ld.CameraData(i, 1, 0, 640, 480, 500.0, 510.0, 320.0, 240.0, np.eye(3, dtype=np.float32), np.array([1, 2, 3], np.float32), np.zeros(0, np.float32),
                          np.zeros(0, np.float32), np.zeros(4, np.float32), f"{i}.png", f"/nonexistent/{i}.png") for i in range(20)]
    tr, va, al = (ld.CameraDataset(cams, s, test_every=8) for s in ("train", "val", "all"))
    assert va.indices == [0, 8, 16] and len(tr) == 17 and 8 not in tr.indices and len(al) == 20     # dataset.hpp:42
    assert tr.image_size(0) == (640, 480) and ld.CameraDataset(cams, "all", 8, resize_factor=2).image_size(0) == (320, 240)
    m = ld.world_to_view(cams[0])
    assert np.array_equal(m[:3, 3], [1, 2, 3]) and np.array_equal(m[3], [0, 0, 0, 1])               # [R | t], camera.cpp:15-23
    K = ld.intrinsics(cams[0], 320, 240)
    assert np.array_equal(K, np.array([[250, 0, 160], [0, 255, 120], [0, 0, 1]], np.float32))          # camera.cpp:77-98


def _native_rgb8(ld, path):
    """liblfs_io only (no Pillow fallback)"""
    import ctypes as C
    lib = ld.io_library()
    data, w, h = C.POINTER(C.c_uint8)(), C.c_int32(), C.c_int32()
    rc = lib.lfs_image_load_rgb8(os.fsencode(path), C.byref(data), C.byref(w), C.byref(h))
    if rc != 0:
        return rc, lib.lfs_io_last_error().decode()
    try:
        return 0, np.ctypeslib.as_array(data, shape=(h.value, w.value, 3)).copy()
    finally:
        lib.lfs_io_free(data)


def test_native_jpeg_decoder_is_bit_identical_to_libjpeg_turbo(ld, tmp_path):
    """csrc_host/lfs_jpeg.cpp (islow IDCT, fancy upsampling, fixed-point YCbCr->RGB: the libjpeg defaults OpenImageIO decodes with) against Pillow's
    libjpeg-turbo, byte for byte: baseline / progressive, 4:4:4 / 4:2:2 / 4:2:0, grey, ragged and tiny sizes, optimised tables, restart markers, noise."""
    from PIL import Image
    rng = np.random.default_rng(4)

    def picture(h, w):
        y, x = np.mgrid[0:h, 0:w]
        img = np.stack([128 + 100 * np.sin(x / 7.0 + y / 13.0), 128 + 90 * np.cos(x / 11.0 - y / 5.0), 128 + 80 * np.sin((x + y) / 9.0)], -1)
        return np.clip(img + rng.normal(0, 12, img.shape), 0, 255).astype(np.uint8)

    p = str(tmp_path / "t.jpg")
    n = 0
    for (h, w) in [(64, 64), (37, 53), (17, 9), (1, 1), (2, 2), (3, 1), (200, 303), (33, 2), (5, 3), (1, 40), (40, 1)]:
        for sub in (0, 1, 2):
            for prog in (False, True):
                for q, extra in ((30, {}), (90, {"optimize": True}), (100, {})):
                    try:
                        Image.fromarray(picture(h, w)).save(p, quality=q, subsampling=sub, progressive=prog, **extra)
                    except OSError:
                        continue        # Pillow's encoder buffer on some tiny optimised files
                    rc, got = _native_rgb8(ld, p)
                    assert rc == 0, got
                    assert np.array_equal(got, np.asarray(Image.open(p).convert("RGB"))), (h, w, sub, prog, q)
                    n += 1
    assert n > 150
    for prog in (False, True):      # grey, and white noise at high quality (long EOB runs, every Huffman code length)
        Image.fromarray(picture(40, 56)[..., 0]).save(p, quality=85, progressive=prog)
        rc, got = _native_rgb8(ld, p)
        assert rc == 0 and np.array_equal(got, np.asarray(Image.open(p).convert("RGB")))
        assert ld.get_image_info(p) == (56, 40, 1)
        Image.fromarray(rng.integers(0, 256, (120, 200, 3), dtype=np.uint8)).save(p, quality=97, subsampling=2, progressive=prog)
        rc, got = _native_rgb8(ld, p)
        assert rc == 0 and np.array_equal(got, np.asarray(Image.open(p).convert("RGB")))
    try:                             # restart intervals (Pillow >= 10.2)
        for kw in ({"restart_marker_blocks": 3}, {"restart_marker_rows": 1}):
            for prog in (False, True):
                Image.fromarray(picture(70, 90)).save(p, quality=80, subsampling=2, progressive=prog, **kw)
                assert b"\xff\xdd" in open(p, "rb").read()
                rc, got = _native_rgb8(ld, p)
                assert rc == 0 and np.array_equal(got, np.asarray(Image.open(p).convert("RGB"))), (kw, prog)
    except TypeError:
        pass
    # error behaviour: truncated data is a format error (or decodes what is there), CMYK is reported as unsupported so that the host layer can fall back
    good = open(p, "rb").read()
    open(p, "wb").write(good[:20])
    rc, msg = _native_rgb8(ld, p)
    assert rc == -3 and "JPEG" in msg
    Image.fromarray(picture(16, 16)).convert("CMYK").save(p)
    rc, msg = _native_rgb8(ld, p)
    assert rc == -4 and "component" in msg
    assert ld.decode_rgb8(p).shape == (16, 16, 3)       # Pillow fallback


def test_blender_transforms_loader_matches_reference_operations(ld, tmp_path):
    """lfs_transforms_open (transforms.cpp:73-265) against the restatement that uses the reference's own torch float32 ops (inverse, mm): synthetic
    NeRF-style set (camera_angle_x, extension-less file paths, comments), an instant-ngp style set (fl_x / fl_y / cx / cy / w / h), error behaviour."""
    import json
    import math
    rng = np.random.default_rng(2)
    base = tmp_path / "lego"
    (base / "train").mkdir(parents=True)
    frames = []
    for i in range(7):
        a, b = rng.uniform(0, 2 * math.pi), rng.uniform(-1, 1)
        Rz = np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]])
        Rx = np.array([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
        m = np.eye(4)
        m[:3, :3] = Rz @ Rx
        m[:3, 3] = rng.uniform(-4, 4, 3)
        frames.append({"file_path": f"./train/r_{i}", "rotation": 0.1, "transform_matrix": m.tolist()})
        ld.write_png(str(base / "train" / f"r_{i}.png"), np.zeros((40, 40, 3), np.uint8))
    text = "{\n  // NeRF synthetic\n  \"camera_angle_x\": 0.6911112070083618, /* fov */\n  \"frames\": " + json.dumps(frames) + "\n}\n"
    (base / "transforms_train.json").write_text(text)
    views, center = ld.read_transforms_cameras_and_images(str(base))
    ref = oc.read_transforms(str(base / "transforms_train.json"), first_image_size=(40, 40))
    assert len(views) == len(ref) == 7 and np.array_equal(center, np.zeros(3, np.float32))
    for v, r in zip(views, ref):
        assert (v.camera_id, v.width, v.height, v.camera_model_type) == (r["camera_id"], 40, 40, 0)
        assert v.image_name == os.path.basename(r["file_path"]) + ".png" and os.path.exists(v.image_path)
        assert v.focal_x == r["focal_x"] and v.focal_y == r["focal_y"] and v.center_x == r["center_x"] == 20.0
        np.testing.assert_allclose(v.R, r["R"], rtol=0, atol=3e-7)
        np.testing.assert_allclose(v.T, r["T"], rtol=0, atol=2e-6)
        assert abs(np.linalg.det(v.R.astype(np.float64)) - 1) < 1e-5
    # instant-ngp style: explicit intrinsics, file path of the json itself, images with extension that do not exist
    ngp = {"w": 800, "h": 600, "fl_x": 700.5, "fl_y": 701.25, "cx": 399.5, "cy": 301.0, "k1": 0.0, "aabb_scale": 4,
           "frames": [{"file_path": "images/a.jpg", "transform_matrix": frames[0]["transform_matrix"]}]}
    p = tmp_path / "ngp.json"
    p.write_text(json.dumps(ngp))
    views, _ = ld.read_transforms_cameras_and_images(str(p))
    assert (views[0].width, views[0].height, views[0].focal_x, views[0].focal_y, views[0].center_x, views[0].center_y) == (800, 600, 700.5, 701.25, 399.5, 301.0)
    assert views[0].image_path == str(tmp_path / "images" / "a.jpg")
    # errors
    with pytest.raises(ld.LoaderError, match="could not find transforms_train.json nor transforms.json"):
        ld.read_transforms_cameras_and_images(str(tmp_path))
    ngp["k1"] = 0.1
    p.write_text(json.dumps(ngp))
    with pytest.raises(ld.LoaderError, match="GS don't support distortion"):
        ld.read_transforms_cameras_and_images(str(p))
    ngp["k1"] = 0.0
    del ngp["fl_y"]
    p.write_text(json.dumps(ngp))
    with pytest.raises(ld.LoaderError, match="no camera_angle_y expected w!=h"):
        ld.read_transforms_cameras_and_images(str(p))
    ngp["frames"][0]["transform_matrix"] = [[1, 0, 0, 0]] * 3
    ngp["fl_y"] = 1.0
    p.write_text(json.dumps(ngp))
    with pytest.raises(ld.LoaderError, match="transform_matrix has the wrong dimensions"):
        ld.read_transforms_cameras_and_images(str(p))
    p.write_text("{\"frames\": [")
    with pytest.raises(ld.LoaderError, match="JSON parse error"):
        ld.read_transforms_cameras_and_images(str(p))
    pc = ld.generate_random_point_cloud()
    assert pc.means.shape == (10000, 3) and pc.colors.dtype == np.uint8 and -1 <= pc.means.min() and pc.means.max() <= 1
    assert np.array_equal(pc.means, ld.generate_random_point_cloud().means)       # seeded: every call (and the reference) sees the same cloud


def test_native_jpeg_decoder_matches_committed_golden(ld):
    """tests/golden/jpeg (made by tests/golden/make_jpeg_golden.py with Pillow / libjpeg-turbo): the native decoder reproduces the stored arrays bit for
    bit - no Pillow needed at test time."""
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg")
    expected = np.load(os.path.join(gdir, "expected_rgb.npz"))
    assert len(expected.files) >= 8
    for name in expected.files:
        rc, got = _native_rgb8(ld, os.path.join(gdir, name + ".jpg"))
        assert rc == 0, (name, got)
        assert got.shape == expected[name].shape and np.array_equal(got, expected[name]), name


def test_libtorch_loader_adapters_equal_python_host_layer(ld, tmp_path):
    """gs::loader::read_colmap_* / read_transforms_* of include/lfs_gsplat_torch.hpp (the signatures of src/loader/formats/colmap.hpp, transforms.hpp
    with torch tensors in CameraData) return exactly what the Python host layer reads through the same C ABI; failures surface as exceptions with
    the reference's message."""
    import json
    import torch
    try:
        from lichtfeld_studio_amd import _lfs_torch_ops as m
    except ImportError:
        pytest.skip("_lfs_torch_ops.so not built (python lichtfeld-studio_amd/build.py --torch-ops)")
    rng = np.random.default_rng(3)
    cams, images, xyz, rgb = _dataset(rng)
    base = str(tmp_path / "scene")
    _write(base, cams, images, xyz, rgb, txt=False)
    _write(base, cams, images, xyz, rgb, txt=True)
    for fn, ref in [(m.read_colmap_cameras_and_images, ld.read_colmap_cameras_and_images), (m.read_colmap_cameras_and_images_text, ld.read_colmap_cameras_and_images_text)]:
        got, center = fn(base, "images")
        views, ref_center = ref(base, "images")
        assert len(got) == len(views) and center.dtype == torch.float32 and np.array_equal(center.numpy(), ref_center)
        for g, v in zip(got, views):
            assert (g["camera_ID"], g["camera_model"], g["camera_model_type"], g["width"], g["height"], g["image_name"], g["image_path"]) == \
                   (v.camera_id, v.colmap_model, v.camera_model_type, v.width, v.height, v.image_name, v.image_path)
            assert (np.float32(g["focal_x"]), np.float32(g["focal_y"]), np.float32(g["center_x"]), np.float32(g["center_y"])) == (v.focal_x, v.focal_y, v.center_x, v.center_y)
            assert g["R"].shape == (3, 3) and np.array_equal(g["R"].numpy(), v.R) and np.array_equal(g["T"].numpy(), v.T)
            assert np.array_equal(g["radial_distortion"].numpy(), v.radial_distortion) and np.array_equal(g["tangential_distortion"].numpy(), v.tangential_distortion)
            assert np.array_equal(g["params"].numpy(), v.params)
    for fn in (m.read_colmap_point_cloud, m.read_colmap_point_cloud_text):
        means, colors = fn(base)
        assert means.dtype == torch.float32 and colors.dtype == torch.uint8
        assert np.array_equal(means.numpy(), xyz.astype(np.float32)) and np.array_equal(colors.numpy(), rgb.astype(np.uint8))
    frame = {"file_path": "images/a.jpg", "transform_matrix": [[1, 0, 0, 1], [0, 1, 0, 2], [0, 0, 1, 3], [0, 0, 0, 1]]}
    p = tmp_path / "ngp.json"
    p.write_text(json.dumps({"w": 80, "h": 60, "fl_x": 70.5, "fl_y": 71.25, "cx": 39.5, "cy": 31.0, "frames": [frame]}))
    got, center = m.read_transforms_cameras_and_images(str(p))
    views, _ = ld.read_transforms_cameras_and_images(str(p))
    assert len(got) == 1 and np.array_equal(got[0]["R"].numpy(), views[0].R) and np.array_equal(got[0]["T"].numpy(), views[0].T)
    assert got[0]["width"] == 80 and got[0]["focal_y"] == 71.25 and not center.any()
    with pytest.raises(RuntimeError, match="could not find transforms_train.json nor transforms.json"):
        m.read_transforms_cameras_and_images(str(tmp_path / "scene"))
    with pytest.raises(RuntimeError):
        m.read_colmap_cameras_and_images(str(tmp_path / "nothing"), "images")
    # splat PLY: the C++ adapter writes the same bytes as the Python host layer and reads them back
    from lichtfeld_studio_amd.rasterizer import SplatModel
    N, K = 37, 16
    t = dict(means=torch.randn(N, 3), sh0=torch.randn(N, 1, 3), shN=torch.randn(N, K - 1, 3), scales=torch.randn(N, 3), quats=torch.randn(N, 4) * 2, opac=torch.randn(N))
    a, b = str(tmp_path / "cpp.ply"), str(tmp_path / "py.ply")
    m.save_ply(a, t["means"], t["sh0"], t["shN"], t["scales"], t["quats"], t["opac"].unsqueeze(1))   # [N,1] opacity, as SplatData holds it
    ld.save_ply(SplatModel(t["means"], t["sh0"], t["shN"], t["scales"], t["quats"], t["opac"], 3), b)
    assert open(a, "rb").read() == open(b, "rb").read()
    back = m.load_ply(a)
    ref = ld.load_ply(b, device="cpu")
    for x, y in zip(back, ref.parameters()):
        assert x.shape == y.shape and torch.equal(x, y.detach())
    open(a, "wb").write(b"ply\nformat binary_little_endian 1.0\nelement vertex 1\nproperty float x\nproperty float y\nproperty float z\nend_header\n" + bytes(12))
    back = m.load_ply(a)
    # positions only: the reference's defaults (ply.cpp:531-600) - shN [N,15,3] zeros, log-scale -5, identity quaternion, opacity 0
    assert back[1].shape == (1, 1, 3) and back[2].shape == (1, 15, 3) and not back[2].any() and back[5].shape == (1,) and not back[5].any()
    assert back[3].tolist() == [[-5.0, -5.0, -5.0]] and back[4].tolist() == [[1.0, 0.0, 0.0, 0.0]]
    with pytest.raises(RuntimeError, match="No end_header"):
        open(a, "wb").write(b"ply\nformat binary_little_endian 1.0\nelement vertex 5\nproperty float x\n" + b" " * 8)
        m.load_ply(a)


def test_decoders_reject_corrupt_input_without_crashing(ld, tmp_path):
    """Untrusted-input hardening of the native decoders (tools/fuzz_io.cpp is the AddressSanitizer harness these cases came from): crafted Huffman
    tables / DC categories are rejected with an error, and randomly mutated files only ever produce an image or an error code with a message."""
    import glob
    src = open(os.path.join(os.path.dirname(__file__), "golden", "jpeg", "baseline_444_q90.jpg"), "rb").read()
    p = str(tmp_path / "bad.jpg")
    dht = src.index(b"\xff\xc4")
    assert src[dht + 4] >> 4 == 0                                   # the first table is a DC table: Tc = 0
    over = bytearray(src)
    k = max(range(16), key=lambda i: src[dht + 5 + i])              # the most populated code length gives up three codes ...
    assert src[dht + 5 + k] >= 3 and k > 0
    over[dht + 5 + k] -= 3
    over[dht + 5] += 3                                              # ... to length 1, which has room for two
    open(p, "wb").write(bytes(over))
    rc, msg = _native_rgb8(ld, p)
    assert rc == -3 and "Huffman" in msg, (rc, msg)                 # LFS_IO_E_FORMAT
    cat = bytearray(src)
    n_vals = sum(src[dht + 5:dht + 21])
    cat[dht + 21:dht + 21 + n_vals] = bytes([200]) * n_vals         # every DC symbol claims a 200-bit difference
    open(p, "wb").write(bytes(cat))
    rc, msg = _native_rgb8(ld, p)
    assert rc == -3 and "DC difference category" in msg, (rc, msg)
    rng = np.random.default_rng(9)
    seeds = [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "jpeg", "*.jpg")))]
    img = rng.integers(0, 256, (19, 23, 3)).astype(np.uint8)
    ld.write_png(str(tmp_path / "s.png"), img)
    seeds.append(open(tmp_path / "s.png", "rb").read())
    outcomes = {"ok": 0, "error": 0}
    for it in range(400):
        d = bytearray(seeds[it % len(seeds)])
        for _ in range(int(rng.integers(1, 4))):
            kind = int(rng.integers(0, 3))
            if kind == 0:
                d[int(rng.integers(0, len(d)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                del d[int(rng.integers(1, len(d))):]
            else:
                a = int(rng.integers(0, len(d)))
                d[a:a + 4] = bytes(rng.integers(0, 256, 4).astype(np.uint8))
        q = str(tmp_path / ("m.png" if seeds[it % len(seeds)][:4] == b"\x89PNG" else "m.jpg"))
        open(q, "wb").write(bytes(d))
        rc, out = _native_rgb8(ld, q)
        if rc == 0:
            assert out.ndim == 3 and out.shape[2] == 3 and out.dtype == np.uint8
            outcomes["ok"] += 1
        else:
            assert rc in (-3, -4) and out, (rc, out)                # LFS_IO_E_FORMAT / _UNSUPPORTED with a message
            outcomes["error"] += 1
    assert outcomes["ok"] > 20 and outcomes["error"] > 20, outcomes
