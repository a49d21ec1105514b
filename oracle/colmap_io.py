"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): pure-Python / numpy restatement of the reference's data-format code.
  COLMAP binary / text readers      src/loader/formats/colmap.cpp:305-640
  camera assembly, model table      src/loader/formats/colmap.cpp:645-880 (+ scaling :172-283)
  PLY attribute layout              src/core/splat_data.cpp:113-169, :402-419, :484-505
  load_image sizes                  src/core/image_io.cpp:112-270
  OpenImageIO resample              third-party (vcpkg "openimageio", version floating with the vcpkg baseline
                                    4334d8b4c8916018600212ab4dd4bbdc343065d1; not vendored): ImageBufAlgo::resample(interpolate=true)
                                    restated from its published algorithm (sample at the destination pixel centre, bilinear, clamp)
  compute_mean_neighbor_distances   src/core/splat_data.cpp:64-111: nanoflann queried with eps = 10 (approximate) - NanoflannTree below restates the vendored
                                    nanoflann's tree and search; mean_neighbor_distances_exact is the exact 3-NN mean the comment there promises
PINNED to the reference's own code where it builds here (tests/test_loader_reference.py): the COLMAP readers and camera assembly to colmap.cpp compiled against
libtorch (oracle/_ref/libref_colmap.so, tests/golden/ref_colmap.npz); the PLY bytes, the neighbour query and init_model_from_pointcloud to splat_data.cpp
(libref_splat_io.so, tests/golden/ref_splat_io.npz). Still PARITY UNPINNED: load_image's sizes and the OpenImageIO resample (image_io.cpp needs OpenImageIO,
absent here) - checked against Pillow and the published filter. The writers below produce the byte layouts COLMAP documents.
"""
import os
import struct

import numpy as np

MODELS = {  # id: (name, n_params)
    0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5), 4: ("OPENCV", 8), 5: ("OPENCV_FISHEYE", 8),
    6: ("FULL_OPENCV", 12), 7: ("FOV", 5), 8: ("SIMPLE_RADIAL_FISHEYE", 4), 9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}
NAME_TO_ID = {v[0]: k for k, v in MODELS.items()}
F32 = np.float32


# ---- writers (the documented COLMAP layouts) ----
def write_cameras_bin(path, cams):
    """cams: list of (camera_id, model_id, width, height, params)"""
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(cams)))
        for cid, model, w, h, params in cams:
            f.write(struct.pack("<IiQQ", cid, model, w, h))
            f.write(struct.pack(f"<{len(params)}d", *params))


def write_images_bin(path, images, n_points2d=3):
    """images: list of (image_id, qvec[4], tvec[3], camera_id, name)"""
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(images)))
        for k, (iid, q, t, cid, name) in enumerate(images):
            f.write(struct.pack("<I4d3dI", iid, *q, *t, cid))
            f.write(name.encode() + b"\0")
            n = (n_points2d + k) % 5
            f.write(struct.pack("<Q", n))
            for j in range(n):
                f.write(struct.pack("<ddQ", 1.5 * j, 2.5 * j, j))


def write_points3d_bin(path, xyz, rgb):
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(xyz)))
        for i, (p, c) in enumerate(zip(xyz, rgb)):
            f.write(struct.pack("<Q3d3Bd", i + 1, *[float(v) for v in p], *[int(v) for v in c], 0.5))
            track = i % 4
            f.write(struct.pack("<Q", track))
            for j in range(track):
                f.write(struct.pack("<II", j, j + 1))


def write_cameras_txt(path, cams, crlf=False):
    nl = "\r\n" if crlf else "\n"
    with open(path, "w", newline="") as f:
        f.write("# Camera list with one line of data per camera:" + nl + "#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]" + nl)
        for cid, model, w, h, params in cams:
            f.write(f"{cid} {MODELS[model][0]} {w} {h} " + " ".join(repr(float(p)) for p in params) + nl)


def write_images_txt(path, images, crlf=False):
    nl = "\r\n" if crlf else "\n"
    with open(path, "w", newline="") as f:
        f.write("# Image list with two lines of data per image:" + nl)
        for iid, q, t, cid, name in images:
            f.write(f"{iid} " + " ".join(repr(float(v)) for v in (*q, *t)) + f" {cid} {name}" + nl)
            f.write("10.5 20.25 -1 3.5 4.5 7" + nl)


def write_points3d_txt(path, xyz, rgb):
    with open(path, "w") as f:
        f.write("# 3D point list\n")
        for i, (p, c) in enumerate(zip(xyz, rgb)):
            f.write(f"{i + 1} {float(p[0])!r} {float(p[1])!r} {float(p[2])!r} {int(c[0])} {int(c[1])} {int(c[2])} 0.5 1 2 3 4\n")


# ---- readers / assembly (restatement of the reference) ----
def folder_scale(folder):
    """colmap.cpp:265-283"""
    us = folder.rfind("_")
    if us < 0:
        return 1.0
    try:
        v = float(F32(float(folder[us + 1:])))
    except ValueError:
        return 1.0
    return v if 0 < v <= 16 else 1.0


def _scaled(model, w, h, params, factor):
    params = [float(p) for p in params]
    if factor != 1.0:
        w, h = int(F32(w) / F32(factor)), int(F32(h) / F32(factor))
        n = 3 if MODELS[model][0] in ("SIMPLE_PINHOLE", "SIMPLE_RADIAL", "RADIAL", "SIMPLE_RADIAL_FISHEYE", "RADIAL_FISHEYE") else 4
        for i in range(n):
            params[i] = params[i] / float(F32(factor))
    return w, h, np.array(params, np.float64).astype(F32)


def read_cameras_bin(path, factor=1.0):
    b = open(path, "rb").read()
    n, = struct.unpack_from("<Q", b, 0)
    o, out = 8, {}
    for _ in range(n):
        cid, model, w, h = struct.unpack_from("<IiQQ", b, o)
        o += 24
        if model not in MODELS:
            raise RuntimeError(f"Unsupported camera-model id {model}")
        k = MODELS[model][1]
        params = struct.unpack_from(f"<{k}d", b, o)
        o += 8 * k
        out[cid] = (model,) + _scaled(model, w, h, params, factor)
    if o != len(b):
        raise RuntimeError("cameras.bin: trailing bytes")
    return out


def read_images_bin(path):
    b = open(path, "rb").read()
    n, = struct.unpack_from("<Q", b, 0)
    o, out = 8, []
    for _ in range(n):
        iid, = struct.unpack_from("<I", b, o)
        q = np.array(struct.unpack_from("<4d", b, o + 4)).astype(F32)
        t = np.array(struct.unpack_from("<3d", b, o + 36)).astype(F32)
        cid, = struct.unpack_from("<I", b, o + 60)
        o += 64
        e = b.index(b"\0", o)
        name = b[o:e].decode()
        o = e + 1
        npts, = struct.unpack_from("<Q", b, o)
        o += 8 + 24 * npts
        out.append((iid, q, t, cid, name))
    if o != len(b):
        raise RuntimeError("images.bin: trailing bytes")
    return out


def _lines(path):
    """colmap.cpp:459-488"""
    lines = []
    with open(path, newline="") as f:
        text = f.read()
        raw = text.split("\n")
        if raw and raw[-1] == "":   # std::getline does not produce an extra line after a final newline
            raw.pop()
        for line in raw:
            if line.startswith("#"):
                continue
            if line.endswith("\r"):
                line = line[:-1]
            lines.append(line)
    if not lines:
        raise RuntimeError("empty")
    if lines[-1] == "":
        lines.pop()
    return lines


def read_cameras_txt(path, factor=1.0):
    out = {}
    for line in _lines(path):
        tok = line.split(" ")
        model = NAME_TO_ID[tok[1]]
        out[int(tok[0])] = (model,) + _scaled(model, int(tok[2]), int(tok[3]), [float(t) for t in tok[4:]], factor)
    return out


def read_images_txt(path):
    lines = _lines(path)
    assert len(lines) % 2 == 0
    out = []
    for i in range(0, len(lines), 2):
        tok = lines[i].split(" ")
        assert len(tok) == 10
        out.append((int(tok[0]), np.array([F32(float(v)) for v in tok[1:5]], F32), np.array([F32(float(v)) for v in tok[5:8]], F32), int(tok[8]), tok[9]))
    return out


def qvec2rotmat(q):
    """colmap.cpp:29-50, float32 throughout"""
    q = q.astype(F32)
    den = max(F32(np.sqrt(F32(np.sum(q * q, dtype=F32)))), F32(1e-12))
    w, x, y, z = [F32(v / den) for v in q]
    one, two = F32(1), F32(2)
    return np.array([[one - two * (y * y + z * z), two * (x * y - z * w), two * (x * z + y * w)],
                     [two * (x * y + z * w), one - two * (x * x + z * z), two * (y * z - x * w)],
                     [two * (x * z - y * w), two * (y * z + x * w), one - two * (x * x + y * y)]], F32)


# (focal count, radial indices, tangential indices, projection type) per model; None = rejected (colmap.cpp:682-830)
LAYOUT = {0: (1, [], [], 0), 1: (2, [], [], 0), 2: (1, [3], [], 0), 3: (1, [3, 4], [], 0), 4: (2, [4, 5], [6, 7], 0), 5: (2, [4, 5, 6, 7], [], 2),
          6: (2, [4, 5, 8, 9, 10, 11], [6, 7], 0), 7: None, 8: (1, [3], [], 2), 9: (1, [3, 4], [], 2), 10: None}


def assemble(cams, images, first_image_size=None):
    """-> list of dicts + scene centre. first_image_size (w, h): the real size of image 0 if the file exists (colmap.cpp:836-865)."""
    out, locs = [], []
    for iid, q, t, cid, name in images:
        model, w, h, p = cams[cid]
        R = qvec2rotmat(q)
        locs.append(-(R.T.astype(F32) @ t.astype(F32)).astype(F32))
        if LAYOUT[model] is None:
            raise RuntimeError("not supported")
        nf, rad, tan, proj = LAYOUT[model]
        v = dict(camera_id=cid, colmap_model=model, camera_model_type=proj, width=w, height=h, focal_x=p[0], focal_y=p[1] if nf == 2 else p[0],
                 center_x=p[nf], center_y=p[nf + 1], R=R, T=t.astype(F32), radial=np.array([p[i] for i in rad], F32),
                 tangential=np.array([p[i] for i in tan], F32), params=p, name=name)
        if model == 2 and p[3] == 0:
            v["radial"] = np.zeros(0, F32)
        out.append(v)
    if out and first_image_size is not None:
        sx, sy = F32(first_image_size[0]) / F32(out[0]["width"]), F32(first_image_size[1]) / F32(out[0]["height"])
        if abs(sx - 1) > 1e-5 or abs(sy - 1) > 1e-5:
            for v in out:
                v["width"], v["height"] = first_image_size
                v["focal_x"] = F32(v["focal_x"] * sx); v["focal_y"] = F32(v["focal_y"] * sy)
                v["center_x"] = F32(v["center_x"] * sx); v["center_y"] = F32(v["center_y"] * sy)
    return out, np.mean(np.stack(locs).astype(np.float64), 0).astype(F32) if locs else None


# ---- images ----
def target_size(w, h, res_div, max_width):
    """image_io.cpp:112-270 (sizes only)"""
    nw, nh = w, h
    if res_div in (2, 4, 8):
        nw, nh = max(1, w // res_div), max(1, h // res_div)
    if max_width > 0 and (nw > max_width or nh > max_width):
        if nw > nh:
            nw, nh = max(1, max_width), max(1, max_width * nh // nw)
        else:
            nw, nh = max(1, max_width * nw // nh), max(1, max_width)
    return nw, nh


def resample_u8(src, dw, dh):
    """OpenImageIO ImageBufAlgo::resample(interpolate=true) on a u8 [h,w,3] image -> u8 [dh,dw,3], float32 arithmetic."""
    sh, sw = src.shape[:2]
    x, y = np.arange(dw, dtype=F32), np.arange(dh, dtype=F32)
    fx = ((x + F32(0.5)) * F32(F32(1) / F32(dw))) * F32(sw) - F32(0.5)
    fy = ((y + F32(0.5)) * F32(F32(1) / F32(dh))) * F32(sh) - F32(0.5)
    flx, fly = np.floor(fx), np.floor(fy)
    ax, ay = (fx - flx).astype(F32)[None, :, None], (fy - fly).astype(F32)[:, None, None]
    x0, x1 = np.clip(flx.astype(int), 0, sw - 1), np.clip(flx.astype(int) + 1, 0, sw - 1)
    y0, y1 = np.clip(fly.astype(int), 0, sh - 1), np.clip(fly.astype(int) + 1, 0, sh - 1)
    f = src.astype(F32) * F32(F32(1) / F32(255))
    v00, v01, v10, v11 = f[y0][:, x0], f[y0][:, x1], f[y1][:, x0], f[y1][:, x1]
    one = F32(1)
    top = (one - ax) * v00 + ax * v01
    bot = (one - ax) * v10 + ax * v11
    v = (one - ay) * top + ay * bot
    return np.clip(v * F32(255) + F32(0.5), 0, 255).astype(np.int32).astype(np.uint8)


def image_to_chw(src_u8, dw=None, dh=None):
    h, w = src_u8.shape[:2]
    if (dw or w) != w or (dh or h) != h:
        src_u8 = resample_u8(src_u8, dw, dh)
    return (src_u8.astype(F32) / F32(255)).transpose(2, 0, 1)


# ---- point cloud ----
def mean_neighbor_distances_exact(points):
    """splat_data.cpp:64-111 with an exact brute-force search (float32 squared distances accumulated in x, y, z order): what the function's comment says it
    computes. The reference's actual query is approximate - see mean_neighbor_distances."""
    p = points.astype(F32)
    n = len(p)
    if n <= 1:
        return np.full(n, 0.01, F32)
    out = np.zeros(n, F32)
    for i in range(n):
        d = p[i] - p
        d2 = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(F32) + d[:, 2] * d[:, 2]).astype(F32)
        best = np.sort(d2, kind="stable")[:min(4, n)]
        out[i] = _mean_of_results(best)
    return out


def _mean_of_results(best):
    """splat_data.cpp:99-109: up to 3 of the (ascending) results with d^2 > 1e-8"""
    vals = [F32(np.sqrt(F32(v))) for v in best if v > F32(1e-8)][:3]
    s = F32(0)
    for v in vals:
        s = F32(s + v)
    return F32(s / F32(len(vals))) if vals else F32(0.01)


class NanoflannTree:
    """The kd-tree the reference queries (vendored include/external/nanoflann.hpp v1.7.1, KDTreeSingleIndexAdaptor<L2_Simple_Adaptor<float>, ., 3> built with
    KDTreeSingleIndexAdaptorParams(10): leaf size 10, one build thread), restated from the published algorithm: divideTree (:1056-1108), middleSplit_
    (:1199-1247), planeSplit (:1258-1297), computeMinMax (:1035-1047), searchLevel (:1740-1805), KNNResultSet::addPoint / worstDist (:222-257). float32
    throughout, python floats only where a value is exactly representable. Pinned by tests/golden/ref_splat_io.npz (the reference's function run on the CPU)."""

    def __init__(self, points, leaf=10):
        self.p = np.ascontiguousarray(points, F32)
        n = len(self.p)
        self.order = list(range(n))
        self.leaf = leaf
        lo, hi = self.p.min(0), self.p.max(0)
        box = [[F32(lo[d]), F32(hi[d])] for d in range(3)]
        self.root = self._divide(0, n, box)
        self.root_box = box                               # divideTree leaves the union of the leaves' boxes in root_bbox_

    def _at(self, k, d):
        return self.p[self.order[k], d]

    def _plane_split(self, ind, count, d, cutval):
        o, at = self.order, self._at
        left, right = 0, count - 1
        while True:
            while left <= right and at(ind + left, d) < cutval:
                left += 1
            while right and left <= right and at(ind + right, d) >= cutval:
                right -= 1
            if left > right or not right:
                break
            o[ind + left], o[ind + right] = o[ind + right], o[ind + left]
            left += 1
            right -= 1
        lim1 = left
        right = count - 1
        while True:
            while left <= right and at(ind + left, d) <= cutval:
                left += 1
            while right and left <= right and at(ind + right, d) > cutval:
                right -= 1
            if left > right or not right:
                break
            o[ind + left], o[ind + right] = o[ind + right], o[ind + left]
            left += 1
            right -= 1
        return lim1, left

    def _middle_split(self, ind, count, box):
        eps = F32(0.00001)
        spans = [F32(box[d][1] - box[d][0]) for d in range(3)]
        max_span = max(spans)
        max_spread, cutfeat, min_elem, max_elem = F32(-1), 0, F32(0), F32(0)
        sub = self.p[self.order[ind:ind + count]]
        for d in range(3):
            if spans[d] >= F32(F32(1 - eps) * max_span):
                lo, hi = sub[:, d].min(), sub[:, d].max()
                if F32(hi - lo) > max_spread:
                    cutfeat, max_spread, min_elem, max_elem = d, F32(hi - lo), lo, hi
        split = F32(F32(box[cutfeat][0] + box[cutfeat][1]) / F32(2))
        cutval = min_elem if split < min_elem else max_elem if split > max_elem else split
        lim1, lim2 = self._plane_split(ind, count, cutfeat, cutval)
        half = count // 2
        index = lim1 if lim1 > half else lim2 if lim2 < half else half
        return index, cutfeat, cutval

    def _divide(self, left, right, box):
        if right - left <= self.leaf:
            sub = self.p[self.order[left:right]]
            for d in range(3):
                box[d][0], box[d][1] = sub[:, d].min(), sub[:, d].max()
            return ("leaf", left, right)
        idx, cutfeat, cutval = self._middle_split(left, right - left, box)
        lb, rb = [list(b) for b in box], [list(b) for b in box]
        lb[cutfeat][1] = cutval
        c1 = self._divide(left, left + idx, lb)
        rb[cutfeat][0] = cutval
        c2 = self._divide(left + idx, right, rb)
        for d in range(3):
            box[d][0], box[d][1] = min(lb[d][0], rb[d][0]), max(lb[d][1], rb[d][1])
        return ("node", cutfeat, lb[cutfeat][1], rb[cutfeat][0], c1, c2)

    def knn(self, vec, k, eps):
        """findNeighbors (:1570-1596) with a KNNResultSet of capacity k -> ascending squared distances, zero-padded like the caller's std::vector"""
        vec = np.asarray(vec, F32)
        res = []                                          # ascending; worstDist() = FLT_MAX until k entries

        def worst():
            return F32(np.finfo(F32).max) if len(res) < k else res[-1]

        def add(dist):
            i = len(res)
            while i > 0 and res[i - 1] > dist:
                i -= 1
            if i < k:
                res.insert(i, dist)
                del res[k:]

        eps_error = F32(1 + eps)
        side = [F32(0)] * 3
        mind = F32(0)
        for d in range(3):                                # computeInitialDistances (:1299-1316)
            if vec[d] < self.root_box[d][0]:
                side[d] = F32(F32(vec[d] - self.root_box[d][0]) * F32(vec[d] - self.root_box[d][0]))
                mind = F32(mind + side[d])
            if vec[d] > self.root_box[d][1]:
                side[d] = F32(F32(vec[d] - self.root_box[d][1]) * F32(vec[d] - self.root_box[d][1]))
                mind = F32(mind + side[d])

        def level(node, mindist):
            if node[0] == "leaf":
                w = worst()                               # read once per leaf
                for i in range(node[1], node[2]):
                    diff = vec - self.p[self.order[i]]
                    sq = (diff * diff).astype(F32)
                    dist = F32(F32(sq[0] + sq[1]) + sq[2])
                    if dist < w:
                        add(dist)
                return
            _, feat, divlow, divhigh, c1, c2 = node
            val = vec[feat]
            diff1, diff2 = F32(val - divlow), F32(val - divhigh)
            if F32(diff1 + diff2) < 0:
                best, other, cut = c1, c2, F32(diff2 * diff2)
            else:
                best, other, cut = c2, c1, F32(diff1 * diff1)
            level(best, mindist)
            dst = side[feat]
            mindist = F32(F32(mindist + cut) - dst)
            side[feat] = cut
            if F32(mindist * eps_error) <= worst():
                level(other, mindist)
            side[feat] = dst

        level(self.root, mind)
        return res + [F32(0)] * (k - len(res))


def mean_neighbor_distances(points, eps=10.0):
    """compute_mean_neighbor_distances (splat_data.cpp:64-111) as the reference runs it: nanoflann::SearchParameters(10) (:97) sets eps = 10, so each query is a
    (1 + eps)-approximate 4-nearest search on the tree above; up to 3 of the results with d^2 > 1e-8 are averaged."""
    p = np.ascontiguousarray(points, F32)
    n = len(p)
    if n <= 1:
        return np.full(n, 0.01, F32)
    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 10000))
    tree = NanoflannTree(p)
    return np.array([_mean_of_results(tree.knn(p[i], min(4, n), eps)) for i in range(n)], F32)


# ---- PLY ----
def ply_attribute_names(n_dc, n_rest):
    """splat_data.cpp:402-419"""
    return (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(n_dc)] + [f"f_rest_{i}" for i in range(n_rest)] + ["opacity"]
            + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)])


def ply_bytes(means, sh0, shN, opacity, scaling, rotation):
    """The file SplatData::save_ply writes (tinyply binary little endian, float properties, vertex-interleaved)."""
    N = len(means)
    f_dc = np.transpose(sh0, (0, 2, 1)).reshape(N, -1)
    f_rest = np.transpose(shN, (0, 2, 1)).reshape(N, -1)
    rot = rotation / np.maximum(np.linalg.norm(rotation, axis=-1, keepdims=True), 1e-12)
    rows = np.concatenate([means, np.zeros_like(means), f_dc, f_rest, opacity.reshape(N, 1), scaling, rot], 1).astype("<f4")
    names = ply_attribute_names(f_dc.shape[1], f_rest.shape[1])
    hdr = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % N + "".join(f"property float {n}\n" for n in names) + "end_header\n"
    return hdr.encode() + rows.tobytes()


# ---- Blender / NeRF-synthetic transforms.json (src/loader/formats/transforms.cpp:73-265), with the reference's own torch float32 operations ----
def read_transforms(path, first_image_size=None):
    import json
    import math
    import re

    import torch
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"(?m)//[^\n\"]*$", "", text)           # comments (nlohmann: ignore_comments = true)
    t = json.loads(text)
    if "w" in t and "h" in t:
        w, h = int(t["w"]), int(t["h"])
    else:
        w, h = first_image_size
    f32 = np.float32
    focal = lambda res, fov: f32(0.5) * f32(res) / f32(math.tan(f32(0.5) * f32(fov)))
    fl_x = f32(t["fl_x"]) if "fl_x" in t else focal(w, t["camera_angle_x"])
    if "fl_y" in t:
        fl_y = f32(t["fl_y"])
    elif "camera_angle_y" in t:
        fl_y = focal(h, t["camera_angle_y"])
    else:
        assert w == h
        fl_y = fl_x
    cx = f32(t["cx"]) if "cx" in t else f32(0.5 * w)
    cy = f32(t["cy"]) if "cy" in t else f32(0.5 * h)
    ang = f32(math.pi)
    fix = torch.eye(4)
    fix[0, 0] = float(np.cos(ang)); fix[0, 2] = float(np.sin(ang)); fix[2, 0] = -float(np.sin(ang)); fix[2, 2] = float(np.cos(ang))
    out = []
    for k, fr in enumerate(t.get("frames", [])):
        c2w = torch.tensor(fr["transform_matrix"], dtype=torch.float32)
        c2w[:3, 1:3] *= -1
        w2c = torch.mm(torch.inverse(c2w), fix)
        out.append(dict(camera_id=k, width=w, height=h, focal_x=fl_x, focal_y=fl_y, center_x=cx, center_y=cy, R=w2c[:3, :3].numpy().copy(),
                        T=w2c[:3, 3].numpy().copy(), file_path=fr["file_path"]))
    return out
