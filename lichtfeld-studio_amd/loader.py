"""Data formats either side of the training path: the host-side mirror of the reference's loaders over
liblfs_io.so (include/lfs_io.h; C++17, no GPU) plus the two GPU kernels of csrc/dataprep.hip.

  read_colmap_cameras_and_images[_text]  src/loader/formats/colmap.cpp:913-957    -> (list[CameraData], scene_center)
  read_colmap_point_cloud[_text]         src/loader/formats/colmap.cpp:907-937    -> PointCloud(means f32 [N,3], colors u8 [N,3])
  load_image                             src/core/image_io.cpp:112-270 + src/core/camera.cpp:101-140 -> f32 [3,h,w] on the GPU
  init_model_from_pointcloud             src/core/splat_data.cpp:508-614          -> (SplatModel, scene_scale)
  save_ply / load_ply                    src/core/splat_data.cpp:113-169, :402-419 / src/loader/formats/ply.cpp
  CameraDataset                          src/training/dataset.hpp:25-75 (every test_every-th image is a validation view)
  colmap_scene                           the Scene the trainer consumes (viewmats [R|t], K scaled to the loaded image size:
                                         src/core/camera.cpp:15-23, :77-98)

liblfs_io decodes PNG, PNM and JPEG itself (baseline and progressive Huffman JPEG, bit-identical to libjpeg-turbo's default pipeline, i.e. to what
the reference gets through OpenImageIO); what it reports as unsupported (CMYK JPEG, interlaced PNG, TIFF ...) is decoded with Pillow here.
Decoding is host work either way; everything after the decoded bytes (resample, CHW, float) runs in one HIP kernel.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_IO = None

IO_EXPORTS = [
    "lfs_io_last_error", "lfs_io_version", "lfs_io_free", "lfs_colmap_open", "lfs_transforms_open", "lfs_colmap_close", "lfs_colmap_num_views", "lfs_colmap_view_at",
    "lfs_colmap_image_name", "lfs_colmap_image_path", "lfs_colmap_scene_center", "lfs_colmap_points_open", "lfs_point_cloud_size",
    "lfs_point_cloud_copy", "lfs_point_cloud_close", "lfs_ply_write_splat", "lfs_ply_open", "lfs_ply_num_vertices", "lfs_ply_num_properties",
    "lfs_ply_property_name", "lfs_ply_read", "lfs_ply_close", "lfs_image_info", "lfs_image_target_size", "lfs_image_load_rgb8",
    "lfs_image_write_png_rgb8",
]
IO_E_UNSUPPORTED = -4


class LoaderError(RuntimeError):
    """What the reference throws as std::runtime_error from its loaders."""


class _View(C.Structure):
    _fields_ = [("camera_id", C.c_uint32), ("colmap_model", C.c_int32), ("camera_model_type", C.c_int32), ("width", C.c_uint64), ("height", C.c_uint64),
                ("focal_x", C.c_float), ("focal_y", C.c_float), ("center_x", C.c_float), ("center_y", C.c_float), ("R", C.c_float * 9), ("T", C.c_float * 3),
                ("n_radial", C.c_int32), ("radial", C.c_float * 6), ("n_tangential", C.c_int32), ("tangential", C.c_float * 2),
                ("n_params", C.c_int32), ("params", C.c_float * 12)]


def io_library_path() -> str:
    return os.path.join(_HERE, "liblfs_io.so")


def io_library():
    global _IO
    if _IO is None:
        path = io_library_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it first (python lichtfeld-studio_amd/build.py)")
        lib = C.CDLL(path)
        for name in IO_EXPORTS:
            if not hasattr(lib, name):
                raise RuntimeError(f"{path} does not export {name}")
        for name in ("lfs_io_last_error", "lfs_io_version", "lfs_colmap_image_name", "lfs_colmap_image_path", "lfs_ply_property_name"):
            getattr(lib, name).restype = C.c_char_p
        for name in ("lfs_colmap_num_views", "lfs_point_cloud_size", "lfs_ply_num_vertices"):
            getattr(lib, name).restype = C.c_uint64
        lib.lfs_ply_num_properties.restype = C.c_uint32
        lib.lfs_io_free.restype = None
        lib.lfs_io_free.argtypes = [C.c_void_p]
        for name in ("lfs_colmap_close", "lfs_point_cloud_close", "lfs_ply_close"):
            getattr(lib, name).restype = None
            getattr(lib, name).argtypes = [C.c_void_p]
        _IO = lib
    return _IO


def _check(rc: int) -> None:
    if rc != 0:
        raise LoaderError(io_library().lfs_io_last_error().decode(errors="replace"))


# ---------------------------------------------------------------------------------------------------------------------
# COLMAP
# ---------------------------------------------------------------------------------------------------------------------
@dataclass
class CameraData:
    """include/loader/... CameraData as filled by read_colmap_cameras (colmap.cpp:645-830)."""
    camera_id: int
    colmap_model: int
    camera_model_type: int          # gsplat::CameraModelType: 0 PINHOLE, 2 FISHEYE
    width: int
    height: int
    focal_x: float
    focal_y: float
    center_x: float
    center_y: float
    R: np.ndarray                   # [3,3] float32 world-to-camera
    T: np.ndarray                   # [3]
    radial_distortion: np.ndarray
    tangential_distortion: np.ndarray
    params: np.ndarray
    image_name: str
    image_path: str


@dataclass
class PointCloud:
    means: np.ndarray               # [N,3] float32
    colors: np.ndarray              # [N,3] uint8


def _read_colmap(base: str, images_folder: Optional[str], fmt: int) -> Tuple[List[CameraData], np.ndarray]:
    lib = io_library()
    h = C.c_void_p()
    if images_folder is None:
        _check(lib.lfs_transforms_open(os.fsencode(base), C.byref(h)))
    else:
        _check(lib.lfs_colmap_open(os.fsencode(base), images_folder.encode(), C.c_int(fmt), C.byref(h)))
    try:
        out = []
        v = _View()
        for i in range(lib.lfs_colmap_num_views(h)):
            _check(lib.lfs_colmap_view_at(h, C.c_uint64(i), C.byref(v)))
            out.append(CameraData(v.camera_id, v.colmap_model, v.camera_model_type, int(v.width), int(v.height), v.focal_x, v.focal_y, v.center_x, v.center_y,
                                  np.array(v.R, np.float32).reshape(3, 3), np.array(v.T, np.float32), np.array(v.radial[:v.n_radial], np.float32),
                                  np.array(v.tangential[:v.n_tangential], np.float32), np.array(v.params[:v.n_params], np.float32),
                                  lib.lfs_colmap_image_name(h, C.c_uint64(i)).decode(), os.fsdecode(lib.lfs_colmap_image_path(h, C.c_uint64(i)))))
        center = (C.c_float * 3)()
        _check(lib.lfs_colmap_scene_center(h, center))
        return out, np.array(center, np.float32)
    finally:
        lib.lfs_colmap_close(h)


def read_colmap_cameras_and_images(base: str, images_folder: str = "images") -> Tuple[List[CameraData], np.ndarray]:
    return _read_colmap(base, images_folder, 0)


def read_colmap_cameras_and_images_text(base: str, images_folder: str = "images") -> Tuple[List[CameraData], np.ndarray]:
    return _read_colmap(base, images_folder, 1)


def read_transforms_cameras_and_images(path: str) -> Tuple[List[CameraData], np.ndarray]:
    """Blender / NeRF-synthetic transforms.json (src/loader/formats/transforms.cpp:73-265)."""
    return _read_colmap(path, None, 0)


def generate_random_point_cloud() -> PointCloud:
    """transforms.cpp:267-283: 10 000 points in [-1,1]^3 with random colours, torch CPU generator seeded with 8128 (so the same bits as the reference)."""
    g = torch.Generator().manual_seed(8128)
    positions = torch.rand((10000, 3), generator=g) * 2.0 - 1.0
    colors = torch.randint(0, 256, (10000, 3), dtype=torch.uint8, generator=g)
    return PointCloud(positions.numpy(), colors.numpy())


def _read_points(base: str, fmt: int) -> PointCloud:
    lib = io_library()
    h = C.c_void_p()
    _check(lib.lfs_colmap_points_open(os.fsencode(base), C.c_int(fmt), C.byref(h)))
    try:
        n = lib.lfs_point_cloud_size(h)
        pos, col = np.empty((n, 3), np.float32), np.empty((n, 3), np.uint8)
        _check(lib.lfs_point_cloud_copy(h, pos.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p)))
        return PointCloud(pos, col)
    finally:
        lib.lfs_point_cloud_close(h)


def read_colmap_point_cloud(base: str) -> PointCloud:
    return _read_points(base, 0)


def read_colmap_point_cloud_text(base: str) -> PointCloud:
    return _read_points(base, 1)


# ---------------------------------------------------------------------------------------------------------------------
# images
# ---------------------------------------------------------------------------------------------------------------------
def get_image_info(path: str) -> Tuple[int, int, int]:
    w, h, c = C.c_int32(), C.c_int32(), C.c_int32()
    _check(io_library().lfs_image_info(os.fsencode(path), C.byref(w), C.byref(h), C.byref(c)))
    return w.value, h.value, c.value


def image_target_size(w: int, h: int, resize_factor: int = -1, max_width: int = 0) -> Tuple[int, int]:
    ow, oh = C.c_int32(), C.c_int32()
    _check(io_library().lfs_image_target_size(C.c_int32(w), C.c_int32(h), C.c_int32(resize_factor), C.c_int32(max_width), C.byref(ow), C.byref(oh)))
    return ow.value, oh.value


def decode_rgb8(path: str) -> np.ndarray:
    """-> uint8 [h,w,3]; native decoder first, Pillow for what liblfs_io does not decode."""
    lib = io_library()
    data, w, h = C.POINTER(C.c_uint8)(), C.c_int32(), C.c_int32()
    rc = lib.lfs_image_load_rgb8(os.fsencode(path), C.byref(data), C.byref(w), C.byref(h))
    if rc == 0:
        try:
            return np.ctypeslib.as_array(data, shape=(h.value, w.value, 3)).copy()
        finally:
            lib.lfs_io_free(data)
    if rc != IO_E_UNSUPPORTED:
        _check(rc)
    from PIL import Image, UnidentifiedImageError
    try:
        im = Image.open(path)
    except (UnidentifiedImageError, OSError) as e:
        raise LoaderError(f"Load failed: {path} : {e}") from None
    with im:
        bands = len(im.getbands())
        if im.mode in ("P", "1", "I;16", "I", "F", "CMYK", "YCbCr"):
            im = im.convert("RGB")
            bands = 3
        a = np.asarray(im)
    if a.ndim == 2:
        a = a[..., None]
    if bands >= 3:
        return np.ascontiguousarray(a[..., :3])
    if bands == 1:
        return np.ascontiguousarray(np.repeat(a[..., :1], 3, -1))
    r, g = a[..., 0].astype(np.int32), a[..., 1].astype(np.int32)      # 2 channels -> (r, g, (r + g) / 2), image_io.cpp:235-247
    return np.stack([r, g, (r + g) // 2], -1).astype(np.uint8)


def u8_to_chw_f32(image_u8: torch.Tensor, out_width: Optional[int] = None, out_height: Optional[int] = None) -> torch.Tensor:
    """GPU: u8 [h,w,3] -> f32 [3,oh,ow] in [0,1] (bilinear resample + 8-bit requantisation when the size changes)."""
    from .capi import check, load_library, ptr, require_gpu, stream
    image_u8 = image_u8.contiguous()
    require_gpu(image_u8)
    assert image_u8.dtype == torch.uint8 and image_u8.dim() == 3 and image_u8.shape[2] == 3, image_u8.shape
    h, w = image_u8.shape[:2]
    ow, oh = out_width or w, out_height or h
    out = torch.empty((3, oh, ow), dtype=torch.float32, device=image_u8.device)
    check(load_library().lfs_image_u8_to_chw_f32(ptr(image_u8), C.c_uint32(w), C.c_uint32(h), ptr(out), C.c_uint32(ow), C.c_uint32(oh), stream()), "image_u8_to_chw_f32")
    return out


def load_image(path: str, resize_factor: int = -1, max_width: int = 0, device="cuda:0") -> torch.Tensor:
    """Camera::load_and_get_image: decode -> [resample] -> f32 [3,h,w] on the device. The upload is the full-size u8 image
    from pinned memory; resample, layout change and normalisation are one kernel."""
    rgb = decode_rgb8(path)
    h, w = rgb.shape[:2]
    ow, oh = image_target_size(w, h, resize_factor, max_width)
    host = torch.from_numpy(rgb)
    if torch.cuda.is_available():
        host = host.pin_memory()
    return u8_to_chw_f32(host.to(device, non_blocking=True), ow, oh)


def write_png(path: str, rgb: np.ndarray) -> None:
    rgb = np.ascontiguousarray(rgb, np.uint8)
    assert rgb.ndim == 3 and rgb.shape[2] == 3
    _check(io_library().lfs_image_write_png_rgb8(os.fsencode(path), rgb.ctypes.data_as(C.c_void_p), C.c_int32(rgb.shape[1]), C.c_int32(rgb.shape[0])))


# ---------------------------------------------------------------------------------------------------------------------
# point cloud -> model, PLY
# ---------------------------------------------------------------------------------------------------------------------
def mean_neighbor_distances(points: torch.Tensor, exact: bool = False) -> torch.Tensor:
    """compute_mean_neighbor_distances (splat_data.cpp:64-111): [N,3] -> [N]. The default reproduces the reference's eps = 10 approximate nanoflann query bit
    for bit (host-built kd-tree, GPU walk; see csrc/dataprep.hip); exact=True is the exact 3-nearest-neighbour mean (an extension)."""
    from .capi import check, load_library, ptr, require_gpu, stream
    points = points.contiguous().float()
    require_gpu(points)
    out = torch.empty(points.shape[0], dtype=torch.float32, device=points.device)
    fn = load_library().lfs_mean_neighbor_distances_exact if exact else load_library().lfs_mean_neighbor_distances
    check(fn(C.c_uint32(points.shape[0]), ptr(points), ptr(out), stream()), "mean_neighbor_distances")
    return out


def init_model_from_pointcloud(pcd: PointCloud, scene_center, sh_degree: int = 3, init_scaling: float = 1.0, init_opacity: float = 0.1, device="cuda:0"):
    """SplatData::init_model_from_pointcloud (splat_data.cpp:508-614), the non-random branch -> (SplatModel, scene_scale)."""
    from .rasterizer import SplatModel
    means = torch.from_numpy(np.ascontiguousarray(pcd.means, np.float32)).to(device)
    colors = torch.from_numpy(np.ascontiguousarray(pcd.colors)).to(device).float() / 255.0
    center = torch.as_tensor(np.asarray(scene_center, np.float32)).to(device)
    scene_scale = float(torch.norm(means - center, 2, 1).median())
    nn_dist = torch.clamp_min(mean_neighbor_distances(means), 1e-7)
    scaling = torch.log(torch.sqrt(nn_dist) * init_scaling).unsqueeze(-1).repeat(1, 3)
    rotation = torch.zeros((means.shape[0], 4), device=device)
    rotation[:, 0] = 1
    opacity = torch.logit(init_opacity * torch.ones((means.shape[0], 1), device=device))
    K = (sh_degree + 1) ** 2
    sh0 = ((colors - 0.5) / 0.28209479177387814).unsqueeze(1).contiguous()           # [N,1,3]
    shN = torch.zeros((means.shape[0], K - 1, 3), device=device)
    mk = lambda t: t.contiguous().requires_grad_(True)
    # a point-cloud initialisation starts at SH degree 0 (SplatData::_active_sh_degree{0}, splat_data.cpp:211); the strategies raise it
    model = SplatModel(mk(means), mk(sh0), mk(shN), mk(scaling), mk(rotation), mk(opacity.squeeze(-1)), sh_degree, active_sh_degree=0)
    return model, scene_scale


def save_ply(model, path: str) -> None:
    """SplatData::save_ply (to_point_cloud :484-505 + write_ply_impl :113-169): f_dc / f_rest channel-major, normals zero,
    rotation normalised, opacity / scaling raw."""
    means, sh0, shN, scales, quats, opac = [p.detach() for p in model.parameters()]
    N = means.shape[0]
    f_dc = sh0.transpose(1, 2).flatten(1).float().cpu().contiguous().numpy()
    f_rest = shN.transpose(1, 2).flatten(1).float().cpu().contiguous().numpy()
    rot = torch.nn.functional.normalize(quats, dim=-1).float().cpu().contiguous().numpy()
    arrs = [means.float().cpu().contiguous().numpy(), f_dc, f_rest, opac.reshape(N).float().cpu().contiguous().numpy(), scales.float().cpu().contiguous().numpy(), rot]
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a.size else None
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    _check(io_library().lfs_ply_write_splat(os.fsencode(path), C.c_uint64(N), C.c_uint32(f_dc.shape[1]), C.c_uint32(f_rest.shape[1]), p(arrs[0]), None, p(f_dc),
                                            p(f_rest), p(arrs[3]), p(arrs[4]), p(rot)))


def read_ply(path: str) -> Tuple[List[str], np.ndarray]:
    """-> (property names, float32 [N,P]) of the vertex element."""
    lib = io_library()
    h = C.c_void_p()
    _check(lib.lfs_ply_open(os.fsencode(path), C.byref(h)))
    try:
        n, P = lib.lfs_ply_num_vertices(h), lib.lfs_ply_num_properties(h)
        names = [lib.lfs_ply_property_name(h, C.c_uint32(i)).decode() for i in range(P)]
        data = np.empty((n, P), np.float32)
        _check(lib.lfs_ply_read(h, data.ctypes.data_as(C.c_void_p)))
        return names, data
    finally:
        lib.lfs_ply_close(h)


def load_ply(path: str, device="cuda:0", active_sh_degree: Optional[int] = None):
    """load_ply (src/loader/formats/ply.cpp:497-640) -> SplatModel, held to the reference's reader run on the CPU (tests/test_loader_reference.py).
    Columns that the file lacks get the reference's defaults (:531-600): sh0 zeros [N,1,3]; shN zeros [N,15,3] (degree 3); opacity 0; log-scale -5 when there
    is no scale_0; the identity quaternion (1,0,0,0) when there is no rot_0. SH degree from the shN width. A LOADED model evaluates every degree it holds
    (active_sh_degree=None: evaluation / rendering of a trained file is what a caller expects; the SplatData the reference constructs starts at 0,
    splat_data.cpp:211, and its viewer / trainer set the degree per request - a resume path that wants that passes active_sh_degree=0)."""
    from .rasterizer import SplatModel
    names, data = read_ply(path)
    col = {n: i for i, n in enumerate(names)}
    if not all(k in col for k in ("x", "y", "z")):
        raise LoaderError("Only binary PLY with position supported")
    N = data.shape[0]
    pick = lambda prefix: [col[f"{prefix}{i}"] for i in range(len(names)) if f"{prefix}{i}" in col]
    dc, rest = pick("f_dc_"), pick("f_rest_")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(device)
    means = t(data[:, [col["x"], col["y"], col["z"]]])
    sh0 = t(data[:, dc]).reshape(N, 3, -1).transpose(1, 2) if dc and len(dc) % 3 == 0 else torch.zeros((N, 1, 3), device=device)
    shN = t(data[:, rest]).reshape(N, 3, -1).transpose(1, 2) if rest and len(rest) % 3 == 0 else torch.zeros((N, 15, 3), device=device)
    opac = t(data[:, col["opacity"]]) if "opacity" in col else torch.zeros(N, device=device)
    sc, ro = pick("scale_"), pick("rot_")
    if "scale_0" in col:
        scales = t(data[:, [col[f"scale_{i}"] for i in range(3)]])
    else:
        scales = torch.full((N, 3), -5.0, device=device)
    if "rot_0" in col:
        quats = t(data[:, [col[f"rot_{i}"] for i in range(4)]])
    else:
        quats = torch.zeros((N, 4), device=device)
        quats[:, 0] = 1.0
    sh_degree = int(np.sqrt(shN.shape[1] + 1)) - 1
    mk = lambda x: x.contiguous().requires_grad_(True)
    return SplatModel(mk(means), mk(sh0), mk(shN), mk(scales), mk(quats), mk(opac), sh_degree, active_sh_degree=active_sh_degree)


# ---------------------------------------------------------------------------------------------------------------------
# dataset / scene
# ---------------------------------------------------------------------------------------------------------------------
@dataclass
class CameraDataset:
    """dataset.hpp:25-75: index i is a validation view iff i % test_every == 0."""
    cameras: List[CameraData]
    split: str = "all"              # "train" | "val" | "all"
    test_every: int = 8
    resize_factor: int = -1
    max_width: int = 0
    indices: List[int] = field(default_factory=list)

    def __post_init__(self):
        self.indices = [i for i in range(len(self.cameras))
                        if self.split == "all" or (self.split == "val") == (i % self.test_every == 0)]

    def __len__(self) -> int:
        return len(self.indices)

    def image_size(self, index: int) -> Tuple[int, int]:
        cam = self.cameras[self.indices[index]]
        w, h = (cam.width, cam.height)
        if os.path.exists(cam.image_path):
            w, h, _ = get_image_info(cam.image_path)
        return image_target_size(w, h, self.resize_factor, self.max_width)

    def get(self, index: int, device="cuda:0") -> Tuple[CameraData, torch.Tensor]:
        if not 0 <= index < len(self.indices):
            raise IndexError("Dataset index out of range")
        cam = self.cameras[self.indices[index]]
        return cam, load_image(cam.image_path, self.resize_factor, self.max_width, device)


def preload(ds: "CameraDataset", device="cuda:0", workers: int = 8) -> List[torch.Tensor]:
    """All images of the split as resident f32 [3,h,w] device tensors. Decoding (liblfs_io, outside the GIL) runs on `workers` host threads;
    upload + resample + layout change stay on the calling thread's stream, in dataset order."""
    import concurrent.futures as cf
    paths = [ds.cameras[i].image_path for i in ds.indices]
    out = []
    with cf.ThreadPoolExecutor(max_workers=max(1, workers)) as ex:
        for rgb in ex.map(decode_rgb8, paths):
            h, w = rgb.shape[:2]
            ow, oh = image_target_size(w, h, ds.resize_factor, ds.max_width)
            host = torch.from_numpy(rgb)
            if torch.cuda.is_available():
                host = host.pin_memory()
            out.append(u8_to_chw_f32(host.to(device, non_blocking=True), ow, oh))
    return out


def world_to_view(cam: CameraData) -> np.ndarray:
    """camera.cpp:15-23: [R | t] with the COLMAP translation as is."""
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] = cam.R
    m[:3, 3] = cam.T
    return m


def intrinsics(cam: CameraData, image_width: int, image_height: int) -> np.ndarray:
    """Camera::K / get_intrinsics (camera.cpp:77-98): the COLMAP intrinsics scaled to the size of the loaded image."""
    sx, sy = np.float32(image_width) / np.float32(cam.width), np.float32(image_height) / np.float32(cam.height)
    K = np.zeros((3, 3), np.float32)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[2, 2] = cam.focal_x * sx, cam.focal_y * sy, cam.center_x * sx, cam.center_y * sy, 1.0
    return K


def colmap_scene(base: str, images_folder: str = "images", split: str = "train", test_every: int = 8, resize_factor: int = -1, max_width: int = 0,
                 sh_degree: int = 3, init_scaling: float = 1.0, init_opacity: float = 0.1, text: bool = False, device="cuda:0"):
    """COLMAP directory -> (Scene for GutTrainer, CameraDataset, scene_scale). All views must share one image size (the trainer's
    view batches are rectangular)."""
    from .scenes import Scene
    cams, center = (read_colmap_cameras_and_images_text if text else read_colmap_cameras_and_images)(base, images_folder)
    pcd = (read_colmap_point_cloud_text if text else read_colmap_point_cloud)(base)
    ds = CameraDataset(cams, split, test_every, resize_factor, max_width)
    if not len(ds):
        raise LoaderError("the requested split has no images")
    w, h = ds.image_size(0)
    model, scene_scale = init_model_from_pointcloud(pcd, center, sh_degree, init_scaling, init_opacity, device)
    used = [cams[i] for i in ds.indices]
    viewmats = torch.from_numpy(np.stack([world_to_view(c) for c in used]))
    Ks = torch.from_numpy(np.stack([intrinsics(c, w, h) for c in used]))
    means, sh0, shN, scales, quats, opac = [p.detach().clone() for p in model.parameters()]
    scene = Scene(os.path.basename(os.path.normpath(base)), w, h, sh_degree, means, quats, scales, opac, sh0, shN, viewmats, Ks,
                  {"scene_scale": scene_scale, "scene_center": center, "active_sh_degree": 0})   # GutTrainer starts the schedule at degree 0
    return scene.to(device), ds, scene_scale
