"""Build the gfx950 HIP library (C ABI, include/lfs_gsplat.h) in-tree.

    python lichtfeld-studio_amd/build.py            # -> lichtfeld-studio_amd/liblfs_gsplat.so

hipcc cross-compiles without a GPU. Objects are cached by source mtime under
lichtfeld-studio_amd/build/. No CUDA path, no hipify, one target: gfx950.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "liblfs_gsplat.so")
BUILD = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-munsafe-fp-atomics",
          "-Wall", "-Wno-unused-function"]
# Streaming kernels keep IEEE un-fused arithmetic (bit-level agreement with the oracle costs nothing
# when HBM-bound); the VALU-bound rasterizer uses fused multiply-adds.
SOURCES = {
    "projection_ut.hip": ["-ffp-contract=off"],
    # LFS_SH_DPP_SUM: the 16-lane sums as DPP row operations instead of ds_bpermute - bit-identical sums (tests/test_emulated_sh.py),
    # GPU-verified in round 2 (tests/test_gpu_projection_sh.py + test_gpu_fused.py green on the variant build; sh_fwd 0.0755 -> 0.0704 ms)
    # -fno-slp-vectorize (round 3): the SLP vectorizer pairs the three colour channels' group sums into v_pk_add_f32, which keeps the DPP moves from folding into the
    # adds and, in the operator form of the forward kernel (sh_fwd_kernel<16, false>), blew the allocation up to 508 VGPRs + AGPR spills: 0.23 ms per launch at 1M
    # Gaussians against 0.04 for the model form (tools/bench_sh_ops.py). Without it: 71 VGPRs; every other kernel of the file needs fewer registers as well.
    "sh.hip": ["-ffp-contract=off", "-DLFS_SH_DPP_SUM", "-fno-slp-vectorize"],
    "intersect.hip": ["-ffp-contract=off"],
    "mcmc.hip": ["-ffp-contract=off"],
    "adam.hip": ["-ffp-contract=off"],
    "l2_fused.hip": ["-ffp-contract=off"],
    "ssim.hip": ["-ffp-contract=off"],
    "bilateral_grid.hip": ["-ffp-contract=off"],
    "dataprep.hip": ["-ffp-contract=off"],
    "fastgs_prep.hip": ["-ffp-contract=off"],
    "fastgs_blend.hip": ["-fno-slp-vectorize"],
    "prof.hip": [],
    # no SLP packing: v_pk_* operand pairing forces SGPR shuffles right after the scalar record
    # load and defeats the software prefetch (measured on the ISA); plain v_fma with SGPR operands
    "raster.hip": ["-fno-slp-vectorize", "-DLFS_SH_DPP_SUM"],   # (LFS_SH_DPP_SUM: the SH lane-group sums of gut_tail_kernel as in sh.hip)
    "gut_step.hip": [],   # host code only: the C++ training-step driver
    "version.hip": [],    # lfs_version(): carries the hash of the sources (recompiled whenever any of them changed)
}
HEADERS = ["lfs_math.cuh", "lfs_sh.cuh", "lfs_adam.cuh", "lfs_camera.cuh", "lfs_prof.h", "lfs_raster_common.cuh", "lfs_cull_conic.cuh", "lfs_raster_pack.cuh", "lfs_tilelists.cuh", "lfs_fastgs.cuh", "lfs_step_internal.h", os.path.join("..", "..", "include", "lfs_gsplat.h")]


def source_hash() -> str:
    """sha1 (12 hex digits) over every file of csrc/ (version.hip excluded: it only carries the hash) and include/lfs_gsplat.h, in name order."""
    import hashlib
    h = hashlib.sha1()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cuh", ".h", ".cpp")) and f != "version.hip")
    for f in files + [os.path.join("..", "..", "include", "lfs_gsplat.h")]:
        h.update(os.path.basename(f).encode())
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def _stale(obj: str, src: str) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [src, __file__] + [os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(name: str, extra: list[str]) -> str:
    src = os.path.join(CSRC, name)
    obj = os.path.join(BUILD, name + ".o")
    stale = _stale(obj, src)
    if name == "version.hip":
        h = source_hash()
        stamp = os.path.join(BUILD, "version.hash")
        stale = stale or not os.path.exists(stamp) or open(stamp).read() != h
        extra = [*extra, f'-DLFS_SRC_HASH="{h}"']
    if stale:
        cmd = [HIPCC, *COMMON, *extra, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {name}:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
        if name == "version.hip":
            with open(os.path.join(BUILD, "version.hash"), "w") as fh:
                fh.write(source_hash())
    return obj


def build(force: bool = False) -> str:
    os.makedirs(BUILD, exist_ok=True)
    if force:
        for f in os.listdir(BUILD):
            os.remove(os.path.join(BUILD, f))
    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(lambda kv: _compile(*kv), SOURCES.items()))
    if force or not os.path.exists(OUT) or any(os.path.getmtime(o) > os.path.getmtime(OUT) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", OUT]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return OUT


VARIANT_DS_WRITE2 = os.path.join(HERE, "liblfs_gsplat_red_ds_write2.so")


def build_variants(force: bool = False) -> str:
    """The test-suite's second build of the two rasterizer files: -DLFS_RED_ADDTID=0 - the backward's 16-value wave reduction through plain ds_write2_b32 stores
    instead of the inline-asm ds_write_addtid_b32 block (lfs_raster_common.cuh) -> liblfs_gsplat_red_ds_write2.so. tests/test_gpu_raster.py runs both
    libraries on the same inputs: the asm path is checked against compiler-generated code on every suite run, not only when somebody remembers to."""
    build()
    vsrc = ["raster.hip", "fastgs_blend.hip"]
    vobjs = []
    for name in vsrc:
        src, obj = os.path.join(CSRC, name), os.path.join(BUILD, name + ".red_ds_write2.o")
        if force or _stale(obj, src):
            cmd = [HIPCC, *COMMON, *SOURCES[name], "-DLFS_RED_ADDTID=0", "-c", src, "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed for the variant of {name}:\n{r.stdout}\n{r.stderr}")
        vobjs.append(obj)
    objs = [os.path.join(BUILD, n + ".o") for n in SOURCES if n not in vsrc] + vobjs
    if force or not os.path.exists(VARIANT_DS_WRITE2) or any(os.path.getmtime(o) > os.path.getmtime(VARIANT_DS_WRITE2) for o in objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", VARIANT_DS_WRITE2], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"variant link failed:\n{r.stdout}\n{r.stderr}")
    return VARIANT_DS_WRITE2


IO_OUT = os.path.join(HERE, "liblfs_io.so")


def build_io(force: bool = False) -> str:
    """Host-only data-format library (csrc_host/lfs_io.cpp: COLMAP, splat PLY, PNG/PNM) -> lichtfeld-studio_amd/liblfs_io.so (g++ + zlib)."""
    srcs = [os.path.join(HERE, "csrc_host", "lfs_io.cpp"), os.path.join(HERE, "csrc_host", "lfs_jpeg.cpp")]
    deps = srcs + [os.path.join(HERE, "..", "include", "lfs_io.h")]
    if not force and os.path.exists(IO_OUT) and all(os.path.getmtime(d) <= os.path.getmtime(IO_OUT) for d in deps):
        return IO_OUT
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", "-ffp-contract=off", "-Wall", "-Wextra", *srcs, "-lz", "-o", IO_OUT]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"liblfs_io build failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    return IO_OUT


TORCH_BACKEND_OUT = os.path.join(HERE, "liblfs_gsplat_torch.so")
TORCH_OPS_OUT = os.path.join(HERE, "_lfs_torch_ops.so")


def _torch_flags():
    import torch
    tdir = os.path.dirname(torch.__file__)
    cflags = ["-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-D_GLIBCXX_USE_CXX11_ABI=1",
              f"-I{tdir}/include", f"-I{tdir}/include/torch/csrc/api/include", "-I/opt/rocm/include"]
    ldflags = [f"-L{tdir}/lib", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", f"-L{HERE}", "-llfs_gsplat", "-llfs_io", "-L/opt/rocm/lib", "-lamdhip64",
               f"-Wl,-rpath,{tdir}/lib", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
    return cflags, ldflags


def _fresh(out: str, deps: list[str]) -> bool:
    return os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps)


def build_torch_backend(force: bool = False) -> str:
    """The drop-in backend library: csrc/torch_ops.cpp (the reference's gsplat:: / fast_gs:: / fusedssim / gs::bilateral_grid C++ signatures on libtorch tensors, over
    the C ABI of liblfs_gsplat.so) -> lichtfeld-studio_amd/liblfs_gsplat_torch.so. This is what replaces the reference's `gsplat_backend` + `fastgs_backend` static
    libraries (gsplat/CMakeLists.txt:42, fastgs/CMakeLists.txt:26) at link time; oracle/Makefile `reflink` links the reference's own rasterizer.cpp /
    rasterizer_autograd.cpp / fused_adam.cpp against it (tests/test_gpu_reference_links.py)."""
    build()
    build_io()
    src = os.path.join(CSRC, "torch_ops.cpp")
    deps = [src, os.path.join(HERE, "..", "include", "lfs_gsplat_torch.hpp"), os.path.join(HERE, "..", "include", "lfs_gut_train_step.hpp"), os.path.join(HERE, "..", "include", "lfs_gsplat.h"),
            os.path.join(HERE, "..", "include", "lfs_io.h"), OUT, IO_OUT]
    if not force and _fresh(TORCH_BACKEND_OUT, deps):
        return TORCH_BACKEND_OUT
    cflags, ldflags = _torch_flags()
    cmd = ["g++", *cflags, src, *ldflags, "-o", TORCH_BACKEND_OUT]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"torch backend build failed:\n{' '.join(cmd)}\n{r.stdout[-3000:]}\n{r.stderr[-6000:]}")
    return TORCH_BACKEND_OUT


def build_torch_ops(force: bool = False) -> str:
    """pybind module over the backend library for the tests (csrc/torch_ops_pybind.cpp) -> lichtfeld-studio_amd/_lfs_torch_ops.so, linked against
    liblfs_gsplat_torch.so (the wrappers are compiled once, into the library a reference build links)."""
    import sysconfig

    import pybind11
    backend = build_torch_backend(force)
    src = os.path.join(CSRC, "torch_ops_pybind.cpp")
    deps = [src, os.path.join(HERE, "..", "include", "lfs_gsplat_torch.hpp"), backend]
    if not force and _fresh(TORCH_OPS_OUT, deps):
        return TORCH_OPS_OUT
    cflags, ldflags = _torch_flags()
    cmd = ["g++", *cflags, "-DTORCH_EXTENSION_NAME=_lfs_torch_ops", "-DTORCH_API_INCLUDE_EXTENSION_H", f"-I{pybind11.get_include()}",
           f"-I{sysconfig.get_paths()['include']}", src, "-llfs_gsplat_torch", *ldflags, "-ltorch_python", "-o", TORCH_OPS_OUT]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"torch ops build failed:\n{' '.join(cmd)}\n{r.stdout[-3000:]}\n{r.stderr[-6000:]}")
    return TORCH_OPS_OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build_io(force="--force" in sys.argv))
    if "--torch-ops" in sys.argv:
        print(build_torch_ops(force="--force" in sys.argv))
