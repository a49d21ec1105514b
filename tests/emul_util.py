"""The WHOLE product library (every csrc/*.hip except the event profiler) compiled as HOST code on the wavefront emulator (tests/emul/hip/hip_runtime.h)
and put behind the package's ctypes layer, so that the Python mirrors of the reference's operators (ops.py, losses.py, fastgs.py, bilateral_grid.py, ...)
run on CPU tensors through the SAME kernels' source: CPU-side parity of every kernel against the oracle without a GPU (tests/test_emulated_*.py).
Test infrastructure only - the product never imports this, and the emulated library is never installed as a fallback: `installed()` patches the
loader for the duration of a test and puts it back.

What an emulated run checks: the kernels' LOGIC (indexing, cross-lane reductions, LDS hand-overs, atomics, ragged sizes) in host float arithmetic, and -
under LFS_EMUL_SANITIZE=1 (see tests/test_emulated_raster.py) - every global-memory access of every lane against exact-size heap blocks.
What it cannot check: ISA-level behaviour (DPP encodings, hazards, inline asm, v_rcp / v_exp rounding): the `-m gpu` tests.

The sources are compiled UNMODIFIED except for one textual rewrite done here: `extern __shared__ T name[];` (dynamic LDS) becomes a pointer to the emulator's
dynamic-LDS block (the files that declare it through LFS_DYN_LDS need no rewrite)."""
from __future__ import annotations

import contextlib
import ctypes as C
import hashlib
import importlib.util
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "lichtfeld-studio_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
_DYN_LDS = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?(\w+)\s+(\w+)\[\];")
_SKIP = {"prof.hip"}   # the HIP-event profiler: tests/emul/emul_stubs.cpp stands in for it
_LIB = None


def _product_sources() -> dict:
    spec = importlib.util.spec_from_file_location("lfs_build_for_emul", os.path.join(ROOT, "lichtfeld-studio_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return {k: [f for f in v if f.startswith("-ffp-contract") or f.startswith("-D")] for k, v in mod.SOURCES.items() if k not in _SKIP}


def available() -> bool:
    return os.path.exists(CLANG)


def build() -> str:
    """-> path of the emulated library (built once per content hash of sources + emulator under the system temp directory)"""
    srcs = _product_sources()
    sanitize = bool(os.environ.get("LFS_EMUL_SANITIZE"))
    extra_defines = os.environ.get("LFS_EMUL_DEFINES", "").split()   # e.g. "-DLFS_BWD_REORTH=1": an emulated build of a compile-time variant of the product library
    h = hashlib.sha1(repr(sorted(srcs.items())).encode() + (b"asan" if sanitize else b"") + " ".join(extra_defines).encode())
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(HERE, "emul", "hip", "hip_runtime.h"), os.path.join(HERE, "emul", "emul_stubs.cpp"),
                                                                       os.path.join(ROOT, "include", "lfs_gsplat.h"), __file__]
    for d in deps:
        if os.path.isfile(d):
            h.update(open(d, "rb").read())
    work = os.path.join(tempfile.gettempdir(), f"lfs_emul_{os.getuid()}_{h.hexdigest()[:16]}")
    out = os.path.join(work, "liblfs_gsplat_emul.so")
    if os.path.exists(out):
        return out
    os.makedirs(work, exist_ok=True)
    common = [CLANG, "-x", "c++", "-std=c++17", "-O1", "-DLFS_EMULATE", "-fPIC", "-I" + os.path.join(HERE, "emul"), "-I" + CSRC, "-Wno-unused-value", "-Wno-unknown-attributes"]
    common += extra_defines
    if sanitize:
        common[1:1] = ["-fsanitize=address", "-shared-libasan", "-fno-omit-frame-pointer", "-g"]

    def compile_one(item):
        name, flags = item
        text = open(os.path.join(CSRC, name)).read()
        text = _DYN_LDS.sub(lambda m: f"{m.group(1)}* {m.group(2)} = reinterpret_cast<{m.group(1)}*>(emu::dyn_lds());", text)
        pp = os.path.join(work, name)
        with open(pp, "w") as fh:
            fh.write(f'#line 1 "{os.path.join(CSRC, name)}"\n' + text)
        obj = pp + ".o"
        if not any(f.startswith("-ffp-contract") for f in flags):
            flags = [*flags, "-ffp-contract=on"]
        r = subprocess.run([*common, *flags, "-c", pp, "-o", obj], capture_output=True, text=True)
        assert r.returncode == 0, f"{name}:\n{r.stderr[-3000:]}"
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, srcs.items()))
    stubs = os.path.join(work, "emul_stubs.o")
    r = subprocess.run([*common, "-c", os.path.join(HERE, "emul", "emul_stubs.cpp"), "-o", stubs], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    tmp = out + f".{os.getpid()}.tmp"
    link = [CLANG, "-shared", "-fPIC", *objs, stubs, "-o", tmp]
    if sanitize:
        link[1:1] = ["-fsanitize=address", "-shared-libasan"]
    r = subprocess.run(link, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    os.replace(tmp, out)
    return out


class _EmulatedLib:
    """ctypes library + one adjustment: debug bit 4 (the DETERMINISTIC accumulation mode: integer atomics, compiled out of the emulated build - there is nothing to
    make deterministic, the emulator executes the wavefronts one after the other) is dropped on its way in, so a test that asks for that mode gets the plain
    float accumulation - reproducible here - instead of kernels that accumulate nothing."""

    def __init__(self, lib):
        self.__dict__["_lib"] = lib

    def __getattr__(self, name):
        return getattr(self._lib, name)

    def lfs_set_debug_flags(self, flags):
        v = int(getattr(flags, "value", flags))
        return self._lib.lfs_set_debug_flags(C.c_uint32(v & ~16))


def library():
    """the emulated library behind ctypes, with the restypes capi.load_library sets"""
    global _LIB
    if _LIB is None:
        lib = C.CDLL(build())
        for name in ("lfs_intersect_tile_workspace_bytes", "lfs_rasterize_workspace_bytes", "lfs_photometric_loss_workspace_bytes", "lfs_fastgs_primitive_workspace_bytes",
                     "lfs_fastgs_instance_workspace_bytes", "lfs_mcmc_relocate_workspace_bytes", "lfs_rasterize_workspace_acc_offset"):
            getattr(lib, name).restype = C.c_size_t
        lib.lfs_version.restype = C.c_char_p
        _LIB = _EmulatedLib(lib)
    return _LIB


@contextlib.contextmanager
def installed():
    """Every module of the package talks to the emulated library and accepts CPU tensors; undone on exit."""
    sys.path.insert(0, ROOT)
    import lichtfeld_studio_amd as lfs   # noqa: F401
    lib = library()
    pkg = sys.modules["lichtfeld_studio_amd"].__name__
    import importlib
    for m in ("capi", "ops", "losses", "fastgs", "bilateral_grid", "fused", "fused_adam", "rasterizer", "strategies", "gut_step", "trainer"):
        importlib.import_module(f"{pkg}.{m}")
    mods = [m for name, m in list(sys.modules.items()) if m is not None and (name == pkg or name.startswith(pkg + "."))]   # every module of the package that holds a loader reference
    repl = {
        "load_library": lambda: lib,
        "require_gpu": _require_cpu_contiguous,
        "stream": lambda: None,
        "workspace": lambda nbytes, dev, tag: torch.zeros(max(int(nbytes), 256), dtype=torch.uint8),
    }
    saved = []
    for m in mods:
        for k, v in repl.items():
            if hasattr(m, k):
                saved.append((m, k, getattr(m, k)))
                setattr(m, k, v)
    try:
        yield lib
    finally:
        for m, k, v in saved:
            setattr(m, k, v)


def _require_cpu_contiguous(*tensors):
    for t in tensors:
        if t is not None and not t.is_contiguous():
            from lichtfeld_studio_amd.capi import LfsError
            raise LfsError("tensor must be contiguous")


# ---- running `-m gpu` test bodies on the emulated library -------------------------------------------------------------------------------
# The GPU parity tests name their device ("cuda:0") in literals; this mode maps every such request to the CPU while it is active, so the SAME test
# functions - same inputs, same oracle calls, same assertions - can be executed against the emulated kernels (tests/test_emulated_gpu_suite.py).
def _is_cuda(d) -> bool:
    if isinstance(d, torch.device):
        return d.type == "cuda"
    return isinstance(d, str) and d.startswith("cuda")


class _CudaToCpu(torch.overrides.TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if _is_cuda(kwargs.get("device")):
            kwargs["device"] = "cpu"
        if func is torch.Tensor.pin_memory:
            return args[0]
        moved = func is torch.Tensor.cuda
        if any(_is_cuda(a) for a in args):
            args = tuple("cpu" if _is_cuda(a) else a for a in args)
            moved = moved or func is torch.Tensor.to
        if func is torch.Tensor.cuda:
            return args[0].clone()   # a transfer makes a NEW tensor (x = a.to(DEV).requires_grad_() must not turn `a` into a leaf that requires grad)
        out = func(*args, **kwargs)
        return out.clone() if moved and out is args[0] else out


@contextlib.contextmanager
def cuda_requests_served_by_the_cpu():
    real_generator, real_sync = torch.Generator, torch.cuda.synchronize

    class generator(real_generator):   # (a class, not a function: `torch.Generator | None` annotations are evaluated while it is in place)
        def __new__(cls, device="cpu"):
            return real_generator.__new__(real_generator, "cpu" if _is_cuda(device) else device)
    class stream:   # what torch.cuda.current_stream() hands out: the emulated kernels have finished when their launch returns
        cuda_stream = 0
        def synchronize(self): pass
        def wait_stream(self, other): pass
        def wait_event(self, event): pass
        def record_event(self, event=None): return event
    class event:    # torch.cuda.Event
        def __init__(self, *a, **k): pass
        def record(self, stream=None): pass
        def synchronize(self): pass
        def wait(self, stream=None): pass
        def query(self): return True
        def elapsed_time(self, other): return 0.0
    saved = {k: getattr(torch.cuda, k) for k in ("current_stream", "current_device", "set_device", "Event")}
    torch.cuda.Event = event
    torch.Generator, torch.cuda.synchronize = generator, (lambda *a, **k: None)
    torch.cuda.current_stream, torch.cuda.current_device, torch.cuda.set_device = (lambda *a, **k: stream()), (lambda: 0), (lambda *a, **k: None)
    torch.Tensor.is_cuda = property(lambda self: True)   # the host layer's "must be a device tensor" checks (torch.Tensor is a Python class: removed again below)
    try:
        with _CudaToCpu():
            yield
    finally:
        del torch.Tensor.is_cuda
        torch.Generator, torch.cuda.synchronize = real_generator, real_sync
        for k, v in saved.items():
            setattr(torch.cuda, k, v)
