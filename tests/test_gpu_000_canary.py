"""First-collected GPU test (the file name sorts in front of every other test_gpu_*): a red GPU run has to explain itself.

1. `test_runtime_report` prints - outside pytest's capture, so it lands in the driver's log even when everything passes - which HIP / HSA
   runtime libraries serve this process (/proc/self/maps), their versions, the device, the kernel driver and the environment switches that
   change how kernels are loaded. liblfs_gsplat.so is linked against libamdhip64.so.7; torch bundles a runtime with the same SONAME, so the
   load order decides which one it binds to - the report shows the answer instead of leaving it to be guessed from a crash.
2. `test_trivial_kernel_in_subprocess` launches ONE trivial lfs_ kernel (quats_to_rotmats on 64 quaternions) in a fresh python process
   and checks the numbers. A GPU memory fault aborts the process that caused it; in a child the parent survives, re-runs the child once with
   AMD_SERIALIZE_KERNEL=3 / HIP_LAUNCH_BLOCKING=1 / AMD_LOG_LEVEL=3 and reports the tail, then stops the session with that message: every
   later test would die the same way with nothing but "Aborted" in the log (GPUTEST_r02).
3. `test_trivial_kernel_in_process` is the same launch inside the pytest process (this is the process whose library loads the driver records).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, {root!r})
import numpy as np, torch
import lichtfeld_studio_amd as lfs
from lichtfeld_studio_amd import ops
g = np.random.default_rng(1)
q = g.standard_normal((64, 4)).astype(np.float32)
R = ops.quats_to_rotmats(torch.from_numpy(q).to("cuda:0")).cpu().numpy()
w, x, y, z = (q / np.linalg.norm(q, axis=1, keepdims=True)).T
ref = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1).reshape(64, 3, 3)
err = float(np.abs(R - ref).max())
assert err < 1e-5, err
print("CANARY_OK", err)
"""


def _maps():
    libs = set()
    with open("/proc/self/maps") as f:
        for line in f:
            p = line.rsplit(" ", 1)[-1].strip()
            if p.endswith((".so",)) or ".so." in p:
                libs.add(p)
    return sorted(libs)


def _report() -> str:
    import torch
    out = [f"torch {torch.__version__} hip {torch.version.hip} | python {sys.version.split()[0]} | pid {os.getpid()}"]
    if torch.cuda.is_available():
        p = torch.cuda.get_device_properties(0)
        out.append(f"device 0: {p.name} gcn {getattr(p, 'gcnArchName', '?')} CUs {p.multi_processor_count} mem {p.total_memory >> 30} GiB | device_count {torch.cuda.device_count()}")
    for f in ("/sys/module/amdgpu/version", "/sys/module/amdgpu/srcversion"):
        if os.path.exists(f):
            out.append(f"{f}: {open(f).read().strip()}")
    try:
        import lichtfeld_studio_amd as lfs
        out.append("liblfs_gsplat: " + lfs.load_library().lfs_version().decode() + " @ " + lfs.library_path())
    except Exception as e:  # noqa: BLE001 - the report must not hide the real failure behind its own
        out.append(f"liblfs_gsplat: NOT LOADED ({e})")
    keep = ("amdhip", "hsa-runtime", "amd_comgr", "liblfs", "rocprofiler", "roctracer", "librccl", "/oracle/")
    out += ["mapped: " + lib for lib in _maps() if any(k in lib for k in keep)]
    env = {k: v for k, v in os.environ.items() if k.startswith(("HSA_", "HIP_", "AMD_", "ROCR_", "ROCM_", "GPU_", "LD_", "PYTORCH_", "CUDA_VISIBLE", "LFS_"))}
    out.append("env: " + (" ".join(f"{k}={v}" for k, v in sorted(env.items())) or "(none of HSA_/HIP_/AMD_/ROCR_/LD_/LFS_ set)"))
    return "\n".join("[canary] " + line for line in out)


def _run_child(extra_env=None, timeout=600):
    env = dict(os.environ)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], capture_output=True, text=True, timeout=timeout, env=env)


def test_runtime_report(lfs, capsys):
    with capsys.disabled():
        print("\n" + _report(), flush=True)
    # collection must not have loaded a checker library into this process (tests/test_collection_loads_nothing.py holds the same on CPU)
    assert not [m for m in _maps() if "/oracle/_ref/" in m], "a reference checker library is mapped before any test asked for it"


def test_trivial_kernel_in_subprocess(capsys):
    r = _run_child()
    if r.returncode == 0 and "CANARY_OK" in r.stdout:
        return
    first = (r.stdout[-1500:] + "\n" + r.stderr[-3000:]).strip()
    dbg = _run_child({"AMD_SERIALIZE_KERNEL": "3", "HIP_LAUNCH_BLOCKING": "1", "AMD_LOG_LEVEL": "3"})
    msg = (f"GPU canary failed: a fresh process cannot run one trivial lfs_ kernel on this box (rc {r.returncode}).\n--- first attempt ---\n{first}\n"
           f"--- second attempt, serialized + AMD_LOG_LEVEL=3: rc {dbg.returncode} ---\n{dbg.stdout[-1500:]}\n{dbg.stderr[-6000:]}\n{_report()}")
    with capsys.disabled():
        print("\n" + msg, flush=True)
    if dbg.returncode == 0 and "CANARY_OK" in dbg.stdout:
        pytest.fail("the canary kernel failed once and passed when re-run serialized: a launch-order / first-touch race, see the log above")
    pytest.exit(msg, returncode=3)


def test_trivial_kernel_in_process(lfs):
    import torch
    from lichtfeld_studio_amd import ops
    q = np.random.default_rng(2).standard_normal((64, 4)).astype(np.float32)
    R = ops.quats_to_rotmats(torch.from_numpy(q).to("cuda:0"))
    torch.cuda.synchronize()
    R = R.cpu().numpy()
    det = np.linalg.det(R.astype(np.float64))
    assert np.isfinite(R).all() and np.abs(det - 1).max() < 1e-4, det
