#!/usr/bin/env python3
"""sha1 over the colours of spherical_harmonics_fwd (operator form, with and without masks) and of the model form (sh_model_fwd) on a fixed set of random problems covering every
lane layout (K = 1, 4, 9, 16, 25): two builds of csrc/sh.hip that claim the same summation order must print the same digest.
    python tools/sh_fwd_bits.py   (GPU, the library LFS_GSPLAT_LIB selects)      python tools/sh_fwd_bits.py --emulated   (CPU; LFS_EMUL_DEFINES selects the build)
Used in round 5 to show that the LDS-transposed forward (same pairing tree as the DPP butterfly) is bit-identical on the emulator and on the MI355X - and no faster: removed."""
import contextlib
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    emulated = "--emulated" in sys.argv
    ctx = contextlib.nullcontext()
    if emulated:
        import emul_util
        ctx = emul_util.installed()
    dev = "cpu" if emulated else "cuda:0"
    rng = np.random.default_rng(11)
    h = hashlib.sha1()
    with ctx:
        import lichtfeld_studio_amd as lfs
        from lichtfeld_studio_amd import fused, ops
        t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
        for n, K, deg in ((100000, 16, 3), (777, 16, 2), (130, 9, 2), (65, 4, 1), (64, 1, 0), (3000, 25, 4), (191, 16, 0)):
            if emulated:
                n = min(n, 1000)
            dirs = rng.standard_normal((n, 3)).astype(np.float32)
            coeffs = rng.standard_normal((n, K, 3)).astype(np.float32)
            mask = rng.random(n) < 0.8
            for m in (None, mask):
                c = ops.spherical_harmonics_fwd(deg, t(dirs), t(coeffs), None if m is None else t(m, torch.bool)).cpu().numpy()
                if m is not None:
                    c = np.where(m[:, None], c, 0)
                h.update(c.tobytes())
            means = rng.standard_normal((n, 3)).astype(np.float32) * 3
            vm = np.eye(4, dtype=np.float32); vm[:3, 3] = [0.3, -0.2, 5.0]
            radii = rng.integers(0, 3, (1, n, 2)).astype(np.int32)
            if K > 1:
                col = fused.sh_model_fwd(deg, t(means), t(vm[None]), t(coeffs[:, :1]), t(coeffs[:, 1:]), t(radii, torch.int32)).cpu().numpy()
                h.update(np.where((radii[0] > 0).all(-1)[:, None], col, 0).tobytes())
        print("library:", "emulated" if emulated else lfs.load_library().lfs_version().decode(), "| sh_fwd digest", h.hexdigest())


if __name__ == "__main__":
    main()
