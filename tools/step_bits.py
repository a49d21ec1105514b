#!/usr/bin/env python3
"""sha1 over what N deterministic training steps leave behind (LFS_DEBUG_FLAGS=16: fixed-point accumulators, so two runs of ONE library print the same digest): the loss of
every step and all six parameter tensors, on SYN-B cut to 200 000 Gaussians (flat disks: the backward's hard case). Two builds that claim the same arithmetic - a record layout,
an instruction selection, a scalar-unit change - must print the same line:
    LFS_DEBUG_FLAGS=16 python tools/step_bits.py            LFS_GSPLAT_LIB=.../liblfs_gsplat_<variant>.so LFS_DEBUG_FLAGS=16 python tools/step_bits.py"""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    assert int(os.environ.get("LFS_DEBUG_FLAGS", "0"), 0) & 16, "run with LFS_DEBUG_FLAGS=16 (deterministic accumulation)"
    import lichtfeld_studio_amd as lfs
    from lichtfeld_studio_amd import scenes
    from lichtfeld_studio_amd.trainer import GutTrainer
    dev = torch.device("cuda:0")
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    h = hashlib.sha1()
    for name, sc in (("syn_b_flat", scenes.syn_b_flat(n=200_000, n_views=4)), ("syn_b", scenes.syn_b(n=200_000, n_views=4))):
        tr = GutTrainer(sc, dev, iterations=7000)
        tr.iteration = 3000
        targets = [scenes.target_image(sc.height, sc.width, seed=43 + v).to(dev) for v in range(4)]
        for s in range(steps):
            loss = tr.train_step([targets[s % 4]])
            h.update(loss.detach().cpu().numpy().tobytes())
        torch.cuda.synchronize()
        for p in tr.model.parameters():
            h.update(p.detach().cpu().numpy().tobytes())
        print(name, "loss", float(loss), "digest so far", h.hexdigest()[:16])
    print(lfs.load_library().lfs_version().decode(), os.environ.get("LFS_GSPLAT_LIB", "default"), "steps", steps, "sha1", h.hexdigest())


if __name__ == "__main__":
    main()
