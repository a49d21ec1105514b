#!/usr/bin/env python3
"""(Superseded reading: profiles/r06/tail_kernel_levels.txt item 4 - the level follows the clock, not the allocation.) Is the spread of the HBM-bound tail kernel (0.285 - 0.338 ms between runs of bench.py on one box, profiles/r06/lease36c_...) a property of the PROCESS - where its
allocations landed - or of the moment? One process, the headline step of SYN-B, W windows of S steps each with the tail kernel and raster_bwd timed (events), then the
whole thing again in a fresh trainer of the same process (new allocations). GPU.   python tools/tail_variance.py [windows] [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    import lichtfeld_studio_amd as lfs  # noqa: F401
    from lichtfeld_studio_amd import capi, scenes
    from lichtfeld_studio_amd.trainer import GutTrainer
    dev = torch.device("cuda:0")
    scene = scenes.syn_b()
    target = [scenes.target_image(scene.height, scene.width, seed=43).to(dev)]
    for trainer_no in range(int(os.environ.get("LFS_TAILVAR_TRAINERS", "3"))):
        tr = GutTrainer(scene, dev, iterations=7000, world=1, rank=0, views_per_rank=1)
        tr.iteration = 3000
        for _ in range(10):
            tr.train_step(target)
        torch.cuda.synchronize()
        for w in range(W):
            capi.profile_collect(); capi.profile_filter(None); capi.profile_enable(True)
            t0 = time.perf_counter()
            for _ in range(S):
                tr.train_step(target)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / S * 1e3
            capi.profile_enable(False)
            t = capi.profile_collect()
            row = {k: round(v[0] / max(v[1], 1), 4) for k, v in t.items() if k in ("tail_sh_finish_adam", "raster_bwd", "raster_fwd", "activations_projection_ut")}
            print(f"trainer {trainer_no} window {w}: {dt:.4f} ms/step (all scopes timed) {row}", flush=True)
        del tr
        torch.cuda.empty_cache()
        if trainer_no == 0:
            time.sleep(3.0)   # (an idle pause between the first two trainers: does the state carry over it?)


if __name__ == "__main__":
    main()
