#!/usr/bin/env python3
"""Does the HBM-bound tail kernel's level (0.29 / 0.32 - 0.34 ms, profiles/r06/tail_kernel_levels.txt) follow the device's own load history? One process, one trainer: the training
step run continuously for `busy` seconds (tail kernel averaged over 2-second slices), an idle pause of `idle` seconds, and again - three cycles. GPU.
    python tools/tail_thermal.py [busy_s] [idle_s]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    busy = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
    idle = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
    import lichtfeld_studio_amd as lfs  # noqa: F401
    from lichtfeld_studio_amd import capi, scenes
    from lichtfeld_studio_amd.trainer import GutTrainer
    dev = torch.device("cuda:0")
    scene = scenes.syn_b()
    target = [scenes.target_image(scene.height, scene.width, seed=43).to(dev)]
    tr = GutTrainer(scene, dev, iterations=7000, world=1, rank=0, views_per_rank=1)
    start = time.time()
    for cycle in range(3):
        t_end = time.time() + busy
        while time.time() < t_end:
            tr.iteration = 3000
            capi.profile_collect(); capi.profile_filter("tail_sh_finish_adam"); capi.profile_enable(True)
            t0 = time.time()
            n = 0
            while time.time() - t0 < 2.0:
                for _ in range(50):
                    tr.train_step(target)
                n += 50
                torch.cuda.synchronize()
            capi.profile_enable(False)
            t = capi.profile_collect()
            v = t.get("tail_sh_finish_adam", (0.0, 1))
            print(f"t = {time.time() - start:6.1f} s  cycle {cycle}  tail {v[0] / max(v[1], 1):.4f} ms  ({n} steps, {(time.time() - t0) / n * 1e3:.3f} ms/step)", flush=True)
        print(f"t = {time.time() - start:6.1f} s  idle for {idle} s", flush=True)
        time.sleep(idle)


if __name__ == "__main__":
    main()
