#!/usr/bin/env python3
"""What decides whether the HBM-bound tail kernel runs at 0.289 or at 0.325 ms (tools/tail_variance.py: constant inside a trainer, different between trainers)? The 18 arrays it
streams (six parameter tensors, their two Adam moments each) are moved into ONE arena at chosen offsets - `skew` bytes times the array's index on top of a 2-MiB-aligned slot -
and the step is timed for each skew, in one process; the addresses of the default placement are printed first. GPU.   python tools/tail_placement.py [skew ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(tr, target, capi, S=100):
    for _ in range(5):
        tr.train_step(target)
    torch.cuda.synchronize()
    capi.profile_collect(); capi.profile_filter(None); capi.profile_enable(True)
    tr.iteration = 3000
    t0 = time.perf_counter()
    for _ in range(S):
        tr.train_step(target)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / S * 1e3
    capi.profile_enable(False)
    t = capi.profile_collect()
    return dt, {k: round(v[0] / max(v[1], 1), 4) for k, v in t.items() if k in ("tail_sh_finish_adam", "raster_bwd", "activations_projection_ut")}


def arrays(tr):
    out = []
    for p in tr.model.parameters():
        st = tr.optimizer.state.get(id(p))
        out.append(("p", p)); 
        if st is not None:
            out.append(("m", st["exp_avg"])); out.append(("v", st["exp_avg_sq"]))
    return out


def main():
    skews = [int(x) for x in sys.argv[1:]] or [0, 256, 4096, 65536, 4096 + 256, 131072 + 4096 + 256]
    import lichtfeld_studio_amd as lfs  # noqa: F401
    from lichtfeld_studio_amd import capi, scenes
    from lichtfeld_studio_amd.trainer import GutTrainer
    dev = torch.device("cuda:0")
    scene = scenes.syn_b()
    target = [scenes.target_image(scene.height, scene.width, seed=43).to(dev)]
    tr = GutTrainer(scene, dev, iterations=7000, world=1, rank=0, views_per_rank=1)
    tr.iteration = 3000
    tr.train_step(target); torch.cuda.synchronize()   # (creates the Adam state)
    arr = arrays(tr)
    print("default placement:", [(k, tuple(t.shape), hex(t.data_ptr()), t.data_ptr() % (2 << 20)) for k, t in arr], flush=True)
    print("default:", timed(tr, target, capi), flush=True)
    SLOT = 2 << 20
    sizes = [((t.numel() * 4 + SLOT - 1) // SLOT + 1) * SLOT for _, t in arr]
    init = [t.detach().clone() for _, t in arr]
    for skew in skews:
        arena = torch.empty(sum(sizes) + SLOT + len(arr) * max(skew, 1) + (1 << 20), dtype=torch.uint8, device=dev)
        base = (arena.data_ptr() + SLOT - 1) // SLOT * SLOT - arena.data_ptr()
        off = base
        views = []
        for i, ((k, t), sz, src) in enumerate(zip(arr, sizes, init)):
            o = off + i * skew
            v = arena[o:o + t.numel() * 4].view(torch.float32).view(t.shape)
            v.copy_(src)
            views.append(v)
            off += sz
        j = 0
        for p in tr.model.parameters():
            st = tr.optimizer.state.get(id(p))
            p.data = views[j]; j += 1
            if st is not None:
                st["exp_avg"] = views[j]; st["exp_avg_sq"] = views[j + 1]; j += 2
        tr.iteration = 3000
        print(f"skew {skew:7d}:", timed(tr, target, capi), [v.data_ptr() % SLOT for v in views][:6], flush=True)
        for p in tr.model.parameters():   # release the arena before the next one
            pass
    print("done")


if __name__ == "__main__":
    main()
