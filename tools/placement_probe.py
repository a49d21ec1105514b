#!/usr/bin/env python3
"""Can a cheap streaming probe tell, at allocation time, which level the tail kernel will run at on a set of freshly allocated arrays (profiles/r06/tail_kernel_levels.txt)?
K candidate sets of the 18 arrays are allocated and HELD together (so each lies on different physical pages), each is probed (a read-modify-write pass over every array,
best of 5), then installed into the trainer and the real step is timed on it. GPU.   python tools/placement_probe.py [K]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


@torch.no_grad()
def probe(cand):
    cand = [t.detach() for t in cand]
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch._foreach_mul_(cand, 1.0)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    import lichtfeld_studio_amd as lfs  # noqa: F401
    from lichtfeld_studio_amd import capi, scenes
    from lichtfeld_studio_amd.trainer import GutTrainer
    from tail_placement import arrays, timed
    dev = torch.device("cuda:0")
    scene = scenes.syn_b()
    target = [scenes.target_image(scene.height, scene.width, seed=43).to(dev)]
    tr = GutTrainer(scene, dev, iterations=7000, world=1, rank=0, views_per_rank=1)
    tr.iteration = 3000
    tr.train_step(target); torch.cuda.synchronize()
    arr = arrays(tr)
    init = [t.detach().clone() for _, t in arr]
    print("as allocated: probe %.4f ms" % probe([t for _, t in arr]), timed(tr, target, capi, 60), flush=True)
    cands = [[torch.empty_like(t) for t in init] for _ in range(K)]
    for c in cands:
        for d, s in zip(c, init):
            d.copy_(s)
    torch.cuda.synchronize()
    pr = [probe(c) for c in cands]
    for i, c in enumerate(cands):
        for d, s in zip(c, init):
            d.copy_(s)
        j = 0
        for p in tr.model.parameters():
            st = tr.optimizer.state.get(id(p))
            p.data = c[j]; j += 1
            if st is not None:
                st["exp_avg"] = c[j]; st["exp_avg_sq"] = c[j + 1]; j += 2
        tr.iteration = 3000
        print(f"candidate {i}: probe {pr[i]:.4f} ms (again {probe(c):.4f})", timed(tr, target, capi, 60), flush=True)


if __name__ == "__main__":
    main()
