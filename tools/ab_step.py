"""A/B timing of module-level switches of the fused step on the same box (alternating runs): python tools/ab_step.py FLAG [reps]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lichtfeld_studio_amd  # noqa
from lichtfeld_studio_amd import fused, scenes
from lichtfeld_studio_amd.trainer import GutTrainer
flag = sys.argv[1]            # fused.<FLAG> or fastgs.<FLAG> (the latter runs the fastgs rasterizer)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mod = fused
if flag.startswith("fastgs."):
    from lichtfeld_studio_amd import fastgs as mod
flag = flag.split(".")[-1]
dev = torch.device("cuda:0")
sc = scenes.syn_b()
tr = GutTrainer(sc, dev, iterations=7000, rasterizer="fastgs" if mod is not fused else "gut")
tr.iteration = 3000
t = [scenes.target_image(sc.height, sc.width).to(dev)]
def run(v, n=40):
    setattr(mod, flag, v)
    for _ in range(5): tr.train_step(t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.train_step(t)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(reps):
    print(flag, "True", round(run(True), 4), "False", round(run(False), 4), flush=True)
