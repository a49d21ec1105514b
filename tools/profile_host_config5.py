"""Developer: where does the HOST time of a config-5 step (SYN-D, MCMC + bilateral grid + L1/D-SSIM) go? cProfile over 60 steps."""
import cProfile, pstats, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import lichtfeld_studio_amd as lfs
from lichtfeld_studio_amd import scenes, strategies
from lichtfeld_studio_amd.trainer import GutTrainer
dev = torch.device("cuda:0")
sc = scenes.syn_d()
tr = GutTrainer(sc, dev, iterations=30000, loss="l1_ssim", strategy="mcmc", opt_params=strategies.OptimizationParameters(iterations=30000, max_cap=sc.N), use_bilateral_grid=True)
tr.iteration = 3000
tg = [scenes.target_image(sc.height, sc.width, seed=43).to(dev)]
for _ in range(5): tr.train_step(tg)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for _ in range(60): tr.train_step(tg)
torch.cuda.synchronize()
pr.disable()
print("ms/step", (time.perf_counter() - t0) / 60 * 1e3)
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
