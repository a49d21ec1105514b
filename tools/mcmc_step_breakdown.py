"""Where does a 3DGUT + MCMC training iteration spend its time (SYN-B, 1M Gaussians)? Wall time with syncs around the parts."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lichtfeld_studio_amd  # noqa
from lichtfeld_studio_amd import scenes, strategies
from lichtfeld_studio_amd.trainer import GutTrainer
dev = torch.device("cuda:0")
sc = scenes.syn_b()
t = [scenes.target_image(sc.height, sc.width).to(dev)]
def timed(tr, n=60, label=""):
    for _ in range(10): tr.train_step(t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.train_step(t)
    torch.cuda.synchronize(); print(label, round((time.perf_counter() - t0) / n * 1e3, 3), "ms/iter", flush=True)
tr = GutTrainer(sc, dev, iterations=30000, loss="l1_ssim"); tr.iteration = 3001
timed(tr, label="no strategy, l1_ssim:")
op = strategies.OptimizationParameters(iterations=30000)
tr = GutTrainer(sc, dev, iterations=30000, loss="l1_ssim", strategy="mcmc", opt_params=op); tr.iteration = 3001
timed(tr, n=90, label="mcmc (refines every 100), l1_ssim:")
st = tr.strategy
acc = {"post_backward": 0.0, "step": 0.0}
for name in acc:
    orig = getattr(st, name)
    def wrap(*a, _o=orig, _n=name, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = _o(*a, **k); torch.cuda.synchronize(); acc[_n] += time.perf_counter() - t0; return r
    setattr(st, name, wrap)
n = 50
tr.iteration = 3101
for _ in range(n): tr.train_step(t)
print({k: round(v / n * 1e3, 3) for k, v in acc.items()}, "ms/iter (synced)")
