"""Developer: time the pieces of one MCMC refinement step (relocate_gs) at SYN-D size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import lichtfeld_studio_amd as lfs
from lichtfeld_studio_amd import scenes, strategies, ops
from lichtfeld_studio_amd.trainer import GutTrainer
dev = torch.device("cuda:0")
sc = scenes.syn_d()
tr = GutTrainer(sc, dev, iterations=30000, loss="l1_ssim", strategy="mcmc", opt_params=strategies.OptimizationParameters(iterations=30000, max_cap=sc.N), use_bilateral_grid=True)
tr.iteration = 3090
tg = [scenes.target_image(sc.height, sc.width, seed=43).to(dev)]
for _ in range(8): tr.train_step(tg)
st = tr.strategy; m = st.model
def T(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); print(f"{label:40s} {(time.perf_counter()-t0)*1e3:9.3f} ms"); return r
with torch.no_grad():
    opac = T("get_opacity", lambda: m.get_opacity().detach())
    dead = T("dead mask", lambda: (opac <= st.params.min_opacity) | ((m.raw_quats.detach() ** 2).sum(-1) < 1e-8))
    dead_idx = T("nonzero", lambda: dead.nonzero().squeeze(-1))
    print("n_dead", dead_idx.numel())
    alive_idx = T("alive nonzero", lambda: (~dead).nonzero().squeeze(-1))
    w = T("index_select", lambda: opac.index_select(0, alive_idx))
    s = T("multinomial", lambda: st.multinomial_sample(w, int(dead_idx.numel()), True))
    cdf = T("cumsum f64", lambda: torch.cumsum(w.double() / w.double().sum(), 0))
    T("searchsorted", lambda: torch.searchsorted(cdf, torch.rand(int(dead_idx.numel()), device=dev, dtype=torch.float64)))
torch.cuda.synchronize(); t0 = time.perf_counter(); tr.iteration = 3099; tr.train_step(tg); torch.cuda.synchronize(); print("refine step total", (time.perf_counter()-t0)*1e3)
torch.cuda.synchronize(); t0 = time.perf_counter(); tr.train_step(tg); torch.cuda.synchronize(); print("next step", (time.perf_counter()-t0)*1e3)
