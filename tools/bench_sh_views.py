"""Owner-side SH cost of the SH-sharded layout at world = 8 (N = 1M: 125k rows, 8 views), timed on one GPU."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lichtfeld_studio_amd  # noqa
from lichtfeld_studio_amd import fused, scenes
from lichtfeld_studio_amd.fused_adam import FusedAdam
dev = "cuda:0"
sc = scenes.syn_b().to(dev)
world, N = 8, sc.means.shape[0]
S = N // world
means, sh0, shN = sc.means[:S].contiguous(), sc.sh0[:S].contiguous(), sc.shN[:S].clone()
vms = sc.viewmats[:world].contiguous()
radii = torch.full((world, S, 2), 3, dtype=torch.int32, device=dev)
v_colors = torch.randn(world, S, 3, device=dev)
opt = FusedAdam([{"params": [shN], "lr": 1e-4}])
def t(fn, n=50):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
colors = fused.sh_model_fwd_views(3, means, vms, sh0, shN, radii)
g0, gm = torch.empty_like(sh0), torch.zeros(S, 3, device=dev)
res = {"rows": S, "views": world,
       "fwd_views_ms": t(lambda: fused.sh_model_fwd_views(3, means, vms, sh0, shN, radii)),
       "bwd_views_adam_ms": t(lambda: fused.sh_model_bwd_views(3, means, vms, sh0, shN, radii, colors, v_colors, g0, None, gm, False, adam=opt.prepare_inline(shN))),
       "per_view_fwd_x8_ms": t(lambda: [fused.sh_model_fwd(3, means, vms[j:j + 1], sh0, shN, radii[j:j + 1]) for j in range(world)])}
print(json.dumps(res))
