"""How much would a longest-first block order buy the rasterizer kernels? Greedy list scheduling of the per-tile work (tile list lengths of SYN-B
view 0) on P concurrent workgroup slots, natural tile order vs longest-processing-time-first."""
import sys, os, heapq, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lichtfeld_studio_amd as lfs
from lichtfeld_studio_amd import ops, scenes
dev = torch.device("cuda:0")
sc = scenes.syn_b(n_views=1).to(dev)
q = torch.nn.functional.normalize(sc.raw_quats, dim=-1); s = sc.raw_scales.exp(); o = torch.sigmoid(sc.raw_opacities)
W, H = sc.width, sc.height
radii, m2, d, _, _ = ops.projection_ut_3dgs_fused(sc.means, q, s, o, sc.viewmats[:1].contiguous(), None, sc.Ks[:1].contiguous(), W, H, 0.3, 0.01, 1e4, 0.0, False, lfs.CameraModelType.PINHOLE)
tw, th = (W + 15) // 16, (H + 15) // 16
_, _, flat, offs = ops.intersect_tile(m2, radii, d, None, None, 1, 16, tw, th, True, return_offsets=True)
off = offs.reshape(-1).cpu().numpy().astype(np.int64)
cnt = np.diff(np.append(off, flat.shape[0])).astype(np.float64) + 30.0     # + a fixed per-block cost (prologue / epilogue), in list entries
def makespan(order, P):
    h = [0.0] * P
    heapq.heapify(h)
    for t in order:
        heapq.heappush(h, heapq.heappop(h) + cnt[t])
    return max(h)
res = {"tiles": int(len(cnt)), "mean": float(cnt.mean()), "max": float(cnt.max()), "p99": float(np.percentile(cnt, 99))}
for P in (1024, 2048):
    nat = makespan(range(len(cnt)), P)
    lpt = makespan(np.argsort(-cnt), P)
    res[f"P{P}"] = {"natural": nat, "lpt": lpt, "ideal": float(cnt.sum() / P), "gain_pct": round(100 * (nat - lpt) / nat, 2)}
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "synb_tile_counts.npy"), np.diff(np.append(off, flat.shape[0])))
print(json.dumps(res))
