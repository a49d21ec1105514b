"""Developer script (GPU): how evenly the tile-list work of one SYN-B view spreads over the 8 XCDs under the rasterizer's workgroup -> tile mappings
(cell_ctx, lfs_raster_common.cuh): contiguous bands (one per XCD) against finer interleaved bands.   python tools/band_balance.py [view]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lichtfeld_studio_amd import scenes  # noqa: E402
from lichtfeld_studio_amd.gut_step import GutStep  # noqa: E402

view = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = "cuda:0"
sc = scenes.syn_b().to(dev)
gs = GutStep(dev)
ps = [sc.means, sc.sh0, sc.shN, sc.raw_scales, sc.raw_quats, sc.raw_opacities]
bg = torch.zeros(3, device=dev)
n = gs.view_forward([p.contiguous() for p in ps], sc.sh_degree, sc.width, sc.height, sc.viewmats[view].contiguous(), sc.Ks[view].contiguous(), bg)
tw, th = (sc.width + 15) // 16, (sc.height + 15) // 16
T = tw * th
off = gs.view("tile_offsets", torch.int32, (T + 1,)).cpu().numpy().astype(np.int64)
cnt = np.diff(off)
print(f"view {view}: {n} intersections, {T} tiles ({tw} x {th}), per tile mean {cnt.mean():.0f} max {cnt.max()} min {cnt.min()}")
rows = cnt.reshape(th, tw).sum(axis=1)
print("per tile row (of %d):" % th, " ".join(str(int(r // 1000)) + "k" for r in rows))
for bands in (1, 2, 4, 8, 17):
    # 8 * bands contiguous chunks of tiles, chunk c -> XCD c % 8
    edges = np.linspace(0, T, 8 * bands + 1).astype(int)
    work = np.zeros(8)
    for c in range(8 * bands):
        work[c % 8] += cnt[edges[c]:edges[c + 1]].sum()
    print(f"{bands:2d} band(s) per XCD: max / mean XCD work = {work.max() / work.mean():.3f}  (per XCD, k entries: {' '.join(str(int(w // 1000)) for w in work)})")
