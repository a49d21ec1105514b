"""Time loader.mean_neighbor_distances (the reference's eps = 10 nanoflann query: host tree build + GPU walk) and the exact all-pairs kernel at point-cloud sizes."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import lichtfeld_studio_amd  # noqa: F401
from lichtfeld_studio_amd import loader

out = {}
for N in (100_000, 1_000_000):
    pts = torch.from_numpy(np.random.default_rng(0).standard_normal((N, 3)).astype(np.float32)).cuda()
    for name, kw in (("reference_query", {}), ("exact", {"exact": True})):
        loader.mean_neighbor_distances(pts[:1000], **kw)
        torch.cuda.synchronize()
        t = time.time()
        r = loader.mean_neighbor_distances(pts, **kw)
        torch.cuda.synchronize()
        out[f"{name}_{N}"] = {"seconds": round(time.time() - t, 4), "mean": float(r.mean())}
print(json.dumps(out))
