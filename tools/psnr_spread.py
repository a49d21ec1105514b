"""Run-to-run spread of the HIP training trajectory (float atomics in the rasterizer backward make every run a different sample of a
chaotic optimisation): trains the convergence task of tests/convergence_check.py R times and prints the final PSNRs."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import convergence_check as cc  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
dev = torch.device("cuda:0")
gt, init = cc.make_task()
targets = cc.render_views_hip(gt, dev)
finals, mids = [], []
for r in range(R):
    hist, _ = cc.train_hip(init, targets, iters, iters, dev, 1000)
    finals.append(hist[iters]); mids.append(hist.get(2000))
print(json.dumps({"runs": R, "iters": iters, "psnr_final": [round(x, 3) for x in finals], "mean": round(float(np.mean(finals)), 3), "std": round(float(np.std(finals)), 3),
                  "psnr_2000": [round(x, 3) for x in mids if x is not None]}))
