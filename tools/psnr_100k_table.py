#!/usr/bin/env python3
"""Round 5: the 7 000-iteration PSNR criterion on the 100 000-Gaussian / 960x540 / SH 3 flat-disk task - the oracle's trajectories (CPU, this container:
profiles/r05/convergence_mse_100k_oracle.json) next to the HIP runs of the same seeds (GPU box: profiles/r05/lease11/psnr_100k_hip.txt). No GPU needed.
    python tools/psnr_100k_table.py          -> profiles/r05/psnr_100k_summary.json + one line per seed"""
import json
import math
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ora = json.load(open(os.path.join(ROOT, "profiles", "r05", "convergence_mse_100k_oracle.json")))
hip = json.loads(open(os.path.join(ROOT, "profiles", "r05", "lease11", "psnr_100k_hip.txt")).read().strip().splitlines()[-1])["seeds"]
rows, gd, ga = {}, [], []
for s in sorted(ora["seeds"], key=int):
    if s not in hip:
        continue
    o = ora["seeds"][s]["oracle_psnr_oracle_renderer"]
    h = hip[s]
    rows[s] = {"oracle": o, "hip_deterministic": h["hip_deterministic"], "deterministic_runs_bit_identical": h["deterministic_runs_bit_identical"], "hip_atomic": h["hip_atomic"],
               "gap_deterministic_db": round(h["hip_deterministic"] - o, 4), "gap_atomic_db": [round(x - o, 4) for x in h["hip_atomic"]]}
    gd.append(h["hip_deterministic"] - o)
    ga += [x - o for x in h["hip_atomic"]]
    print(f"seed {s}: oracle {o:.4f} dB | HIP deterministic {h['hip_deterministic']:.4f} ({gd[-1]:+.4f}) | float atomics " + ", ".join(f"{x:.4f} ({x - o:+.4f})" for x in h["hip_atomic"]))
out = {"task": ora["task"], "library": open(os.path.join(ROOT, "profiles", "r05", "lease11", "library.txt")).read().strip().splitlines()[-1], "seeds": rows,
       "summary": {"n_seeds": len(gd), "mean_gap_deterministic_db": round(float(np.mean(gd)), 4), "max_abs_gap_deterministic_db": round(float(np.max(np.abs(gd))), 4),
                   "mean_gap_atomic_db": round(float(np.mean(ga)), 4), "max_abs_gap_atomic_db": round(float(np.max(np.abs(ga))), 4),
                   "atomic_runs_within_0p05": int(sum(abs(x) <= 0.05 for x in ga)), "atomic_runs": len(ga),
                   "std_gap_deterministic_db": round(float(np.std(gd, ddof=1)), 4) if len(gd) > 1 else None}}
json.dump(out, open(os.path.join(ROOT, "profiles", "r05", "psnr_100k_summary.json"), "w"), indent=1)
print(json.dumps(out["summary"]))
