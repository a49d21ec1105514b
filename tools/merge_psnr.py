#!/usr/bin/env python
"""Merges the per-call outputs of tests/convergence_l1ssim.py --hip (gpurun_out/psnr_*/convergence_mse_hip.json) into one file with the statistics over all seeds:
    python tools/merge_psnr.py out.json in1.json in2.json ..."""
import json
import sys

import numpy as np
from scipy.stats import t as student_t


def ci95(xs):
    return round(float(student_t.ppf(0.975, len(xs) - 1) * np.std(xs, ddof=1) / np.sqrt(len(xs))), 4) if len(xs) > 1 else None


def main():
    out_path, ins = sys.argv[1], sys.argv[2:]
    seeds, task = {}, None
    for f in ins:
        a = json.load(open(f))
        task = task or a.get("task")
        seeds.update(a["seeds"])
    keys = sorted(seeds, key=int)
    gaps = [seeds[k]["gap_deterministic_db"] for k in keys]
    atomic = [float(np.mean(seeds[k]["hip_atomic"])) - seeds[k]["oracle"] for k in keys if seeds[k].get("hip_atomic")]
    runs = [x - seeds[k]["oracle"] for k in keys for x in seeds[k].get("hip_atomic", [])]
    summary = {"n_seeds": len(keys), "oracle_mean": round(float(np.mean([seeds[k]["oracle"] for k in keys])), 4),
               "oracle_std": round(float(np.std([seeds[k]["oracle"] for k in keys], ddof=1)), 4),
               "hip_deterministic_mean": round(float(np.mean([seeds[k]["hip_deterministic"] for k in keys])), 4),
               "hip_deterministic_std": round(float(np.std([seeds[k]["hip_deterministic"] for k in keys], ddof=1)), 4),
               "mean_gap_deterministic_db": round(float(np.mean(gaps)), 4), "mean_gap_deterministic_ci95_db": ci95(gaps),
               "gap_std_over_seeds_db": round(float(np.std(gaps, ddof=1)), 4), "gap_min_db": min(gaps), "gap_max_db": max(gaps),
               "seeds_with_float_atomic_runs": len(atomic), "mean_gap_float_atomic_db": round(float(np.mean(atomic)), 4) if atomic else None,
               "mean_gap_float_atomic_ci95_db": ci95(atomic) if atomic else None,
               "float_atomic_runs": len(runs), "float_atomic_runs_within_0p05_db": int(sum(abs(x) <= 0.05 for x in runs)),
               "deterministic_mode_bit_identical_when_repeated": all(seeds[k].get("deterministic_runs_bit_identical", True) for k in keys),
               "deterministic_mode_repeated_on_seeds": int(sum("deterministic_runs_bit_identical" in seeds[k] for k in keys))}
    json.dump({"task": task, "summary": summary, "seeds": {k: seeds[k] for k in keys}}, open(out_path, "w"), indent=1)
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
