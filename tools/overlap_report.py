#!/usr/bin/env python3
"""Do the collective library's kernels run BESIDE ours? From a rocprofv3 kernel trace (CSV) of a data-parallel step: every kernel whose name contains nccl / rccl, its duration,
and how much of it overlaps in time with lfs:: kernels (which ones). World size 1 with LFS_DIST_FORCE_COLLECTIVES=1 executes every collective of a layout through RCCL on one
GPU: this shows the STREAM structure of the step (which collectives are off the compute stream and hidden under which kernels), not inter-GPU bandwidth.

    python tools/overlap_report.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
    rows.sort()
    ours = [r for r in rows if "lfs::" in r[2]]
    coll = [r for r in rows if "nccl" in r[2].lower() or "rccl" in r[2].lower()]
    if not coll:
        print("no collective-library kernel in the trace")
        return
    short = lambda n: n.split("(")[0].replace("void ", "").replace("lfs::", "")[:70]
    agg = defaultdict(lambda: [0, 0.0, 0.0, defaultdict(float)])
    for s, e, name, q, st in coll:
        a = agg[short(name)]
        a[0] += 1; a[1] += (e - s) / 1e3
        for s2, e2, n2, q2, st2 in ours:
            if s2 >= e:
                break
            ov = min(e, e2) - max(s, s2)
            if ov > 0:
                a[2] += ov / 1e3; a[3][short(n2)] += ov / 1e3
    print(f"{'collective kernel':70s} {'launches':>8} {'total us':>10} {'beside lfs:: kernels':>22}")
    for k, (n, tot, ov, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:70s} {n:8d} {tot:10.1f} {ov:12.1f} us = {100 * ov / max(tot, 1e-9):5.1f} %")
        for kk, v in sorted(by.items(), key=lambda kv: -kv[1])[:4]:
            print(f"      under {kk:60s} {v:10.1f} us")
    qs = sorted({(r[3], r[4]) for r in coll}), sorted({(r[3], r[4]) for r in ours})
    print("queues / streams of the collective kernels:", qs[0], "| of lfs:: kernels:", qs[1])


if __name__ == "__main__":
    main()
