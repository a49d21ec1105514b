#!/usr/bin/env python3
"""Start / end / queue of every kernel of ONE training step out of a `rocprofv3 --kernel-trace --output-format csv` directory (the last-but-one complete step of the run):
what ran beside what on the two streams of the pipelined step.   python tools/step_timeline.py <trace dir> [<trace dir> ...]"""
import csv
import glob
import re
import sys


def timeline(d):
    f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    if not f:
        print(d, "no trace"); return
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "projection_ut_kernel" in r["Kernel_Name"]]
    if len(idx) < 4:
        print(d, "too few steps"); return
    a, b = idx[-3], idx[-2]
    t0 = int(rows[a]["Start_Timestamp"])
    print(f"== {d}: one step = {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
    end = max(int(r["End_Timestamp"]) for r in rows[a:b])
    for r in rows[a:]:
        if int(r["Start_Timestamp"]) > end:
            break
        name = re.sub(r"[(<].*", "", r["Kernel_Name"]).replace("void ", "").replace("lfs::", "")
        print(f"  {name:32s} q{r.get('Queue_Id', '?'):>3s} start {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f}  end {(int(r['End_Timestamp']) - t0) / 1e3:8.1f}  dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us")


for d in sys.argv[1:]:
    timeline(d)
