#!/usr/bin/env python3
"""Launch sequence of one steady-state training step from a rocprofv3 kernel trace (the CSV of `--kernel-trace --output-format csv`).

    python tools/step_sequence.py <dir with *_kernel_trace.csv> [anchor substring, default projection_ut_kernel]

Prints the kernels between the last two launches of the anchor kernel in start order: duration, the idle gap before each, and a footer with the
kernels that are not ours (torch fills / copies, rocclr blits) - the "stray launches" of the step."""
import csv, glob, os, sys


def main():
    d = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "projection_ut_kernel"
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *kernel_trace.csv under {d}")
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if anchor in r[2]]
    if len(marks) < 3:
        raise SystemExit("fewer than three anchor launches in the trace")
    a, b = marks[-3], marks[-2]   # the last complete step but one (the last one may be followed by teardown work)
    step = rows[a:b]
    t0, prev_end, busy, stray = step[0][0], step[0][0], 0, []
    print(f"{'start_us':>9} {'dur_us':>8} {'gap_us':>7}  kernel")
    for s, e, name in step:
        short = name.split("(")[0].replace("void ", "")[:110]
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}  {short}")
        busy += e - s
        prev_end = max(prev_end, e)
        if "lfs::" not in name:
            stray.append((short, (e - s) / 1e3))
    wall = rows[b][0] - t0
    print(f"step wall {wall / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, {len(step)} launches, {len(stray)} not lfs:: ({sum(x[1] for x in stray):.1f} us):")
    for name, us in stray:
        print(f"   {us:7.1f} us  {name}")


if __name__ == "__main__":
    main()
