"""Summarise rocprofv3 CSV output of tools/profile.sh: per-kernel STEADY-STATE duration from the kernel trace and per-kernel PMC averages
(FETCH_SIZE / WRITE_SIZE in KiB-units as reported; see MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide
coalesced reads - double it before comparing with a byte count).

Round 6 ("What's weak" 1): a launch whose duration is below HALF the median of its kernel is not a steady-state launch - the
emptied-list launches of an aborted speculative step (csrc/gut_step.hip re-runs the attempt with a larger workspace), the oracle-side reference step of the
bench - and is left out of every figure: the trace table prints median / mean / min / max / n of the launches that remain (and how many were dropped), and the
PMC tables average over the same set. The PMC passes carry no durations of their own, so there a launch is dropped when ITS position in the kernel's launch
sequence was dropped in the trace of the same command (the bench is deterministic: the same attempt aborts in every pass); when the counts of a kernel differ
between the trace and a PMC pass the launch with the smallest SQ_WAVES / counter sum below half the median is dropped instead.

    python tools/summarize_prof.py <dir of tools/profile.sh> [--json out.json]
"""
import csv, glob, json, os, re, statistics, sys
from collections import defaultdict

out = sys.argv[1]
json_out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("lfs::", "")
    return name[:60]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


def steady(values):
    """indices of the launches kept: those at or above half the median"""
    if len(values) < 3:
        return list(range(len(values)))
    med = statistics.median(values)
    return [i for i, v in enumerate(values) if v >= 0.5 * med]


summary = {"trace": {}, "pmc": {}}
dropped_positions = {}   # kernel -> set of positions (in launch order) dropped in the trace

for f in find("trace/**/*kernel_trace.csv"):
    rows = defaultdict(list)
    for r in csv.DictReader(open(f)):
        rows[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    print(f"== kernel trace {os.path.relpath(f, out)} (us; launches below half their kernel's median are dropped: aborted speculative attempts, checker steps)")
    stats = {}
    for k, v in rows.items():
        v.sort()
        d = [x[1] for x in v]
        keep = steady(d)
        dropped_positions[k] = set(range(len(d))) - set(keep)
        kd = [d[i] for i in keep]
        stats[k] = dict(n=len(kd), dropped=len(d) - len(kd), median=statistics.median(kd), mean=sum(kd) / len(kd), min=min(kd), max=max(kd), total=sum(kd))
    tot = sum(s["total"] for s in stats.values())
    for k, s in sorted(stats.items(), key=lambda kv: -kv[1]["total"])[:40]:
        print(f"{k:60s} n {s['n']:4d} (dropped {s['dropped']}) median {s['median']:9.2f} mean {s['mean']:9.2f} min {s['min']:9.2f} max {s['max']:9.2f} total {s['total']:11.1f} {100 * s['total'] / tot:5.1f}%")
    summary["trace"] = stats

for tag in ["pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"]:
    for f in find(f"{tag}/**/*counter_collection.csv"):
        per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))   # kernel -> dispatch id -> counter -> value
        for r in csv.DictReader(open(f)):
            per[short(r["Kernel_Name"])][int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        print(f"== {tag} {os.path.relpath(f, out)} (per-launch averages over the steady-state launches)")
        table = {}
        for k, disp in per.items():
            ids = sorted(disp)
            drop = dropped_positions.get(k, set())
            if k in summary["trace"] and len(ids) == summary["trace"][k]["n"] + summary["trace"][k]["dropped"]:
                keep = [d for i, d in enumerate(ids) if i not in drop]
            else:   # launch counts differ from the trace: fall back to the counters themselves
                sums = [sum(disp[d].values()) for d in ids]
                keep = [ids[i] for i in steady(sums)]
            n = max(len(keep), 1)
            acc = defaultdict(float)
            for d in keep:
                for c, v in disp[d].items():
                    acc[c] += v
            table[k] = dict(n=len(keep), dropped=len(ids) - len(keep), **{c: v / n for c, v in acc.items()})
        for k in sorted(table, key=lambda k: -sum(v for c, v in table[k].items() if c not in ("n", "dropped")))[:30]:
            t = table[k]
            print(f"{k:60s} n {t['n']:4d} (dropped {t['dropped']}) " + " ".join(f"{c}={v:.4g}" for c, v in sorted(t.items()) if c not in ("n", "dropped")))
        summary["pmc"][tag] = table

if json_out:
    with open(json_out, "w") as fh:
        json.dump(summary, fh, indent=1)
