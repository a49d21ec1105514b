"""Summarise rocprofv3 CSV output of tools/profile.sh: per-kernel average duration from the
kernel trace and per-kernel PMC sums (FETCH_SIZE / WRITE_SIZE in KiB-units as reported; see
MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide
coalesced reads — double it before comparing with a byte count)."""
import csv, glob, os, re, sys
from collections import defaultdict

out = sys.argv[1]

def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("lfs::", "")
    return name[:60]

def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))

# kernel trace
for f in find("trace/**/*kernel_trace.csv"):
    dur = defaultdict(list)
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"== kernel trace {os.path.relpath(f, out)} (us)")
    tot = sum(sum(v) for v in dur.values())
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:40]:
        print(f"{k:60s} calls {len(v):5d} avg {sum(v)/len(v):10.2f} total {sum(v):12.1f} {100*sum(v)/tot:5.1f}%")

for tag in ["pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"]:
    for f in find(f"{tag}/**/*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r.get("Dispatch_Id"), k)
            if key not in seen:
                seen.add(key); cnt[k] += 1
        print(f"== {tag} {os.path.relpath(f, out)} (per-launch averages)")
        for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:30]:
            n = max(cnt[k], 1)
            print(f"{k:60s} launches {n:4d} " + " ".join(f"{c}={v/n:.4g}" for c, v in sorted(acc[k].items())))
