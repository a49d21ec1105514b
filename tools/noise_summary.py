"""Summarise LFS_NOISE_LOG files (tests/gpu_util.py noise_check): per label, the runs, the largest value : bar ratio, the bar. Sorted worst first."""
import collections
import json
import sys

by = collections.defaultdict(list)
for path in sys.argv[1:]:
    for line in open(path):
        r = json.loads(line)
        by[r["label"]].append(r)
rows = sorted(((max(x["ratio"] for x in v), k, v) for k, v in by.items()), reverse=True)
print(f"{len(sys.argv) - 1} suite runs, {len(rows)} noise-limited assertions; columns: max ratio | mean ratio | samples | largest bar | label")
for mx, k, v in rows:
    print(f"{mx:7.3f} | {sum(x['ratio'] for x in v) / len(v):7.3f} | {len(v):3d} | {max(x['bar'] for x in v):.2e} | {k}")
print("worst ratio:", f"{rows[0][0]:.3f}" if rows else "n/a")
