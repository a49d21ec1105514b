#!/bin/bash
# Round 6, lease 9: the pipelined step (lfs_gut_train_step_pipelined: SH Adam pass + SH colours on the library's side stream) - its tests, then the alternating A/B against
# the one-stream step on the same box, then a kernel trace of both (start / end timestamps per kernel: what actually ran beside what)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease9; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
timeout 900 python -m pytest tests/test_gpu_000_canary.py tests/test_gpu_gut_step.py -q -m gpu -p no:cacheprovider -x > $OUT/tests.log 2>&1; echo "tests rc $?: $(tail -1 $OUT/tests.log)"
grep -n "FAILED\|Error\|assert" $OUT/tests.log | head -20
for r in 1 2 3; do for v in serial pipelined; do
  if [ $v = serial ]; then F=--no-pipeline; else F=; fi
  timeout 300 python bench.py --no-cpu-baseline --steps 300 --warmup 20 $F 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('[$v]', d['value'], d['ms_per_step'], d['config'].get('step_form'), {k: v['avg_ms'] for k, v in d['kernels'].items()})"
done; done 2>&1 | tee $OUT/ab.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_default.json
python -c "
import json; d = json.load(open('$OUT/bench_default.json')); print('driver command:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline'].get('parity_vs_oracle', {}).get('grad_rel_l2'))"
cd /tmp && export TMPDIR=/tmp
for v in pipelined serial; do
  if [ $v = serial ]; then F=--no-pipeline; else F=; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/trace_$v -o t -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile $F > $REPO/$OUT/trace_$v.log 2>&1
done
cd $REPO; python - <<'PY'
import csv, glob, re
for v in ("pipelined", "serial"):
    f = glob.glob(f"gpurun_out/r6_lease9/trace_{v}/**/*kernel_trace.csv", recursive=True)
    if not f: print(v, "no trace"); continue
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last complete step: from the last-but-one projection kernel to the last one
    idx = [i for i, r in enumerate(rows) if "projection_ut_kernel" in r["Kernel_Name"]]
    if len(idx) < 3: print(v, "too few steps"); continue
    a, b = idx[-3], idx[-2]
    t0 = int(rows[a]["Start_Timestamp"])
    print(f"== {v}: one step = {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
    for r in rows[a:b + 1]:
        name = re.sub(r"[(<].*", "", r["Kernel_Name"]).replace("void ", "").replace("lfs::", "")
        print(f"  {name:32s} q{r.get('Queue_Id', '?'):>3s} start {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} us  dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us")
PY
