#!/bin/bash
# Round 6, lease 10: the pipelined step under its measurement knobs - side stream priority (low / high), side stream confined to every 2nd / 3rd / 4th CU, SH Adam pass started
# behind the finish pass instead of beside it - alternating against the one-stream step on one box; then a kernel trace (with start / end timestamps) of the best two
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease10; mkdir -p $OUT
run() { # name, bench flags, env...
  local name=$1 flags=$2; shift 2
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-ops-route --steps 200 --warmup 20 $flags 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('[$name]', d['value'], d['ms_per_step'], {k: v['avg_ms'] for k, v in d['kernels'].items()})"
}
for r in 1 2; do
  run serial --no-pipeline X=1
  run pipe_default "" X=1
  run pipe_low "" LFS_PIPE_PRIO=low
  run pipe_high "" LFS_PIPE_PRIO=high
  run pipe_after_finish "" LFS_PIPE_START=finish
  run pipe_after_finish_low "" LFS_PIPE_START=finish LFS_PIPE_PRIO=low
  run pipe_cu2 "" LFS_PIPE_CUMASK=2
  run pipe_cu3 "" LFS_PIPE_CUMASK=3
  run pipe_cu4 "" LFS_PIPE_CUMASK=4
  run pipe_after_finish_cu2 "" LFS_PIPE_START=finish LFS_PIPE_CUMASK=2
  run pipe_after_finish_cu3 "" LFS_PIPE_START=finish LFS_PIPE_CUMASK=3
done 2>&1 | tee $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in serial default after_finish; do
  F=; E="X=1"
  [ $v = serial ] && F=--no-pipeline
  [ $v = after_finish ] && E="LFS_PIPE_START=finish"
  env $E rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/trace_$v -o t -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-ops-route $F > $REPO/$OUT/trace_$v.log 2>&1
done
cd $REPO; python - <<'PY'
import csv, glob, re
for v in ("serial", "default", "after_finish"):
    f = glob.glob(f"gpurun_out/r6_lease10/trace_{v}/**/*kernel_trace.csv", recursive=True)
    if not f: print(v, "no trace"); continue
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "projection_ut_kernel" in r["Kernel_Name"]]
    if len(idx) < 4: print(v, "too few steps"); continue
    a, b = idx[-3], idx[-2]
    t0 = int(rows[a]["Start_Timestamp"])
    print(f"== {v}: one step = {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
    end = max(int(r["End_Timestamp"]) for r in rows[a:b])
    for r in rows[a:]:
        if int(r["Start_Timestamp"]) > end + 400000: break
        name = re.sub(r"[(<].*", "", r["Kernel_Name"]).replace("void ", "").replace("lfs::", "")
        print(f"  {name:32s} q{r.get('Queue_Id', '?'):>3s} start {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f}  end {(int(r['End_Timestamp']) - t0) / 1e3:8.1f}  dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us")
PY
