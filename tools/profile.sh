#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/profile.sh r03'
# Writes gpurun_out/prof_<tag>/...; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ops-route"   # the driver's command without its two side legs (oracle check, Ops.h route): the headline path only
# 1) kernel trace + stats (same command as the bench line, events enabled)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/bench_trace.log" 2>&1
# 2) PMC passes, each on its own (TCC slots: FETCH_SIZE costs 3, WRITE_SIZE 2)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- $BENCH --no-profile > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- $BENCH --no-profile > "$OUT/bench_write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq" -o pmc -- $BENCH --no-profile > "$OUT/bench_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD --output-format csv -d "$OUT/pmc_sq2" -o pmc -- $BENCH --no-profile > "$OUT/bench_sq2.log" 2>&1
find "$OUT" -name "*.csv" | head -50
python $REPO/tools/summarize_prof.py "$OUT" --json "$OUT/summary.json" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt" | head -80
