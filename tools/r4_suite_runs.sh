#!/bin/bash
# tools/r4_suite_runs.sh <tag> <n_runs>: the driver's round-end GPU suite command, verbatim, <n_runs> times on this lease, every noise-limited
# assertion's value : bar logged (tests/gpu_util.py noise_check) -> profiles/r04/suite_<tag>/ (pytest tails, noise ratio summary, library stamp).
set -u
TAG=${1:-a}; N=${2:-3}
OUT=gpurun_out/suite_$TAG; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; from lichtfeld_studio_amd import capi; print(capi.load_library().lfs_version().decode())" > $OUT/library.txt 2>&1
rc_all=0
for i in $(seq 1 $N); do
  LFS_NOISE_LOG=$PWD/$OUT/noise_run$i.jsonl timeout 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider ${PYTEST_EXTRA:-} > $OUT/pytest_run$i.log 2>&1
  rc=$?; echo "run $i rc $rc: $(tail -1 $OUT/pytest_run$i.log)" | tee -a $OUT/summary.txt
  [ $rc -ne 0 ] && rc_all=$rc && tail -40 $OUT/pytest_run$i.log
  tail -5 $OUT/pytest_run$i.log > $OUT/pytest_run${i}_tail.log; [ $rc -eq 0 ] && rm $OUT/pytest_run$i.log
done
python tools/noise_summary.py $OUT/noise_run*.jsonl > $OUT/noise_ratios.txt; tail -25 $OUT/noise_ratios.txt
exit $rc_all
