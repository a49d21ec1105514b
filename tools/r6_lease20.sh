#!/bin/bash
# Round 6, lease 20: the C++ one-call step of INTEGRATION.md 1b on the fused tail (lfs::GutTrainStep::step over lfs_gut_train_step_ex, optional next_viewmat) and the
# pinned-count read-back of gsplat::intersect_tile: the reference-links / torch-ops / step tests, then the driver's command once
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease20; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
timeout 1200 python -m pytest tests/test_gpu_reference_links.py tests/test_gpu_torch_ops.py tests/test_gpu_gut_step.py tests/test_gpu_intersect.py -q -m gpu -p no:cacheprovider -s > $OUT/tests.log 2>&1; echo "tests rc $?: $(tail -1 $OUT/tests.log)"
grep -n "^FAILED\|^ERROR\|linked reference on SYN-B" $OUT/tests.log | head -20
cp gpurun_out/reference_links_synb_timing.json $OUT/ 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_default.json
python -c "
import json; d = json.load(open('$OUT/bench_default.json')); print('driver command:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['config'].get('step_form'), d.get('ops_route', {}))"
