#!/bin/bash
# BASELINE.json configs[3] (per-GPU share: 3M Gaussians, 1600x1200, 8 views per step and rank) and configs[4] (2M Gaussians, MCMC strategy + Relocation
# kernel + bilateral grid, L1 + D-SSIM loss) on ONE GPU: bench lines + rocprofv3 kernel stats.   gpurun --timeout 1500 -- 'bash tools/r2_configs.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/configs; mkdir -p $OUT; cd $REPO
timeout 600 python bench.py --workload syn-c --views-per-rank 8 --steps 10 --warmup 4 > $OUT/bench_config4_sync_8views.json 2> $OUT/bench_config4.err
timeout 600 python bench.py --workload syn-d --strategy mcmc --bilateral-grid --loss l1_ssim --steps 100 --warmup 12 > $OUT/bench_config5_synd_mcmc_bilateral.json 2> $OUT/bench_config5.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace4 -o trace -- python $REPO/bench.py --workload syn-c --views-per-rank 8 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/trace4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace5 -o trace -- python $REPO/bench.py --workload syn-d --strategy mcmc --bilateral-grid --loss l1_ssim --steps 100 --warmup 12 --no-cpu-baseline > $OUT/trace5.log 2>&1
cd $REPO
python - <<'PY'
import json, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "configs")
for f in ("bench_config4_sync_8views.json", "bench_config5_synd_mcmc_bilateral.json"):
    try:
        d = json.loads(open(os.path.join(out, f)).read())
        print(f, d["value"], d["ms_per_step"], d["config"]["visible_gaussians"], d["config"]["n_isects"], d["roofline"])
        print({k: (v["avg_ms"], v["launches_per_step"]) for k, v in d["kernels"].items()})
        print((d.get("cpu_baseline") or {}).get("parity_vs_oracle"))
    except Exception as e:
        print(f, "failed:", e); print(open(os.path.join(out, f.replace(".json", ".err").replace("_sync_8views", "").replace("_synd_mcmc_bilateral", ""))).read()[-2000:])
PY
