python tools/make_synthetic_colmap.py /tmp/syn_colmap --views 96 --width 1296 --height 840 --gaussians 300000 --points 60000 > /dev/null 2>&1
F='s/"plan".*//'
echo "== E3a gut, no strategy"; timeout 120 python tools/r4_scale_debug.py -d /tmp/syn_colmap --gut --strategy none -i 7500 --every 2500 2>&1 | grep iter | sed "$F" | cut -c1-330
echo "== E3b fastgs, no strategy"; timeout 120 python tools/r4_scale_debug.py -d /tmp/syn_colmap --strategy none -i 7500 --every 2500 2>&1 | grep iter | sed "$F" | cut -c1-330
echo "== E1 gut mcmc noise_lr 0"; timeout 200 python tools/r4_scale_debug.py -d /tmp/syn_colmap --gut --strategy mcmc --noise-lr 0 -i 12500 --every 2500 2>&1 | grep iter | sed "$F" | cut -c1-330
echo "== E4 gut mcmc det"; timeout 200 python tools/r4_scale_debug.py -d /tmp/syn_colmap --gut --strategy mcmc --det -i 12500 --every 1250 2>&1 | grep iter | sed "$F" | cut -c1-330
