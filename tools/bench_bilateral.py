"""Times the bilateral-grid kernels at 1080p (16x16x8 grid, the reference defaults) with HIP events. GPU only."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lichtfeld_studio_amd  # noqa
from lichtfeld_studio_amd import bilateral_grid as bg

dev = "cuda:0"
h, w = 1080, 1920
grid = torch.randn(12, 8, 16, 16, device=dev)
grids = torch.randn(200, 12, 8, 16, 16, device=dev)
res = {}
for name, chw in (("hwc", False), ("chw", True)):
    rgb = torch.rand((3, h, w) if chw else (h, w, 3), device=dev)
    go = torch.randn_like(rgb)
    gg = torch.zeros_like(grid)
    for fn, label in ((lambda: bg.slice_forward(grid, rgb, chw=chw, clamp_input=True), "slice_fwd"),
                      (lambda: bg.slice_backward(grid, rgb, go, chw=chw, clamp_input=True, grad_grid=gg), "slice_bwd")):
        for _ in range(5):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            fn()
        b.record(); torch.cuda.synchronize()
        res[f"{label}_{name}_ms"] = a.elapsed_time(b) / 50
acc = torch.zeros(1, device=dev)
gacc = torch.zeros_like(grids)
for fn, label in ((lambda: bg.tv_loss_forward(grids, 1.0, acc), "tv_fwd"), (lambda: bg.tv_loss_backward(grids, 1.0, gacc), "tv_bwd")):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        fn()
    b.record(); torch.cuda.synchronize()
    res[f"{label}_200grids_ms"] = a.elapsed_time(b) / 50
# algorithmic HBM bytes: slice = image in + out (+ grad in/out) ; tv = grids once (fwd), grids + grad rw (bwd)
res["slice_fwd_GBps_hwc"] = 2 * h * w * 12 / res["slice_fwd_hwc_ms"] / 1e6
res["slice_bwd_GBps_hwc"] = 3 * h * w * 12 / res["slice_bwd_hwc_ms"] / 1e6
res["tv_fwd_GBps"] = grids.numel() * 4 / res["tv_fwd_200grids_ms"] / 1e6
res["tv_bwd_GBps"] = 3 * grids.numel() * 4 / res["tv_bwd_200grids_ms"] / 1e6
print(json.dumps(res))
