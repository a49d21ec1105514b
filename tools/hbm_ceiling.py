"""Practical HBM ceilings of the box ("confirm with a copy / triad microbench and report that as the practical ceiling"):
device-to-device copy (read + write), fill (write only), reduction (read only) of 2 GiB buffers, and the Adam kernel (16 B read + 12 B write)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lichtfeld_studio_amd  # noqa
from lichtfeld_studio_amd import ops
dev = "cuda:0"
n = 512 * 1024 * 1024            # floats: 2 GiB
a, b = torch.rand(n, device=dev), torch.empty(n, device=dev)
def t(fn, reps=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / reps * 1e-3
res = {"buffer_GiB": n * 4 / 2**30}
res["copy_GBps"] = round(2 * n * 4 / t(lambda: b.copy_(a)) / 1e9, 1)
res["fill_GBps"] = round(n * 4 / t(lambda: b.fill_(1.0)) / 1e9, 1)
res["sum_GBps"] = round(n * 4 / t(lambda: a.sum()) / 1e9, 1)
m = 128 * 1024 * 1024
p, ea, es, g = (torch.rand(m, device=dev) for _ in range(4))
res["adam_GBps"] = round(28 * m / t(lambda: ops.adam_step_wrapper(p, ea, es, g, 1e-3, 0.9, 0.999, 1e-15, 1.0, 1.0)) / 1e9, 1)
res["spec_GBps"] = 8000.0
print(json.dumps(res))
