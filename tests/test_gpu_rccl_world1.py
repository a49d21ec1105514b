"""GPU: the RCCL code path on ONE GPU. RCCL refuses two ranks per device, so the multi-rank GPU tests (test_gpu_dp2.py) stage through gloo;
here backend "nccl" (= RCCL on ROCm) is initialised with world size 1 and LFS_DIST_FORCE_COLLECTIVES=1 makes dist.py issue every collective
anyway: device tensors go through GradBucket.all_reduce / all_reduce_early (async, RCCL's own stream), ShExchange._all_to_all / gather_rows,
all_reduce_sum, barrier and max_over_ranks exactly as they would on 8 GPUs - and, with one rank, must return their inputs. A subprocess,
because the flag and the process group are per process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
import lichtfeld_studio_amd
from lichtfeld_studio_amd import dist as ld, scenes
from lichtfeld_studio_amd.trainer import GutTrainer
rank, world, local = ld.init_distributed()
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
ld.stats_enable(True)
assert ld.ranks_seen(dev) == 1
g = torch.Generator(device=dev).manual_seed(0)
N = 100003
params = [torch.zeros(N, 3, device=dev), torch.zeros(N, 1, 3, device=dev), torch.zeros(N, 15, 3, device=dev), torch.zeros(N, 3, device=dev), torch.zeros(N, 4, device=dev), torch.zeros(N, device=dev)]
b = ld.GradBucket(params, deferred=[2])
grads = [torch.randn(p.shape, device=dev, generator=g) for p in params]
b.gather(grads)
b.all_reduce_early([3, 4, 5]); b.all_reduce(skip_deferred=False)
torch.cuda.synchronize()
assert all(torch.equal(v, gr) for v, gr in zip(b.views, grads))
ex = ld.ShExchange(N, 1, 0)
x = torch.randn(1, ex.S, 3, device=dev, generator=g)
assert torch.equal(ex._all_to_all(x), x)
rows = torch.randn(ex.n, 15, 3, device=dev, generator=g)
assert torch.equal(ex.gather_rows(rows), rows)
t = torch.randn(1000, device=dev, generator=g); t0 = t.clone(); ld.all_reduce_sum(t); assert torch.equal(t, t0)
ld.barrier(); assert ld.max_over_ranks(3.5, dev) == 3.5
# a training step with the collectives in the loop (SH-sharded layout, the multi-GPU default): same parameters as without them
sc = scenes.syn_a(n=3000, sh_degree=2)
tg = [scenes.target_image(sc.height, sc.width).to(dev)]
a = GutTrainer(sc, dev, iterations=100, world=1, rank=0, sh_sharded=True)
c = GutTrainer(sc, dev, iterations=100, world=1, rank=0, sh_sharded=False)
a.inline_shN_adam = c.inline_shN_adam = False
for tr in (a, c):
    tr.iteration = 1500
    for _ in range(3): tr.train_step(tg, views_all=[[0]]) if tr.sh_exchange is not None else tr.train_step(tg, views=[0])
torch.cuda.synchronize()
for pa, pc in zip(a.model.parameters(), c.model.parameters()):
    assert float((pa - pc).abs().max()) < 5e-3
st = ld.stats_collect()
print("RCCL_OK", {k: (v["calls"], v["bytes"], v["ms"]) for k, v in st.items()})
assert st["all_reduce"]["calls"] >= 3 and st["all_to_all"]["calls"] >= 9 and st["all_reduce_early"]["calls"] >= 1
dist.destroy_process_group()
"""


def test_rccl_collectives_execute_on_one_gpu():
    env = dict(os.environ, LFS_DIST_FORCE_COLLECTIVES="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29631",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    print(r.stdout[-600:])
