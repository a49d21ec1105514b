"""Developer: HOST time of the SH-sharded step with forced collectives on one GPU (world 1, RCCL): does the Python / c10d side keep ahead of the GPU?
   LFS_DIST_FORCE_COLLECTIVES=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 python tools/profile_host_sharded.py"""
import cProfile, pstats, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import lichtfeld_studio_amd as lfs
from lichtfeld_studio_amd import dist as lfs_dist, scenes
from lichtfeld_studio_amd.trainer import GutTrainer
rank, world, local_rank = lfs_dist.init_distributed()
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
sc = scenes.syn_b()
tr = GutTrainer(sc, dev, iterations=7000, world=world, rank=rank, sh_sharded=True)
tr.iteration = 3000
tg = [scenes.target_image(sc.height, sc.width, seed=43).to(dev)]
for _ in range(10): tr.train_step(tg)
torch.cuda.synchronize()
# host-only time: enqueue 100 steps without waiting for the GPU in between (the n_isects read is the one wait per step)
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for _ in range(100): tr.train_step(tg)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
pr.disable()
print("ms/step wall", (time.perf_counter() - t0) / 100 * 1e3, " host loop returned after", t_host / 100 * 1e3, "ms/step")
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
torch.distributed.destroy_process_group()
