import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import lichtfeld_studio_amd as lfs
from lichtfeld_studio_amd import dist as lfs_dist, scenes
from lichtfeld_studio_amd.trainer import GutTrainer
rank, world, local_rank = lfs_dist.init_distributed()
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
sc = scenes.syn_b()
tr = GutTrainer(sc, dev, iterations=7000, world=world, rank=rank, sh_sharded=True)
tr.iteration = int(os.environ.get("START_IT", "3000"))
tg = [scenes.target_image(sc.height, sc.width, seed=43).to(dev)]
for i in range(30):
    tr.train_step(tg)
    if os.environ.get("SYNC_EACH"): torch.cuda.synchronize(); print("step", i, "ok", flush=True)
torch.cuda.synchronize()
print("done", flush=True)
torch.distributed.destroy_process_group()
