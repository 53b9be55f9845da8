"""Development aid: step time of the tiled renderer in a process that also holds an RCCL communicator
(world size 1), created before or after the renderer."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from raytracer_amd import renderer as R, workloads

order = sys.argv[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
R.load()
scene = workloads.dragon_cornell()


def communicator():
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    x = torch.ones(4, device="cuda")
    dist.all_reduce(x)
    dist.barrier()
    torch.cuda.synchronize()


if order == "before":
    communicator()
r = R.Renderer(scene, 1920, 1080, 0)
for _ in range(3):
    r.render()
r.synchronize()
if order == "after":
    communicator()
for _ in range(3):
    r.render()
r.synchronize()
t0 = time.perf_counter()
for _ in range(24):
    r.render()
r.synchronize()
print(f"communicator {order} the renderer: {(time.perf_counter() - t0) / 24 * 1e3:.3f} ms", flush=True)
dist.destroy_process_group()
