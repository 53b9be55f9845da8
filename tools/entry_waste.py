"""Development aid: how many instance entries of a batch of rays end without a hit inside the instance entered
(rays that cross an instance's box and miss its mesh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from raytracer_amd import engine, workloads
from helpers import engine_scene

engine.load()
scene = workloads.dragon_cornell()
top, blases = engine_scene(engine, scene)
folded = top.folded()
for name, (o, d) in (("camera", workloads.camera_rays(scene, 1920, 1080)), ("random", workloads.random_rays(scene, 2000000, seed=3))):
    o, d = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    c = torch.zeros(4, dtype=torch.int64, device="cuda")
    hits, inst = top.trace_closest(o, d, counters=c)
    c = c.cpu().numpy()
    tube = int((inst == 4).sum())
    print(f"{name}: rays {c[0]}, nodes/ray {c[1] / c[0]:.2f}, triangles/ray {c[2] / c[0]:.2f}, entries/ray {c[3] / c[0]:.3f}, "
          f"closest hit on the tube {tube / c[0]:.3f}, folded {folded}")
