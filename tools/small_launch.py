"""How long does a ray launch with few rays take?  Rays from the centre of the glass ball of the
caustic-lab scene in random directions (the late bounces of the final profile are of this kind), in
launches of 1 to 65536 rays: event-timed, after a warm-up launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

from raytracer_amd import engine, workloads
from helpers import engine_scene

engine.load()
top, blases = engine_scene(engine, workloads.caustic_lab())
rng = np.random.default_rng(1)
for n in (1, 64, 1024, 4096, 16384, 65536):
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    org = np.tile(np.array([0.3, 0.15, 0.05, 1e-3], np.float32), (n, 1))
    dirs = np.concatenate([d, np.full((n, 1), 1e16)], 1).astype(np.float32)
    o, dd = torch.from_numpy(org).cuda(), torch.from_numpy(dirs).cuda()
    hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    inst = torch.empty(n, dtype=torch.int32, device="cuda")
    counters = torch.zeros(4, dtype=torch.int64, device="cuda")
    top.trace_closest(o, dd, hits=hits, inst=inst, counters=counters)
    torch.cuda.synchronize()
    c = counters.cpu().numpy()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(20):
        top.trace_closest(o, dd, hits=hits, inst=inst)
    stop.record()
    torch.cuda.synchronize()
    print(f"{n:6d} rays: {start.elapsed_time(stop) / 20 * 1e3:7.1f} us per launch; per ray {c[0] / n:.1f} nodes, "
          f"{c[1] / n:.1f} triangles, {c[2] / n:.1f} instances")
