"""Development aid: traversal cost of the tube mesh against the number of clusters the upper levels are rebuilt over."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from raytracer_amd import engine, workloads
from helpers import engine_scene
engine.load()
scene = workloads.dragon_cornell()
o, d = workloads.random_rays(scene, 1 << 21)
o, d = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
for clusters in (0, 2048, 8192, 32768, 131072, 400000):
    os.environ["RT_BLAS_REFINE"] = "0" if clusters == 0 else "1"
    os.environ["RT_BLAS_CLUSTERS"] = str(max(clusters, 2))
    top, blases = engine_scene(engine, scene)
    build = sum(b.stats().buildMilliseconds for b in blases)
    counters = torch.zeros(4, dtype=torch.int64, device="cuda")
    top.trace_closest(o, d, counters=counters)
    c = counters.cpu().numpy()
    hits, inst = top.trace_closest(o, d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        top.trace_closest(o, d, hits=hits, inst=inst)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"clusters {clusters:7d}  build {build:7.2f} ms  nodes/ray {c[1]/c[0]:.3f} tris/ray {c[2]/c[0]:.3f}  {o.shape[0]/dt/1e6:.0f} Mrays/s", flush=True)
