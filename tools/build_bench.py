"""Development aid: BVH build time and traversal cost (nodes / triangles per ray, Mrays/s) of the two
bottom-level builders on a few meshes.  RT_BLAS_BUILDER is read at build time, so each builder runs in
this one process by switching the variable between builds."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from raytracer_amd import engine, workloads
from helpers import engine_scene


def one(name, scene, builder, rays=1 << 21):
    os.environ["RT_BLAS_BUILDER"] = "ploc" if builder == "ploc" else "radix"
    os.environ["RT_BLAS_REFINE"] = "0" if builder == "radix" else "1"  # "refined" = radix tree + surface-area upper levels
    print(f"{name}: building", flush=True)
    top, blases = engine_scene(engine, scene)
    print(f"{name}: built, generating rays", flush=True)
    build = sum(b.stats().buildMilliseconds for b in blases)
    nodes = sum(b.stats().nodeCount for b in blases)
    depth = max(getattr(b.stats(), "depth", -1) for b in blases)
    o, d = workloads.random_rays(scene, rays)
    o, d = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
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
    print(f"{name:14s} {builder:8s} build {build:8.2f} ms  nodes {nodes:8d} depth {depth:3d}  nodes/ray {c[1] / c[0]:6.2f}  "
          f"tris/ray {c[2] / c[0]:6.2f}  {rays / dt / 1e6:8.1f} Mrays/s", flush=True)


def main():
    engine.load()
    if len(sys.argv) > 1:  # soup sizes on the command line, default builder only: the large-scene measurements
        for n in (int(a) for a in sys.argv[1:]):
            print(f"soup {n}: generating triangles", flush=True)
            for builder in ("radix", "refined"):
                one(f"soup {n}", workloads.triangle_soup(n), builder, rays=1 << 24)
        return
    cases = [("tube 871k", workloads.dragon_cornell()), ("soup 100k", workloads.triangle_soup(100000)),
             ("soup 1M", workloads.triangle_soup(1000000)), ("spheres", workloads.sphere_field())]
    for name, scene in cases:
        for builder in ("radix", "refined", "ploc"):
            one(name, scene, builder)


main()
