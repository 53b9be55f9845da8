"""Development aid: regions x samples in flight on the full frame of a workload (dragon-cornell, caustic-lab, sphere-field)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracer_amd import renderer as R, workloads
R.load()
name = sys.argv[1] if len(sys.argv) > 1 else "dragon-cornell"
scene, profile, size = {"dragon-cornell": (workloads.dragon_cornell(), 0, (1920, 1080)),
                        "caustic-lab": (workloads.caustic_lab(), 1, (1920, 1080)),
                        "sphere-field": (workloads.sphere_field(sky=True), 0, (3840, 2160))}[name]
r = R.Renderer(scene, size[0], size[1], profile)
reference = None
for tiles, depth in ((3, 1), (1, 3), (1, 4), (1, 5), (1, 6), (1, 8), (2, 2), (2, 3), (2, 4)):
    r.set_concurrent_tiles(tiles, depth)
    for _ in range(8):
        r.render()
    r.synchronize()
    image = r.linear_pixels().copy()
    if reference is None:
        reference = image
    same = np.array_equal(image.view(np.uint32), reference.view(np.uint32))
    steps = 48
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render()
    r.synchronize()
    t1 = time.perf_counter()
    print(f"{name}: {tiles} parts x {depth} in flight: {(t1 - t0) / steps * 1e3:.3f} ms per step, image after 8 samples "
          f"{'identical' if same else 'DIFFERENT'}", flush=True)
