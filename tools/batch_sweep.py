"""Development aid: the whole 1080p dragon frame against samples per pass x passes in flight x parts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import renderer as R, workloads
R.load()
scene = workloads.dragon_cornell()
r = R.Renderer(scene, 1920, 1080, 0)
for batch, depth, parts in ((8, 4, 1), (16, 2, 1), (16, 4, 1), (32, 1, 1), (32, 2, 1), (8, 2, 2), (8, 3, 1), (8, 6, 1), (12, 4, 1), (16, 3, 1)):
    r.set_concurrent_tiles(parts, depth)
    r.set_sample_batch(batch)
    for _ in range(2 * batch * depth):
        r.render()
    r.synchronize()
    steps = 4 * batch * depth
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render()
    r.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{batch} samples per pass, {depth} in flight, {parts} parts: {ms:.3f} ms per sample ({batch * depth * parts * 2.07e6 * 328 / 2**30 / parts:.1f} GiB)", flush=True)
