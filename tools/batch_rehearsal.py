"""Development aid: one rank's share of the 1080p dragon frame (every N-th band) against the number of samples a
pass takes (set_sample_batch) and the number of passes in flight.  ms per sample."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import renderer as R, workloads
R.load()
scene = workloads.dragon_cornell()
r = R.Renderer(scene, 1920, 1080, 0)
shares = [int(a) for a in sys.argv[1:]] or [8, 4, 2, 1]
for share in shares:
    for batch, depth in ((1, 4), (2, 2), (2, 4), (4, 1), (4, 2), (4, 4), (8, 2), (8, 4)):
        r.set_concurrent_tiles(1, depth)
        r.set_sample_batch(batch)
        r.set_row_interleave(0, share)
        for _ in range(2 * batch * depth):
            r.render()
        r.synchronize()
        steps = 64
        t0 = time.perf_counter()
        for _ in range(steps):
            r.render()
        r.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        print(f"1/{share} share, passes of {batch} samples, {depth} in flight: {ms:.3f} ms per sample", flush=True)
