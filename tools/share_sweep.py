"""Development aid: what one rank of an N-GPU job does per step - every N-th band of the 1080p dragon frame -
timed on one GPU for N = 1, 2, 4, 8 and 1 to 3 concurrent parts; also the host's time to enqueue a step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import renderer as R, workloads
R.load()
scene = workloads.dragon_cornell()
r = R.Renderer(scene, 1920, 1080, 0)
for share in (1, 4, 8):
    for tiles, depth in ((1, 4), (1, 6), (1, 8), (1, 12)):
        r.set_concurrent_tiles(tiles, depth)
        r.set_row_interleave(0, share)
        for _ in range(8):
            r.render()
        r.synchronize()
        t0 = time.perf_counter()
        for _ in range(48):
            r.render()
        t1 = time.perf_counter()
        r.synchronize()
        t2 = time.perf_counter()
        print(f"1/{share} of the frame, {tiles} parts x {depth} samples in flight: {(t2 - t0) / 48 * 1e3:.3f} ms per step, host enqueue {(t1 - t0) / 48 * 1e3:.3f} ms", flush=True)
