"""Development aid: how evenly the rays of one 1080p dragon-cornell sample fall on N GPUs when the frame is cut
into contiguous blocks of rows and when bands of eight rows are dealt out in turn.  Largest share / mean share, by rays and by time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import renderer as R, workloads
R.load()
W, H = 1920, 1080
scene = workloads.dragon_cornell()
r = R.Renderer(scene, W, H, 0)
import time


def rays():
    """(rays per sample, ms per sample) of the share now set, four samples in flight."""
    for _ in range(8):
        r.render()
    r.statistics()
    t0 = time.perf_counter()
    for _ in range(32):
        r.render()
    c = r.statistics()  # waits for the GPU
    ms = (time.perf_counter() - t0) / 32 * 1e3
    return (c["extend_rays"] + c["shadow_rays"]) / 32, ms


for world in (2, 4, 8):
    blocks, bands = [], []
    for rank in range(world):
        r.set_row_interleave(0, 1)
        r.set_pixel_range(H * rank // world * W, H * (rank + 1) // world * W)
        blocks.append(rays())
        r.set_pixel_range(0, W * H)
        r.set_row_interleave(rank, world)
        bands.append(rays())
    for name, per in (("blocks of rows", blocks), ("bands in turn", bands)):
        count = [p[0] for p in per]
        ms = [p[1] for p in per]
        mean, mean_ms = sum(count) / world, sum(ms) / world
        print(f"{world} GPUs, {name}: rays largest {max(count) / mean:.3f} x mean, smallest {min(count) / mean:.3f} x mean; "
              f"time largest {max(ms):.3f} ms = {max(ms) / mean_ms:.3f} x mean, smallest {min(ms):.3f} ms", flush=True)
