"""Development aid: step time of the dragon stand-in over the number of concurrent parts of the frame and
the share of the wave slots a ray launch of one part takes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from raytracer_amd import renderer as R, workloads


def main():
    R.load()
    scene = workloads.dragon_cornell()
    r = R.Renderer(scene, 1920, 1080, 0)
    for tiles in (1, 2, 3, 4, 5, 6):
        for share in ((1, 1), (3, 4), (2, 3), (1, 2), (2, 5), (1, 3), (1, 4)):
            r.set_concurrent_tiles(tiles)
            r.set_ray_share(*share)
            for _ in range(4):
                r.render()
            r.synchronize()
            t0 = time.perf_counter()
            for _ in range(24):
                r.render()
            r.synchronize()
            print(f"tiles {tiles} share {share[0]}/{share[1]}: {(time.perf_counter() - t0) / 24 * 1e3:6.3f} ms", flush=True)


main()
