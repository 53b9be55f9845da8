"""Development aid: share of the wave slots per ray launch with four samples in flight (dragon-cornell, 1080p)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import renderer as R, workloads
R.load()
name = sys.argv[1] if len(sys.argv) > 1 else "dragon-cornell"
scene, profile = (workloads.caustic_lab(), 1) if name == "caustic-lab" else (workloads.dragon_cornell(), 0)
r = R.Renderer(scene, 1920, 1080, profile)
for depth in (3, 4):
    for share in ((1, 1), (3, 4), (1, 2), (1, 3), (1, 4)):
        r.set_concurrent_tiles(1, depth)
        r.set_ray_share(*share)
        for _ in range(8):
            r.render()
        r.synchronize()
        t0 = time.perf_counter()
        for _ in range(48):
            r.render()
        r.synchronize()
        print(f"{name}: {depth} in flight, share {share[0]}/{share[1]}: {(time.perf_counter() - t0) / 48 * 1e3:.3f} ms per step", flush=True)
