"""Development aid: the whole 1080p dragon frame, samples per pass x passes in flight x share of the wave slots a ray launch takes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import renderer as R, workloads
R.load()
scene = workloads.dragon_cornell()
r = R.Renderer(scene, 1920, 1080, 0)
takers = int(sys.argv[1]) if len(sys.argv) > 1 else 1
print(f"every {takers}th band of the frame")
for batch, depth, share in ((8, 4, (1, 2)), (8, 4, (1, 1)), (8, 4, (3, 4)), (8, 4, (1, 3)), (16, 2, (1, 1)), (16, 2, (1, 2)), (16, 1, (1, 1)), (32, 1, (1, 1)), (64, 1, (1, 1)), (8, 2, (1, 1)), (4, 4, (1, 1)), (16, 4, (1, 1)), (32, 2, (1, 1))):
    r.set_concurrent_tiles(1, depth)
    r.set_sample_batch(batch)
    r.set_row_interleave(0, takers)
    r.set_ray_share(*share)
    for _ in range(2 * batch * depth):
        r.render()
    r.synchronize()
    steps = max(64, 4 * batch * depth)
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render()
    r.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{batch} samples per pass, {depth} in flight, share {share[0]}/{share[1]}: {ms:.3f} ms per sample", flush=True)
