"""Development aid: what one of N GPUs would spend per sample on its share of the 1080p dragon frame (every N-th
band of eight rows) at the renderer's defaults - the single-GPU rehearsal of bench.py --gpus N."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import renderer as R, workloads
R.load()
scene = workloads.dragon_cornell()
first = None
for share in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    r = R.Renderer(scene, 1920, 1080, 0)
    r.set_row_interleave(0, share)
    for _ in range(64):
        r.render()
    r.synchronize()
    steps = 128
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render()
    r.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    first = first or ms
    print(f"1/{share} of the bands, passes of {r.sample_batch()} samples, {r.samples_in_flight()} in flight: {ms:.3f} ms per sample, "
          f"efficiency {first / (ms * share):.2f}", flush=True)
    del r
