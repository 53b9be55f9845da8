"""Development aid: per-stage event times of one rank's share of the 1080p dragon frame, one sample at a time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import renderer as R, workloads
R.load()
r = R.Renderer(workloads.dragon_cornell(), 1920, 1080, 0)
for share in (1, 2, 4, 8):
    r.set_concurrent_tiles(1, 1)
    r.set_row_interleave(0, share)
    for timed in (False, True):
        r.set_measurement(time_stages=timed)
        for _ in range(4):
            r.render()
        r.synchronize()
        if timed:
            r.stage_timings()
        t0 = time.perf_counter()
        for _ in range(16):
            r.render()
        r.synchronize()
        ms = (time.perf_counter() - t0) / 16 * 1e3
        if timed:
            t = r.stage_timings()
            print(f"1/{share}: {ms:.3f} ms per step with events; stages per step:", {k: (round(v[0] / 16, 3), v[1] // 16) for k, v in t.items()}, flush=True)
        else:
            print(f"1/{share}: {ms:.3f} ms per step, one sample at a time", flush=True)
