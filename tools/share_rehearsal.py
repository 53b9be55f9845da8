"""Development aid: what one of N GPUs would spend per sample on its share of a 1080p frame (every N-th band of
eight rows) at the renderer's defaults - the single-GPU rehearsal of bench.py --gpus N.

    python tools/share_rehearsal.py [--workload dragon-cornell|caustic-lab] [--batch B] [--in-flight D] [shares ...]

caustic-lab (final profile, photon map at the reference's sizes): the share also traces its 1/N of the photon
emissions before every fourth sample, as raytracer_amd.distributed does, and then builds the grid over a full
photon list (the one a complete pass left in the photon buffer; what the peers would have sent).  The exchange
itself is not rehearsed."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import renderer as R, workloads

ap = argparse.ArgumentParser()
ap.add_argument("shares", type=int, nargs="*", default=[1, 2, 4, 8])
ap.add_argument("--workload", default="dragon-cornell", choices=["dragon-cornell", "caustic-lab"])
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--in-flight", type=int, default=0)
ap.add_argument("--steps", type=int, default=128)
ap.add_argument("--warmup", type=int, default=64)
ap.add_argument("--ray-share", default="", help="N/D: the share of the wave slots a ray launch takes")
args = ap.parse_args()
R.load()
final = args.workload == "caustic-lab"
scene = workloads.caustic_lab() if final else workloads.dragon_cornell()
first = None
for share in args.shares:
    r = R.Renderer(scene, 1920, 1080, R.FINAL if final else R.PREVIEW)
    if final:
        r.set_photon_budget(1 << 24, 1 << 23, 1 << 22)
    if args.batch:
        r.set_sample_batch(args.batch)
    if args.in_flight:
        r.set_concurrent_tiles(1, args.in_flight)
    if args.ray_share:
        r.set_ray_share(*(int(x) for x in args.ray_share.split("/")))
    r.set_row_interleave(0, share)
    full = []
    if final and share > 1:
        r.set_manual_photon_passes(True)
        passes = r.begin_photon_passes(r.photon_passes_together())
        # the merged lists of the passes that run together, left in their photon buffers
        full = r.trace_photon_range_passes(0, r.photon_emissions(), passes)
        r.finish_photon_passes(full)
        together = passes

    def step():
        if final and share > 1 and r.photon_pass_due():
            passes = r.begin_photon_passes(together)
            r.trace_photon_range_passes(0, r.photon_emissions() // share, passes, staging=True)
            r.finish_photon_passes(full[:passes])
        r.render()

    for _ in range(args.warmup):
        step()
    r.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    r.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    first = first or ms
    print(f"{args.workload} 1/{share} of the bands, passes of {r.sample_batch()} samples, {r.samples_in_flight()} in flight"
          + (f", {r.photon_passes_together()} photon passes together" if final else "") + ": "
          f"{ms:.3f} ms per sample, efficiency {first / (ms * share):.2f}", flush=True)
    del r
