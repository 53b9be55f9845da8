"""Development aid: renders one rank's share of the 1080p dragon frame (every N-th band) for a kernel trace.
    rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/share_trace.py 8 [parts] [in flight]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import renderer as R, workloads
R.load()
share = int(sys.argv[1]) if len(sys.argv) > 1 else 8
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 1
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 4
r = R.Renderer(workloads.dragon_cornell(), 1920, 1080, 0)
r.set_concurrent_tiles(parts, depth)
r.set_row_interleave(0, share)
for _ in range(8):
    r.render()
r.synchronize()
t0 = time.perf_counter()
for _ in range(32):
    r.render()
r.synchronize()
print(f"1/{share} share, {parts} x {depth}: {(time.perf_counter() - t0) / 32 * 1e3:.3f} ms per step", flush=True)
