"""Development aid: step time of the tiled renderer in a process that owns other HIP streams, created
before or after the renderer (python tools/streams_test.py <extra streams> before|after): shows whether
the renderer's streams ended up on hardware queues of their own."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import renderer as R, workloads
extra = int(sys.argv[1]); when = sys.argv[2]
R.load()
scene = workloads.dragon_cornell()
keep = []
def make():
    for _ in range(extra):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            x = torch.zeros(16, device="cuda") + 1
        keep.append((s, x))
    torch.cuda.synchronize()
if when == "before":
    make()
r = R.Renderer(scene, 1920, 1080, 0)
if when == "after":
    make()
for _ in range(4):
    r.render()
r.synchronize()
t0 = time.perf_counter()
for _ in range(24):
    r.render()
r.synchronize()
print(f"extra streams {extra} created {when}: {(time.perf_counter() - t0) / 24 * 1e3:.3f} ms")
