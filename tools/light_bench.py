"""Development aid: 1080p steps of the Cornell box lit by two emissive icospheres (40,960 lights) with the
light index and with the walks over the light list (RT_LIGHT_INDEX=0), and the time to build the index."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import renderer as R, workloads, engine
R.load()
scene = workloads.glowing_spheres_room()
for choice in ("1", "0"):
    os.environ["RT_LIGHT_INDEX"] = choice
    r = R.Renderer(scene, 1920, 1080, 0)
    steps = 16 if choice == "1" else 2
    for _ in range(2):
        r.render()
    r.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render()
    r.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    s = r.statistics()
    print(f"light index {'on' if choice == '1' else 'off'}: {ms:.2f} ms per step, {r.launch_params().lightCount} lights", flush=True)
    del r
engine.load()
for n in (1 << 12, 1 << 15, 1 << 18):
    lights = torch.zeros(n * 92, dtype=torch.uint8, device="cuda")
    lights.view(torch.float32).view(-1, 23)[:, 16] = torch.rand(n, device="cuda") + 0.5
    lights.view(torch.int32).view(-1, 23)[:, 22] = 3
    lights.view(torch.int32).view(-1, 23)[:, 21] = torch.arange(n, device="cuda", dtype=torch.int32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    index = engine.LightIndex(lights)
    torch.cuda.synchronize()
    print(f"index of {n} lights built in {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
