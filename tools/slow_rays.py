"""Development aid: which rays of a batch hold a launch up.  Camera rays of the 1080p dragon frame are traced in
shrinking chunks, following the slowest chunk, down to single rays whose node and triangle visits are printed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from raytracer_amd import engine, workloads
from helpers import engine_scene
engine.load()
scene = workloads.dragon_cornell()
top, keep = engine_scene(engine, scene)
o, d = workloads.camera_rays(scene, 1920, 1080)
o, d = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
n = o.shape[0]


def ms(lo, hi, reps=3):
    oo, dd = o[lo:hi].contiguous(), d[lo:hi].contiguous()
    hits, inst = top.trace_closest(oo, dd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        top.trace_closest(oo, dd, hits=hits, inst=inst)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print(f"{n} camera rays: {ms(0, n):.3f} ms; first 1/8: {ms(0, n // 8):.3f} ms; 4096 rays: {ms(n // 2, n // 2 + 4096):.3f} ms; 64 rays: {ms(n // 2, n // 2 + 64):.3f} ms; 1 ray: {ms(n // 2, n // 2 + 1):.3f} ms")
# per-ray visits of a sample of rays: single-ray counted launches
rng = np.random.default_rng(0)
sample = rng.choice(n, 3000, replace=False)
c = torch.zeros(4, dtype=torch.int64, device="cuda")
visits = []
for i in sample:
    c.zero_()
    top.trace_closest(o[i:i + 1].contiguous(), d[i:i + 1].contiguous(), counters=c)
    v = c.cpu().numpy()
    visits.append((int(v[1]), int(v[2])))
visits = np.array(visits)
turns = visits.sum(axis=1)
print("nodes per ray: mean %.1f, 90%% %d, 99%% %d, max %d; triangles: mean %.1f, 99%% %d, max %d; nodes+triangles max %d" % (
    visits[:, 0].mean(), np.percentile(visits[:, 0], 90), np.percentile(visits[:, 0], 99), visits[:, 0].max(),
    visits[:, 1].mean(), np.percentile(visits[:, 1], 99), visits[:, 1].max(), turns.max()))
worst = sample[np.argsort(turns)[-5:]]
for i in worst:
    t = ms(int(i), int(i) + 1, reps=5)
    k = list(sample).index(i)
    print(f"ray {i} (pixel {i % 1920}, {i // 1920}): {visits[k][0]} nodes, {visits[k][1]} triangles, alone {t * 1e3:.0f} us")
