"""Development aid: does a ray launch end sooner when the rays that start on the large mesh are handed out first?
Bounce rays of the 1080p dragon frame (origins = the camera rays' hit points, directions uniform) are traced in
pixel order, with the mesh's rays first in each of the feed's eight chunks, and with them last."""
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
hits, inst = top.trace_closest(o, d)
found = hits[:, 3].view(torch.int32) != -1
t = hits[:, 0:1]
origin = o[:, :3] + d[:, :3] * t
gen = torch.Generator(device="cuda").manual_seed(1)
direction = torch.randn((o.shape[0], 3), generator=gen, device="cuda")
direction /= direction.norm(dim=1, keepdim=True)
origin = origin + direction * 1e-3
bo = torch.cat([origin, torch.full((o.shape[0], 1), 0.001, device="cuda")], dim=1)[found].contiguous()
bd = torch.cat([direction, torch.full((o.shape[0], 1), 1e16, device="cuda")], dim=1)[found].contiguous()
binst = inst[found]
sizes = [len(g["indices"]) for g in scene["geometries"]]
big = int(np.argmax([sizes[g] for g, _ in scene["instances"]]))
slow = binst == big
n = bo.shape[0]
print(f"{n} bounce rays, {int(slow.sum())} start on the mesh (instance {big})")


def ms(oo, dd, reps=5):
    h, i = top.trace_closest(oo, dd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        top.trace_closest(oo, dd, hits=h, inst=i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def per_chunk(order_key):
    """Permutation that sorts every one of the feed's eight chunks by order_key (stable)."""
    chunk = ((n + 7) // 8 + 63) & ~63
    idx = torch.arange(n, device="cuda")
    key = (idx // chunk) * 4 + order_key
    return torch.sort(key, stable=True).indices


def interleaved(order_key):
    """Slow rays first over the whole batch, then dealt to the eight chunks 64 at a time."""
    first = torch.sort(order_key, stable=True).indices
    chunk = ((n + 7) // 8 + 63) & ~63
    pos = torch.arange(n, device="cuda")
    granule, lane = pos // 64, pos % 64
    target = (granule % 8) * chunk + (granule // 8) * 64 + lane
    ok = target < n
    out = torch.empty(n, dtype=torch.long, device="cuda")
    out[:] = -1
    out[target[ok]] = first[ok]
    rest = first[~ok]
    out[out < 0] = rest
    return out


print(f"pixel order: {ms(bo, bd):.3f} ms")
for name, key in (("mesh rays first in each chunk", (~slow).long()), ("mesh rays last in each chunk", slow.long())):
    p = per_chunk(key)
    print(f"{name}: {ms(bo[p].contiguous(), bd[p].contiguous()):.3f} ms")
p = interleaved((~slow).long())
print(f"mesh rays first, dealt to the chunks in turn: {ms(bo[p].contiguous(), bd[p].contiguous()):.3f} ms")
p = torch.randperm(n, device="cuda")
print(f"random order: {ms(bo[p].contiguous(), bd[p].contiguous()):.3f} ms")
print(f"mesh rays only: {ms(bo[slow].contiguous(), bd[slow].contiguous()):.3f} ms; the others only: {ms(bo[~slow].contiguous(), bd[~slow].contiguous()):.3f} ms")
for share in (2, 4, 8):
    print(f"every {share}th ray, pixel order: {ms(bo[::share].contiguous(), bd[::share].contiguous()):.3f} ms")
for name, mask in (("mesh rays", slow), ("the others", ~slow)):
    c = torch.zeros(4, dtype=torch.int64, device="cuda")
    top.trace_closest(bo[mask].contiguous(), bd[mask].contiguous(), counters=c)
    c = c.cpu().numpy()
    print(f"{name}: {c[1] / c[0]:.1f} nodes, {c[2] / c[0]:.1f} triangles, {c[3] / c[0]:.2f} instances per ray")
# distribution over single rays of the mesh class
idx = torch.nonzero(slow).flatten()[:: max(1, int(slow.sum()) // 1500)]
c = torch.zeros(4, dtype=torch.int64, device="cuda")
v = []
for i in idx.tolist():
    c.zero_()
    top.trace_closest(bo[i:i + 1].contiguous(), bd[i:i + 1].contiguous(), counters=c)
    x = c.cpu().numpy()
    v.append((int(x[1]), int(x[2])))
v = np.array(v)
tot = v.sum(axis=1)
print("mesh rays, per ray: nodes mean %.1f 50%% %d 90%% %d 99%% %d max %d; triangles mean %.1f 90%% %d 99%% %d max %d; both max %d" % (
    v[:, 0].mean(), np.percentile(v[:, 0], 50), np.percentile(v[:, 0], 90), np.percentile(v[:, 0], 99), v[:, 0].max(),
    v[:, 1].mean(), np.percentile(v[:, 1], 90), np.percentile(v[:, 1], 99), v[:, 1].max(), tot.max()))
