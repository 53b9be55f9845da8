"""How uneven are the photon queries of the caustic-lab frame?  Takes the renderer's own photon map
(default budget), queries it on a grid of points on the table as dense as the 1080p pixel footprint,
and prints the distribution of photons read per query and, per block of 8 x 8 neighbouring queries
(one wave), the ratio of the heaviest lane to the mean: the share of lane-time a wave spends waiting."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from raytracer_amd import engine, renderer as R, workloads

R.load()
scene = workloads.caustic_lab()
r = R.Renderer(scene, 1920, 1080, 1)  # the caustic radius follows the frame height
r.begin_photon_pass()
stored = r.trace_photon_range(0, r.photon_emissions() if hasattr(r, "photon_emissions") else 1 << 24)
radius = r.caustic_radius()
params = r.launch_params()
buckets = params.photonBucketCount
photons = torch.from_numpy(R.device_to_host(r.photon_buffer(), stored * 36).view(np.float32).reshape(-1, 9).copy()).cuda()
print("photons", stored, "radius", radius, "buckets", buckets)
grid = engine.PhotonGrid(stored, buckets)
grid.build(photons, radius, sync=True)
off = grid.offsets.cpu().numpy().astype(np.int64)
print("offsets: last", off[-1], "monotonic", bool((np.diff(off) >= 0).all()), "largest cell", int(np.diff(off).max()),
      "finite photons", bool(np.isfinite(photons.cpu().numpy()).all()))
# the first-bounce queries of the 1080p frame: pinhole rays through the pixel centres, numbered in
# tiles of 8 x 8 pixels as the renderer numbers its paths, kept where they land on the table
W, H = 1920, 1080
cam = scene["cameras"][0]
m = np.asarray(cam["transform"], np.float64).reshape(4, 4).T
right, up, forward, eye = m[:3, 0], m[:3, 1], -m[:3, 2], m[:3, 3]
ty, tx, iy, ix = np.meshgrid(np.arange(H // 8), np.arange(W // 8), np.arange(8), np.arange(8), indexing="ij")
px, py = (tx * 8 + ix).reshape(-1), (ty * 8 + iy).reshape(-1)
half = np.tan(cam["verticalFov"] / 2.0)
sx = (2.0 * (px + 0.5) / W - 1.0) * half * W / H
sy = (1.0 - 2.0 * (py + 0.5) / H) * half
d = sx[:, None] * right + sy[:, None] * up + forward
d /= np.linalg.norm(d, axis=1, keepdims=True)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import engine_scene
top, blases = engine_scene(engine, scene)
org = np.concatenate([np.tile(eye, (len(d), 1)), np.full((len(d), 1), 1e-3)], 1).astype(np.float32)
dirs = np.concatenate([d, np.full((len(d), 1), 1e16)], 1).astype(np.float32)
hits, inst = top.trace_closest(torch.from_numpy(org).cuda(), torch.from_numpy(dirs).cuda())
torch.cuda.synchronize()
t = hits[:, 0].cpu().numpy()
on_table = (inst.cpu().numpy().view(np.uint32) == 0) & np.isfinite(t)
q = (org[:, :3] + t[:, None] * dirs[:, :3])[on_table].astype(np.float32)
print("pixels", len(d), "on the table", len(q))
qn = np.tile(np.array([0, 1, 0], np.float32), (len(q), 1))
qpos, qnor = torch.from_numpy(q).cuda(), torch.from_numpy(qn).cuda()
total = torch.zeros(1, dtype=torch.int64, device="cuda")


def timed(mask, label):
    qp, qn_ = qpos[mask].contiguous(), qnor[mask].contiguous()
    total.zero_()
    grid.gather(qp, qn_, visited=total)
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(5):
        grid.gather(qp, qn_)
    stop.record()
    torch.cuda.synchronize()
    ms = start.elapsed_time(stop) / 5
    plain = grid.gather(qp, qn_)
    grid.mark_buckets()
    start.record()
    for _ in range(5):
        masked = grid.gather(qp, qn_, masked=True)
    stop.record()
    torch.cuda.synchronize()
    same = bool(torch.equal(plain.view(torch.int32), masked.view(torch.int32)))
    print(f"    with bucket bits: {start.elapsed_time(stop) / 5:8.3f} ms, same floats: {same}")
    print(f"{label}: {len(qp)} queries, {int(total.item()) / len(qp):9.1f} photons read per query, {ms:8.3f} ms, "
          f"{len(qp) / ms / 1e3:7.1f} M queries/s")


x, z = qpos[:, 0], qpos[:, 2]
everything = torch.ones_like(x, dtype=torch.bool)
timed(everything, "all first-bounce queries")
timed(x < -0.5, "x < -0.5")
timed((x >= -0.5) & (x < 0.0), "-0.5 <= x < 0")
timed((x >= 0.0) & (x < 0.5), "0 <= x < 0.5")
timed(x >= 0.5, "x >= 0.5")
