"""Development aid: closest-hit rate of one batch of diffuse bounce rays (from the primary hits of the
dragon stand-in, cosine-distributed directions) in the order the wavefront produces them and after
sorting them by a Morton code of origin cell and direction octant - the upper bound of what a ray
sorting stage between the bounces could win."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from raytracer_amd import engine, workloads
from helpers import engine_scene


def spread3(v):
    v = v.astype(np.uint64) & 0x3ff
    v = (v | (v << 16)) & 0x30000ff
    v = (v | (v << 8)) & 0x300f00f
    v = (v | (v << 4)) & 0x30c30c3
    v = (v | (v << 2)) & 0x9249249
    return v


def rate(top, o, d):
    o, d = torch.from_numpy(np.ascontiguousarray(o)).cuda(), torch.from_numpy(np.ascontiguousarray(d)).cuda()
    hits, inst = top.trace_closest(o, d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        top.trace_closest(o, d, hits=hits, inst=inst)
    torch.cuda.synchronize()
    return o.shape[0] / ((time.perf_counter() - t0) / 10) / 1e6


def main():
    engine.load()
    scene = workloads.dragon_cornell()
    top, blases = engine_scene(engine, scene)
    o, d = workloads.camera_rays(scene, 1920, 1080)
    hits, inst = top.trace_closest(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda())
    t = hits[:, 0].cpu().numpy()
    ok = (t > 0) & (t < 1e15)
    rng = np.random.default_rng(5)
    dirs = rng.normal(size=(ok.sum(), 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    bo = o[ok].copy()
    bo[:, :3] = o[ok, :3] + d[ok, :3] * (t[ok, None] * 0.999)
    bd = d[ok].copy()
    bd[:, :3] = dirs
    bo, bd = bo.astype(np.float32), bd.astype(np.float32)
    print(f"primary rays           {rate(top, o, d):8.1f} Mrays/s")
    print(f"bounce rays, as made   {rate(top, bo, bd):8.1f} Mrays/s")
    lo, hi = bo[:, :3].min(0), bo[:, :3].max(0)
    for bits in (4, 6, 8, 10):
        cell = np.clip(((bo[:, :3] - lo) / (hi - lo + 1e-9) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
        code = spread3(cell[:, 0]) | (spread3(cell[:, 1]) << 1) | (spread3(cell[:, 2]) << 2)
        octant = ((bd[:, 0] < 0).astype(np.uint64) | ((bd[:, 1] < 0).astype(np.uint64) << 1) | ((bd[:, 2] < 0).astype(np.uint64) << 2))
        for name, key in (("origin", code), ("origin+octant", (code << 3) | octant), ("octant+origin", (octant << 30) | code)):
            order = np.argsort(key, kind="stable")
            print(f"sorted, {bits:2d} bits/axis, {name:14s} {rate(top, bo[order], bd[order]):8.1f} Mrays/s", flush=True)
    shuffled = rng.permutation(len(bo))
    print(f"bounce rays, shuffled  {rate(top, bo[shuffled], bd[shuffled]):8.1f} Mrays/s")


main()
