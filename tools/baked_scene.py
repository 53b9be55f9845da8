"""Development aid: what a single tree over world-space triangles would win over the two-level walk.  The
dragon stand-in is traced as it is (five instances) and with every instance's triangles moved to world space
and merged into one mesh behind one identity instance; same rays (primary, and diffuse bounce rays from the
primary hits), closest-hit rate and visits per ray."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from raytracer_amd import engine, workloads
from helpers import engine_scene


def rate(top, o, d):
    o, d = torch.from_numpy(np.ascontiguousarray(o)).cuda(), torch.from_numpy(np.ascontiguousarray(d)).cuda()
    counters = torch.zeros(4, dtype=torch.int64, device="cuda")
    hits, inst = top.trace_closest(o, d, counters=counters)
    torch.cuda.synchronize()
    c = counters.cpu().numpy().astype(np.float64)
    hits, inst = top.trace_closest(o, d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        top.trace_closest(o, d, hits=hits, inst=inst)
    torch.cuda.synchronize()
    n = o.shape[0]
    return n / ((time.perf_counter() - t0) / 10) / 1e6, c[1] / n, c[2] / n, c[3] / n, hits


def merged(scene):
    vertices, indices, base = [], [], 0
    for gi, m in scene["instances"]:
        g = scene["geometries"][gi]
        v = np.array(g["vertices"], dtype=np.float32, copy=True)
        M = np.asarray(m, dtype=np.float32).reshape(4, 4).T  # column-major in the scene
        p = v[:, 0:3] @ M[:3, :3].T + M[:3, 3]
        v[:, 0:3] = p.astype(np.float32)
        vertices.append(v)
        indices.append(np.asarray(g["indices"]).reshape(-1, 3) + base)
        base += v.shape[0]
    out = dict(scene)
    out["geometries"] = [{"vertices": np.concatenate(vertices), "indices": np.concatenate(indices), "material": 0}]
    out["instances"] = [(0, np.eye(4, dtype=np.float32).T.reshape(-1))]
    return out


def main():
    engine.load()
    scene = workloads.dragon_cornell()
    one = merged(scene)
    o, d = workloads.camera_rays(scene, 1920, 1080)
    results = {}
    for name, s in (("five instances", scene), ("one world-space mesh", one)):
        top, blases = engine_scene(engine, s)
        r = rate(top, o, d)
        print(f"{name:22s} primary {r[0]:8.1f} Mrays/s  nodes {r[1]:.2f} triangles {r[2]:.2f} instances {r[3]:.2f}")
        t = r[4][:, 0].cpu().numpy()
        if name == "five instances":
            ok = (t > 0) & (t < 1e15)
            rng = np.random.default_rng(5)
            dirs = rng.normal(size=(ok.sum(), 3))
            dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
            bo = o[ok].copy()
            bo[:, :3] = o[ok, :3] + d[ok, :3] * (t[ok, None] * 0.999)
            bd = d[ok].copy()
            bd[:, :3] = dirs
            bo, bd = bo.astype(np.float32), bd.astype(np.float32)
        r = rate(top, bo, bd)
        print(f"{name:22s} bounce  {r[0]:8.1f} Mrays/s  nodes {r[1]:.2f} triangles {r[2]:.2f} instances {r[3]:.2f}", flush=True)
        results[name] = t


main()
