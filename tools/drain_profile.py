"""Development aid (engine built with -DRT_TRACE_PROFILE): one batched closest-hit launch at a time over
the dragon stand-in, and how its duration splits into the time until some wave finds the ray feed empty
and the drain after that, with the lane use during the drain."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("RT_BENCH_ROOT"):
    sys.path.insert(0, os.path.abspath(os.environ["RT_BENCH_ROOT"]))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from raytracer_amd import engine, workloads
from helpers import engine_scene


def main():
    lib = engine.load()
    scene = workloads.dragon_cornell()
    top, blases = engine_scene(engine, scene)
    out = (C.c_ulonglong * 8)()
    cases = {"primary 1080p": workloads.camera_rays(scene, 1920, 1080), "random 2M": workloads.random_rays(scene, 1 << 21),
             "random 8M": workloads.random_rays(scene, 1 << 23)}
    # bounce rays: from the primary hits along random directions
    o, d = cases["primary 1080p"]
    hits, inst = top.trace_closest(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda())
    t = hits[:, 0].cpu().numpy()
    ok = t > 0
    rng = np.random.default_rng(5)
    dirs = rng.normal(size=(ok.sum(), 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    bo = o[ok].copy()
    bo[:, :3] = o[ok, :3] + d[ok, :3] * (t[ok, None] * 0.999)
    bd = d[ok].copy()
    bd[:, :3] = dirs
    cases["bounce from primary hits"] = (bo.astype(np.float32), bd.astype(np.float32))
    for name, (o, d) in cases.items():
        o, d = torch.from_numpy(np.ascontiguousarray(o)).cuda(), torch.from_numpy(np.ascontiguousarray(d)).cuda()
        top.trace_closest(o, d)
        lib.rt_trace_clock(out, 1)
        top.trace_closest(o, d)
        lib.rt_trace_clock(out, 1)
        start, drained, end, turns, lanes = out[0], out[1], out[2], out[3], out[4]
        print(f"{name:26s} rays {o.shape[0]:8d}  total {(end - start) / 100:8.1f} us  until the feed is empty "
              f"{(drained - start) / 100:8.1f} us  drain {(end - drained) / 100:8.1f} us  "
              f"lanes per wave turn in the drain {lanes / max(turns, 1):5.1f}", flush=True)


main()
