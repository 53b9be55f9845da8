"""Incoherent rays on a uniform triangle soup generated on the GPU (the distribution of workloads.triangle_soup):
build time, nodes and triangles per ray, Mrays/s and the algorithmic HBM roofline as one JSON line per size.
Run under rocprofv3 (kernel trace, FETCH_SIZE, WRITE_SIZE in passes of their own) for the committed summaries.
    python tools/soup_profile.py 10000000 100000000 [--rays 16777216]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("RT_BENCH_ROOT"):  # a side-by-side build of the package (tools/variants.sh)
    sys.path.insert(0, os.path.abspath(os.environ["RT_BENCH_ROOT"]))
import torch
from raytracer_amd import engine

NODE_BYTES, TRIANGLE_BYTES, RAY_IO_BYTES, PEAK_GBS = 80.0, 36.0, 52.0, 8000.0


def soup_on_device(n, seed=4):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    centre = torch.rand((n, 1, 3), generator=gen, device="cuda")
    size = (torch.rand((n, 1, 1), generator=gen, device="cuda") + 0.5) * n ** (-1.0 / 3.0)
    offs = torch.randn((n, 3, 3), generator=gen, device="cuda")
    offs /= offs.norm(dim=2, keepdim=True)
    offs *= 0.5 * size
    offs += centre
    return offs.reshape(-1, 3).contiguous(), torch.arange(3 * n, dtype=torch.int32, device="cuda").reshape(-1, 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sizes", type=int, nargs="+")
    ap.add_argument("--rays", type=int, default=1 << 24)
    ap.add_argument("--repeats", type=int, default=5)
    args = ap.parse_args()
    engine.load()
    warm = soup_on_device(4096)  # the first build of a process also loads the build kernels: not part of a build time
    engine.Blas(*warm)
    del warm
    for n in args.sizes:
        positions, indices = soup_on_device(n)
        blas = engine.Blas(positions, indices)
        top = engine.Scene([(blas, None)])
        stats = top.stats()
        gen = torch.Generator(device="cuda").manual_seed(5)
        o = torch.rand((args.rays, 4), generator=gen, device="cuda") * 1.1 - 0.05
        o[:, 3] = 0.001
        d = torch.randn((args.rays, 4), generator=gen, device="cuda")
        d[:, :3] /= d[:, :3].norm(dim=1, keepdim=True)
        d[:, 3] = 1e16
        counters = torch.zeros(4, dtype=torch.int64, device="cuda")
        top.trace_closest(o, d, counters=counters)
        c = counters.cpu().numpy().astype(float)
        hits, inst = top.trace_closest(o, d)
        torch.cuda.synchronize()
        begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        begin.record()
        for _ in range(args.repeats):
            top.trace_closest(o, d, hits=hits, inst=inst)
        end.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / args.repeats
        ms = begin.elapsed_time(end) / args.repeats
        nodes, tris = c[1] / c[0], c[2] / c[0]
        per_ray = RAY_IO_BYTES + nodes * NODE_BYTES + tris * TRIANGLE_BYTES
        gbs = args.rays * per_ray / (ms * 1e-3) / 1e9
        found = float((hits[:, 3].view(torch.int32) != -1).float().mean())
        print(json.dumps({
            "workload": f"uniform triangle soup, {n} triangles, {args.rays} incoherent rays per launch (closest hit)",
            "triangles": n, "nodes": int(stats.nodeCount), "depth": int(stats.maxDepth),
            "build_ms": float(blas.stats().buildMilliseconds), "mrays_per_s": args.rays / (ms * 1e-3) / 1e6,
            "launch_ms": ms, "wall_ms": wall * 1e3, "nodes_per_ray": nodes, "triangles_per_ray": tris,
            "bytes_per_ray": per_ray, "hit_fraction": found,
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": PEAK_GBS, "unit": "GB/s", "frac": gbs / PEAK_GBS,
                         "algorithmic_bytes_per_launch": args.rays * per_ray},
        }), flush=True)
        del top, blas, positions, indices, o, d, hits, inst
        torch.cuda.empty_cache()


main()
