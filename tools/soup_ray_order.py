"""Development aid: what ordering the incoherent rays of the soup benchmark by a Morton code of their origin cell
(and direction octant) would win - the upper bound for a sorting step inside rt_trace_closest.
    python tools/soup_ray_order.py 1000000 100000000"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import engine
from soup_profile import soup_on_device


def spread3(v):
    v = v & 0x3ff
    v = (v | (v << 16)) & 0x30000ff
    v = (v | (v << 8)) & 0x300f00f
    v = (v | (v << 4)) & 0x30c30c3
    v = (v | (v << 2)) & 0x9249249
    return v


def rate(top, o, d):
    hits, inst = top.trace_closest(o, d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        top.trace_closest(o, d, hits=hits, inst=inst)
    torch.cuda.synchronize()
    return o.shape[0] / ((time.perf_counter() - t0) / 3) / 1e6


def main():
    engine.load()
    rays = 1 << 24
    for n in (int(a) for a in sys.argv[1:]):
        positions, indices = soup_on_device(n)
        blas = engine.Blas(positions, indices)
        top = engine.Scene([(blas, None)])
        gen = torch.Generator(device="cuda").manual_seed(5)
        o = torch.rand((rays, 4), generator=gen, device="cuda") * 1.1 - 0.05
        o[:, 3] = 0.001
        d = torch.randn((rays, 4), generator=gen, device="cuda")
        d[:, :3] /= d[:, :3].norm(dim=1, keepdim=True)
        d[:, 3] = 1e16
        print(f"{n} triangles, as drawn: {rate(top, o, d):8.1f} Mrays/s", flush=True)
        for bits in (5, 7, 10):
            cell = ((o[:, :3] + 0.05) / 1.1 * (1 << bits)).long().clamp(0, (1 << bits) - 1)
            code = spread3(cell[:, 0]) | (spread3(cell[:, 1]) << 1) | (spread3(cell[:, 2]) << 2)
            octant = (d[:, 0] < 0).long() | ((d[:, 1] < 0).long() << 1) | ((d[:, 2] < 0).long() << 2)
            for name, key in (("origin", code), ("origin+octant", (code << 3) | octant), ("octant+origin", (octant << 30) | code)):
                order = torch.argsort(key, stable=True)
                print(f"{n} triangles, {bits:2d} bits/axis, {name:14s} {rate(top, o[order].contiguous(), d[order].contiguous()):8.1f} Mrays/s", flush=True)


main()
