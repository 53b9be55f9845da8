#!/usr/bin/env python3
"""Benchmark of the ray-intersection + photon-gather path on MI355X.

One "step" is one pass of the hot path over one resident batch of synthetic
input: the closest-hit kernel over every ray of a 1-sample-per-pixel, 5-bounce
frame of the Dragon-Cornell stand-in at 1920x1080 (primary rays plus four
generations of bounce rays), followed by one photon-grid build and one
radius-gather pass over the reference-size photon map.  The headline value is
closest-hit rays per second over the whole job; photon gathers/s and the grid
build time ride along in the same JSON line.

    python bench.py --gpus N --steps K --warmup W

For N > 1 launch with torch.distributed.run (one rank per GPU); every rank
holds a replica of the BVH and traces the frame for its own sample index (weak
scaling, no collective on the data path).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--bounces", type=int, default=5)
    p.add_argument("--workload", default="dragon-cornell", choices=["dragon-cornell", "soup1m", "sphere-field"])
    p.add_argument("--photons", type=int, default=1 << 23)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    return p.parse_args()


def build_scene(eng, scene):
    import torch
    blases = []
    for g in scene["geometries"]:
        v = torch.from_numpy(np.ascontiguousarray(g["vertices"], dtype=np.float32)).cuda()
        i = torch.from_numpy(np.ascontiguousarray(g["indices"].astype(np.int32))).cuda()
        blases.append(eng.Blas(v, i))
    return eng.Scene([(blases[gi], m) for gi, m in scene["instances"]]), blases


def bounce_rays(org, dirs, hits, seed):
    """Next ray generation: from every hit point into a random direction of the
    hemisphere around the reversed incoming ray (synthetic stand-in for a
    diffuse bounce).  Torch ops on the device; only used to prepare inputs."""
    import torch
    found = hits[:, 3].view(torch.int32) != -1
    t = hits[found, 0:1]
    p = org[found, :3] + dirs[found, :3] * t
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    d = torch.randn((p.shape[0], 3), device="cuda", generator=gen)
    d = d / d.norm(dim=1, keepdim=True)
    flip = (d * dirs[found, :3]).sum(1, keepdim=True) > 0
    d = torch.where(flip, -d, d)
    o = p - dirs[found, :3] * 1e-3
    org2 = torch.cat([o, torch.full((p.shape[0], 1), 0.001, device="cuda")], 1).contiguous()
    dir2 = torch.cat([d, torch.full((p.shape[0], 1), 1e16, device="cuda")], 1).contiguous()
    return org2, dir2


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from raytracer_amd import engine as eng
    from raytracer_amd import workloads

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    eng.load()

    # ---- scene + rays (resident in HBM before the timed region) ----
    if args.workload == "dragon-cornell":
        scene = workloads.dragon_cornell()
    elif args.workload == "soup1m":
        scene = workloads.triangle_soup(1_000_000)
    else:
        scene = workloads.sphere_field()
    top, blases = build_scene(eng, scene)
    stats = top.stats()
    if scene["cameras"]:
        o, d = workloads.camera_rays(scene, args.width, args.height, seed=rank)
    else:
        o, d = workloads.random_rays(scene, args.width * args.height, seed=rank)
    org, dirs = [torch.from_numpy(o).cuda()], [torch.from_numpy(d).cuda()]
    for b in range(1, args.bounces):
        hits, _ = top.trace_closest(org[-1], dirs[-1])
        o2, d2 = bounce_rays(org[-1], dirs[-1], hits, seed=1000 * rank + b)
        if o2.shape[0] == 0:
            break
        org.append(o2)
        dirs.append(d2)
    org_all = torch.cat(org).contiguous()
    dir_all = torch.cat(dirs).contiguous()
    n_rays = org_all.shape[0]
    hits = torch.empty((n_rays, 4), dtype=torch.float32, device="cuda")
    inst = torch.empty(n_rays, dtype=torch.int32, device="cuda")

    # per-ray traversal averages (for the algorithmic byte count), outside the timed region
    counters = torch.zeros(4, dtype=torch.int64, device="cuda")
    top.trace_closest(org_all, dir_all, counters=counters, hits=hits, inst=inst)
    c = counters.cpu().numpy().astype(np.float64)
    nodes_per_ray, tris_per_ray = c[1] / c[0], c[2] / c[0]
    bytes_per_ray = 52.0 + nodes_per_ray * 80.0 + tris_per_ray * 36.0

    # ---- photon map (reference sizes) ----
    n_ph = args.photons
    buckets = workloads.PHOTON_BUCKETS
    photons = torch.from_numpy(workloads.caustic_photons(n_ph, seed=2 + rank)).cuda()
    radius = 0.0015
    grid = eng.PhotonGrid(n_ph, buckets)
    qp, qn = workloads.gather_queries(args.width * args.height, seed=7 + rank)
    qp, qn = torch.from_numpy(qp).cuda(), torch.from_numpy(qn).cuda()
    flux = torch.empty_like(qp)
    n_queries = qp.shape[0]

    stream = torch.cuda.current_stream()

    def step(events=None):
        if events is not None:
            events[0].record(stream)
        top.trace_closest(org_all, dir_all, stream=stream, hits=hits, inst=inst)
        if events is not None:
            events[1].record(stream)
        grid.build(photons, radius, stream=stream)
        if events is not None:
            events[2].record(stream)
        grid.gather(qp, qn, stream=stream, out=flux)
        if events is not None:
            events[3].record(stream)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(ev[k])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        counts = torch.tensor([n_rays, n_queries], dtype=torch.float64, device="cuda")
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        total_rays, total_queries = float(counts[0].item()), float(counts[1].item())
    else:
        total_rays, total_queries = float(n_rays), float(n_queries)

    trace_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    build_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
    gather_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in ev]))

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    mrays = total_rays * args.steps / elapsed / 1e6
    achieved = n_rays * bytes_per_ray / (trace_ms * 1e-3) / 1e9
    out = {
        "metric": "closest_hit_mrays_per_s",
        "value": mrays,
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload} stand-in, {args.width}x{args.height}, 1 spp, {args.bounces} ray generations "
                        f"(primary + synthetic bounces), then photon grid build + gather "
                        f"({n_ph} photons, {buckets} buckets, {n_queries} queries)",
            "rays_per_step_per_gpu": n_rays,
            "triangles": int(stats.triangleCount),
            "bvh_nodes": int(stats.nodeCount),
            "bvh_bytes": int(stats.nodeBytes + stats.triangleBytes),
            "instances": int(stats.instanceCount),
            "parallelism": f"replicated BVH, {world} x sample-index shards",
        },
        "trace_kernel_mrays_per_s": n_rays / (trace_ms * 1e-3) / 1e6,
        "photon_gathers_per_s": total_queries / (gather_ms * 1e-3),
        "photon_grid_build_ms": build_ms,
        "photon_grid_build_gbs": (84.0 * n_ph + 4.0 * (buckets + 1)) / (build_ms * 1e-3) / 1e9,
        "roofline": {
            "bound": "hbm",
            "kernel": "traceKernel<closest>",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": None,
            "bytes_per_ray": bytes_per_ray,
            "nodes_per_ray": nodes_per_ray,
            "triangles_per_ray": tris_per_ray,
            "kernel_ms": trace_ms,
        },
    }

    if not args.no_cpu_baseline:
        from oracle import binding as orc  # the CPU baseline leg: the oracle, timed, never shipped
        ref = orc.Scene()
        for g in scene["geometries"]:
            ref.add_geometry(g["vertices"], g["indices"], g["material"])
        for gi, m in scene["instances"]:
            ref.add_instance(gi, m)
        ref.commit()
        threads = min(16, os.cpu_count() or 1)
        rng = np.random.default_rng(0)
        o_h, d_h = org_all.cpu().numpy(), dir_all.cpu().numpy()
        probe = rng.choice(n_rays, size=min(n_rays, 100000), replace=False)
        t1 = time.perf_counter()
        ref.trace_closest(o_h[probe], d_h[probe], use_bvh=True, threads=threads)
        rate = len(probe) / (time.perf_counter() - t1)
        sample = rng.choice(n_rays, size=int(min(n_rays, max(len(probe), rate * args.cpu_seconds))), replace=False)
        t1 = time.perf_counter()
        tuv, prim, rinst = ref.trace_closest(o_h[sample], d_h[sample], use_bvh=True, threads=threads)
        cpu_s = time.perf_counter() - t1
        # the sample doubles as a parity check of the timed GPU output
        h = hits.cpu().numpy()
        same = bool(np.array_equal(h[sample, 3].view(np.uint32), prim) and
                    np.array_equal(h[sample, 0].view(np.uint32), tuv[:, 0].view(np.uint32)))
        out["cpu_baseline"] = {
            "value": len(sample) / cpu_s / 1e6,
            "unit": "Mrays/s",
            "cores": threads,
            "kind": "port",
            "sample": f"{len(sample)} rays drawn uniformly from the step's {n_rays} rays, oracle BVH, "
                      f"{threads} std::thread workers",
            "gpu_output_matches_on_sample": same,
        }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
