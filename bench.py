#!/usr/bin/env python3
"""Benchmark of the ray-intersection + photon-gather path on MI355X.

One "step" is one pass of the hot path over one frame of synthetic input: one camera sample per
pixel of the Dragon-Cornell stand-in (a ~870k-triangle absorbing-glass mesh in a Cornell box with an
emissive panel) at 1920x1080, preview profile (5 bounces), through rt::Renderer::render(): generate,
then per bounce trace / shade (the shadow rays of a bounce ride in the same persistent launch as the
closest-hit rays of the next), then accumulate; the renderer takes the calls in passes of several
samples (sixteen at this size, two passes in flight; the image is the same).  Scene, BVH and every buffer are resident in HBM
before the timed region.  The headline value is rays traced (closest-hit + shadow) per second over
the whole step; the trace kernel's share is priced against the HBM roofline.

After the timed steps the photon map of the reference's size (2^23 photons, 2^22 buckets) is built
and gathered once per step count for the second metric (photon gathers/s, grid build time).

    python bench.py --gpus N --steps K --warmup W

For N > 1 either launch with torch.distributed.run, one rank per GPU over RCCL, or run the plain command:
when --gpus N > 1 and the environment names no WORLD_SIZE, this process starts the N ranks itself as child
processes (before anything here has touched the GPU), passes rank 0's JSON line on to its stdout and
everything else to stderr, and fails when a rank fails.  Every rank holds a replica of
the scene and the N ranks share each frame (strong scaling of one frame): the bands of eight pixel rows
are dealt out in turn (band b to rank b mod N), so a step is still one sample of every pixel of the
frame; in the final profile every rank traces its share of the photon emissions and the lists are
exchanged rank to rank into every rank's photon buffer before each rebuilds the same grid.  The bands
are all-gathered once after the timed steps (reported as tile_allgather_ms).  N = 1 runs the plain
single-GPU renderer.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
if os.environ.get("RT_BENCH_ROOT"):  # development aid: a side-by-side build of the package (tools/variants.sh)
    sys.path.insert(0, os.path.abspath(os.environ["RT_BENCH_ROOT"]))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
NODE_BYTES, TRIANGLE_BYTES, RAY_IO_BYTES = 80.0, 36.0, 52.0  # algorithmic bytes: wide node, 3 x float3, ray in + hit out


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=64)
    p.add_argument("--warmup", type=int, default=8)
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--workload", default="dragon-cornell",
                   choices=["dragon-cornell", "cornell", "sphere-field", "caustic-lab"])
    p.add_argument("--photons", type=int, default=1 << 23)
    p.add_argument("--tiles", type=int, default=1,
                   help="regions of the frame that run side by side per render() call")
    p.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                   help="nccl (RCCL) for one rank per GPU; gloo rehearses the multi-rank path with several ranks on one GPU")
    p.add_argument("--batch", type=int, default=None,
                   help="samples a wavefront pass takes (default: as many as the renderer's state budget holds)")
    p.add_argument("--in-flight", type=int, default=0,
                   help="passes of a region in flight at once, each on a stream of its own (default 0: the renderer's choice)")
    p.add_argument("--shared", action="store_true",
                   help="run the multi-GPU code path (process group, dealt-out bands, photon exchange) with however "
                        "many ranks there are, one included: a rehearsal of the N > 1 path on one GPU")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-photon-bench", action="store_true")
    p.add_argument("--no-extra-legs", action="store_true",
                   help="leave out the caustic-lab and triangle-soup legs that ride along with the default line at N = 1")
    p.add_argument("--soup-sizes", type=int, nargs="*", default=[1000000, 10000000, 100000000],
                   help="triangle counts of the soup leg (a size that does not fit the free memory is left out)")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    return p.parse_args()


def make_scene(name):
    from raytracer_amd import workloads
    if name == "dragon-cornell":
        return workloads.dragon_cornell(), 0
    if name == "cornell":
        return workloads.cornell_box(light="panel"), 0
    if name == "sphere-field":
        return workloads.sphere_field(), 0
    return workloads.caustic_lab(), 1


def photon_benchmark(eng, workloads, args, rank, steps, cpu_seconds=0.0):
    """Grid build + gather over the reference-size photon map; queries are one per pixel of a frame
    looking at the table (row-major, so neighbouring lanes query neighbouring cells)."""
    import torch
    n_ph, buckets = args.photons, workloads.PHOTON_BUCKETS
    photons = torch.from_numpy(workloads.caustic_photons(n_ph, seed=2 + rank)).cuda()
    radius = 0.0015
    grid = eng.PhotonGrid(n_ph, buckets)
    w, h = args.width, args.height
    xs = (np.arange(w, dtype=np.float32) + 0.5) / w * 2.0 - 1.0
    zs = ((np.arange(h, dtype=np.float32) + 0.5) / h * 2.0 - 1.0) * (h / w)
    qp = np.zeros((h, w, 3), dtype=np.float32)
    qp[..., 0], qp[..., 2] = xs[None, :], zs[:, None]
    qn = np.zeros_like(qp)
    qn[..., 1] = 1.0
    qp, qn = torch.from_numpy(qp.reshape(-1, 3)).cuda(), torch.from_numpy(qn.reshape(-1, 3)).cuda()
    flux = torch.empty_like(qp)
    visited = torch.zeros(1, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream()
    grid.build(photons, radius, stream=stream)
    grid.gather(qp, qn, stream=stream, out=flux, visited=visited)
    per_query = float(visited.item()) / qp.shape[0]
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
    for k in range(steps):
        ev[k][0].record(stream)
        grid.build(photons, radius, stream=stream)
        ev[k][1].record(stream)
        grid.gather(qp, qn, stream=stream, out=flux)
        ev[k][2].record(stream)
    torch.cuda.synchronize()
    build_ms = float(np.median([e[0].elapsed_time(e[1]) for e in ev]))
    gather_ms = float(np.median([e[1].elapsed_time(e[2]) for e in ev]))
    gather_bytes = 36.0 + 27.0 * 8.0 + 36.0 * per_query  # query in + flux out, 27 cell ranges, photons read
    cpu = cpu_gather_baseline(grid, qp, qn, radius, buckets, cpu_seconds) if cpu_seconds > 0.0 else None
    return {
        "cpu_baseline": cpu,
        "photons": n_ph, "buckets": buckets, "queries": int(qp.shape[0]), "radius": radius,
        "photons_read_per_query": per_query,
        "gathers_per_s": qp.shape[0] / (gather_ms * 1e-3),
        "gather_ms": gather_ms,
        "gather_gbs": qp.shape[0] * gather_bytes / (gather_ms * 1e-3) / 1e9,
        "gather_frac_of_hbm_peak": qp.shape[0] * gather_bytes / (gather_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "grid_build_ms": build_ms,
        "grid_build_gbs": (84.0 * n_ph + 4.0 * (buckets + 1)) / (build_ms * 1e-3) / 1e9,
        "grid_build_frac_of_hbm_peak": (84.0 * n_ph + 4.0 * (buckets + 1)) / (build_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
    }


def library_hash():
    from raytracer_amd import engine as eng
    with open(eng.LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def profiled_traffic():
    """HBM bytes per trace launch (FETCH_SIZE x2-corrected plus WRITE_SIZE, separate rocprofv3 --pmc passes of
    this same command) from the committed summary that was taken with the very library this run loads
    (profiles/*_bench_summary.txt record the sha256 of librt_engine.so), or None when no summary matches:
    a number measured on other kernels would be stale."""
    import glob
    current = library_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_summary.txt")), reverse=True):
        lines = open(path).read().splitlines()
        if not any(line.startswith("librt_engine.so sha256") and current in line for line in lines):
            continue
        counters = False  # the summary lists kernel times first; the counter table follows its own heading
        for line in lines:
            if line.startswith("== per-launch PMC means"):
                counters = True
                continue
            f = line.split()
            if counters and len(f) >= 6 and f[0].startswith("traceKernel<false>"):
                try:
                    return float(f[3]) * 1e6 + float(f[4]) * 1024.0, os.path.basename(path)  # fetch_x2_MB, WRITE_KiB
                except ValueError:
                    pass
    return None


def copy_ceiling_gbs():
    """What a plain device-to-device copy of 1 GiB reaches on this GPU, read + write bytes per second: the
    achievable ceiling to read next to the 8 TB/s specification."""
    import torch
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    src.fill_(1)
    dst.copy_(src)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for _ in range(8):
        dst.copy_(src)
    stop.record()
    torch.cuda.synchronize()
    return 8 * 2.0 * n / (start.elapsed_time(stop) * 1e-3) / 1e9


def host_cpu():
    """(model name, threads this process may use): the affinity mask, cut to the pool's per-GPU share when the
    environment states one (OMP_NUM_THREADS)."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    share = os.environ.get("OMP_NUM_THREADS", "")
    threads = min(allowed, int(share)) if share.isdigit() and int(share) > 0 else allowed
    return model, allowed, threads


def cpu_baseline(scene, args, width, height, profile):
    """The oracle's integrator (same frame, sample 0) on the host cores over a bounded band of pixel
    rows from the middle of the image, rays counted the way the GPU counts them."""
    from oracle import binding as orc  # the CPU baseline leg: the oracle, timed, never shipped
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ref_setup import Setup
    ref = orc.Scene()
    for g in scene["geometries"]:
        ref.add_geometry(g["vertices"], g["indices"], g["material"])
    for gi, m in scene["instances"]:
        ref.add_instance(gi, m)
    ref.commit()
    model, allowed, threads = host_cpu()
    setup = Setup(scene, width, height, final=False)  # preview path lengths; no photon map on the CPU leg
    setup.params.sample = 0
    # probe: one sample of the middle half of the frame; then as many further samples of the same
    # rows as fit the time budget (each sample index draws new random streams, like the GPU's steps)
    rows = height // 2
    first = (height - rows) // 2
    orc.ray_counter(reset=True)
    t1 = time.perf_counter()
    ref.render_sample(setup.params, first * width, (first + rows) * width, threads)
    probe_s = time.perf_counter() - t1
    probe_rays = orc.ray_counter(reset=True)
    samples = int(max(1, min(64, args.cpu_seconds / max(probe_s, 1e-3))))
    t1 = time.perf_counter()
    for k in range(samples):
        setup.params.sample = 1 + k
        ref.render_sample(setup.params, first * width, (first + rows) * width, threads)
    cpu_s = time.perf_counter() - t1
    rays = orc.ray_counter(reset=True)
    return {
        "value": rays / cpu_s / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port", "cpu_model": model,
        "hardware_threads_visible": allowed,
        "sample": f"{samples} samples of {rows} pixel rows ({samples * rows * width} paths, {rays} rays) from the middle "
                  f"of the same frame, preview depth, the oracle's integrator over its own BVH, {threads} std::thread "
                  f"workers ({allowed} hardware threads visible, {threads} this job may use), {cpu_s:.1f} s "
                  f"(probe: {probe_rays} rays in {probe_s:.2f} s)",
    }


def cpu_gather_baseline(grid, qp, qn, radius, buckets, seconds):
    """The oracle's causticLighting loop (device.cu:722-766 restated) over the GPU-built grid of the photon
    benchmark, on every k-th query of the same query set, timed on the host threads."""
    from oracle import binding as orc  # the CPU baseline leg: the oracle, timed, never shipped
    model, allowed, threads = host_cpu()
    srt = grid.sorted.cpu().numpy()
    off = grid.offsets.cpu().numpy().view(np.uint32)
    qp, qn = qp.cpu().numpy(), qn.cpu().numpy()
    probe = slice(0, None, max(1, qp.shape[0] // 16384))
    t1 = time.perf_counter()
    orc.photon_gather(srt, off, buckets, radius, qp[probe], qn[probe], threads=threads)
    probe_s = time.perf_counter() - t1
    per_query = probe_s / max(len(qp[probe]), 1)
    stride = max(1, int(qp.shape[0] * per_query / max(seconds, 0.1)))
    pick = slice(0, None, stride)
    t1 = time.perf_counter()
    orc.photon_gather(srt, off, buckets, radius, qp[pick], qn[pick], threads=threads)
    cpu_s = time.perf_counter() - t1
    n = len(qp[pick])
    return {"value": n / cpu_s, "unit": "gathers/s", "cores": threads, "kind": "port", "cpu_model": model,
            "sample": f"every {stride}th of the benchmark's {qp.shape[0]} queries ({n}) against the same "
                      f"{srt.shape[0]}-photon grid, the oracle's gather loop on {threads} std::thread workers, {cpu_s:.2f} s"}


def run_workload(args, workload, steps, warmup, world, rank, wire):
    """Builds the scene and its renderer, counts the traversal of one step, runs `warmup` untimed and `steps`
    timed steps between barriers, then repeats the steps on one stream with every launch between HIP events.
    Returns what the result line is made of."""
    import torch
    import torch.distributed as dist
    from raytracer_amd import distributed as D
    from raytracer_amd import renderer as R
    from raytracer_amd import workloads  # noqa: F401
    scene, profile = make_scene(workload)
    budget = (1 << 24, 1 << 23, 1 << 22) if profile == R.FINAL else None
    if world > 1 or args.shared:
        # the ranks share every frame: bands of eight rows dealt out in turn, emission ranges per rank
        shared = D.DistributedRenderer(scene, args.width, args.height, profile, photon_budget=budget, batch=args.batch)
        renderer = shared.renderer
        step = shared.render
    else:
        shared = None
        renderer = R.Renderer(scene, args.width, args.height, profile)
        if budget:
            renderer.set_photon_budget(*budget)
        if args.batch:
            renderer.set_sample_batch(args.batch)
        if os.environ.get("RT_BENCH_RAY_SHARE"):  # development aid: "4/5" leaves a fifth of the wave slots to other launches
            renderer.set_ray_share(*(int(x) for x in os.environ["RT_BENCH_RAY_SHARE"].split("/")))
        step = renderer.render

    # traversal averages per ray (for the algorithmic byte count), outside the timed region
    if profile == R.FINAL:
        # sample 0 starts with a photon pass, whose rays the traversal counters do not see
        step()
        renderer.statistics()
    renderer.set_measurement(count_traversal=True)
    step()
    c = renderer.statistics()
    counted = np.array([c["extend_rays"] + c["shadow_rays"], c["nodes"], c["triangles"]], dtype=np.float64)
    if world > 1:
        t = torch.from_numpy(counted).to(wire)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        counted = t.cpu().numpy()
    counted_rays = max(counted[0], 1.0)
    nodes_per_ray = counted[1] / counted_rays
    tris_per_ray = counted[2] / counted_rays
    bytes_per_ray = RAY_IO_BYTES + nodes_per_ray * NODE_BYTES + tris_per_ray * TRIANGLE_BYTES

    renderer.set_concurrent_tiles(args.tiles, args.in_flight)
    in_flight = renderer.samples_in_flight()  # 0 on the command line: the renderer's own choice
    renderer.set_measurement(time_stages=False)
    for _ in range(warmup):
        step()
    renderer.synchronize()
    renderer.statistics()
    if shared:
        shared.exchanges.clear()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    renderer.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    stats = renderer.statistics()
    rays = float(stats["extend_rays"] + stats["shadow_rays"])
    exchanges = list(shared.exchanges) if shared else []

    # once per image, not per sample: the bands of every rank into the full frame
    tile_allgather_ms = None
    if shared:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        frame = shared.gather_buffer(0)
        torch.cuda.synchronize()
        tile_allgather_ms = (time.perf_counter() - t1) * 1e3
        assert frame.shape == (args.height, args.width, 4)

    # Per-stage kernel time, outside the timed region: the same number of steps on one stream (the
    # parts of a tiled frame overlap in time, which would count the overlap twice), every launch
    # bracketed by HIP events on the stream it runs on.
    # the passes keep their number of samples: a launch here is a launch of the timed region, alone on the GPU
    batch = renderer.sample_batch()
    renderer.set_concurrent_tiles(1)
    renderer.set_sample_batch(batch)
    renderer.set_measurement(time_stages=True)
    for _ in range(batch):
        step()
    renderer.stage_timings()
    renderer.statistics()
    staged_steps = max(batch, steps // batch * batch)  # whole passes
    for _ in range(staged_steps):
        step()
    renderer.synchronize()
    stages = renderer.stage_timings()
    staged = renderer.statistics()
    staged_rays = float(staged["extend_rays"] + staged["shadow_rays"])
    per_rank = [rays]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=wire)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([rays, float(stats["gather_queries"])], dtype=torch.float64, device=wire)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [float(e[0].item()) for e in every]
        total_rays, total_gathers = sum(per_rank), sum(float(e[1].item()) for e in every)
    else:
        total_rays, total_gathers = rays, float(stats["gather_queries"])

    # one photon pass by itself (final profile on one GPU): trace of every emission, grid build, bucket bits
    photon_pass_ms = stored_photons = None
    if profile == R.FINAL and not shared:
        renderer.synchronize()
        renderer.set_manual_photon_passes(True)
        while not renderer.photon_pass_due():
            step()
        renderer.synchronize()
        t1 = time.perf_counter()
        renderer.begin_photon_pass()
        stored_photons = renderer.trace_photon_range(0, renderer.photon_emissions())
        renderer.finish_photon_pass(stored_photons)
        renderer.synchronize()
        photon_pass_ms = (time.perf_counter() - t1) * 1e3
    p = renderer.launch_params()
    result = dict(scene=scene, photon_pass_ms=photon_pass_ms, stored_photons=stored_photons, scene_triangles=int(sum(scene["geometries"][g]["indices"].shape[0] for g, _ in scene["instances"])),
                  instances=len(scene["instances"]), light_count=int(p.lightCount), profile=profile, shared=bool(shared),
                  elapsed=elapsed, rays=rays, total_rays=total_rays, total_gathers=total_gathers, per_rank=per_rank,
                  stats=stats, stages=stages, staged_steps=staged_steps, staged_rays=staged_rays, batch=batch,
                  in_flight=in_flight, bytes_per_ray=bytes_per_ray, nodes_per_ray=nodes_per_ray, tris_per_ray=tris_per_ray,
                  exchanges=exchanges, tile_allgather_ms=tile_allgather_ms)
    renderer.close()
    return result


def caustic_lab_leg(args, wire):
    """caustic-lab stand-in at 1080p, final profile (32 bounces, 2^24 emissions into at most 2^23 photons over 2^22
    buckets before every fourth sample, renderer.cpp:177, 533-535) through the same measurement as the headline."""
    import copy
    mine = copy.copy(args)
    mine.tiles, mine.batch, mine.in_flight = 1, None, 0
    steps, warmup = 64, 32
    m = run_workload(mine, "caustic-lab", steps, warmup, 1, 0, wire)
    trace_ms, launches = m["stages"]["trace"]
    achieved = m["staged_rays"] * m["bytes_per_ray"] / (trace_ms * 1e-3) / 1e9
    return {
        "workload": f"caustic-lab stand-in, {args.width}x{args.height}, final profile, photon pass before every fourth step",
        "steps": steps, "warmup": warmup,
        "ms_per_step": m["elapsed"] / steps * 1e3,
        "mrays_per_s": m["total_rays"] / m["elapsed"] / 1e6,
        "rays_per_step": m["total_rays"] / steps,
        "gathers_per_s_in_frame": m["total_gathers"] / m["elapsed"],
        "gather_queries_per_step": m["total_gathers"] / steps,
        "photon_pass_ms": m["photon_pass_ms"],
        "photons_stored_per_pass": m["stored_photons"],
        "samples_per_pass": m["batch"], "passes_in_flight": m["in_flight"],
        "stage_ms_per_step": {k: v[0] / m["staged_steps"] for k, v in m["stages"].items()},
        "roofline": {"bound": "hbm", "kernel": "traceKernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "bytes_per_ray": m["bytes_per_ray"],
                     "nodes_per_ray": m["nodes_per_ray"], "triangles_per_ray": m["tris_per_ray"],
                     "avg_launch_ms": trace_ms / max(launches, 1), "launches": launches},
    }


def soup_leg(eng, sizes, rays=1 << 24, repeats=3):
    """Uniform triangle soups (centroids uniform in the unit cube, edge length ~ U(0.5, 1.5) n^(-1/3), drawn on the
    GPU) and 2^24 incoherent closest-hit rays: build time of the bottom-level tree, Mrays/s, the ray kernel against
    the HBM roofline by algorithmic bytes (52 B per ray + 80 B per node visited + 36 B per triangle tested)."""
    import torch
    out = []
    # a small build first: the first one of a process also loads the build kernels, which is no part of a build time
    eng.Blas(torch.rand((3 * 4096, 3), device="cuda"), torch.arange(3 * 4096, dtype=torch.int32, device="cuda").reshape(-1, 3))
    for n in sizes:
        need = n * (36 + 12 + 48 + 40) * 2 + rays * 64  # triangles, indices, packed copy, nodes and sort scratch; rays
        if torch.cuda.mem_get_info()[0] < need + (8 << 30):
            out.append({"triangles": n, "skipped": "not enough free device memory"})
            continue
        gen = torch.Generator(device="cuda").manual_seed(4)
        centre = torch.rand((n, 1, 3), generator=gen, device="cuda")
        size = (torch.rand((n, 1, 1), generator=gen, device="cuda") + 0.5) * n ** (-1.0 / 3.0)
        corners = torch.randn((n, 3, 3), generator=gen, device="cuda")
        corners /= corners.norm(dim=2, keepdim=True)
        corners *= 0.5 * size
        corners += centre
        positions = corners.reshape(-1, 3).contiguous()
        indices = torch.arange(3 * n, dtype=torch.int32, device="cuda").reshape(-1, 3)
        del centre, size, corners
        blas = eng.Blas(positions, indices)
        top = eng.Scene([(blas, None)])
        gen = torch.Generator(device="cuda").manual_seed(5)
        o = torch.rand((rays, 4), generator=gen, device="cuda") * 1.1 - 0.05
        o[:, 3] = 0.001
        d = torch.randn((rays, 4), generator=gen, device="cuda")
        d[:, :3] /= d[:, :3].norm(dim=1, keepdim=True)
        d[:, 3] = 1e16
        counters = torch.zeros(4, dtype=torch.int64, device="cuda")
        top.trace_closest(o, d, counters=counters)
        c = counters.cpu().numpy().astype(float)
        hits, inst = top.trace_closest(o, d)
        torch.cuda.synchronize()
        begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        begin.record()
        for _ in range(repeats):
            top.trace_closest(o, d, hits=hits, inst=inst)
        end.record()
        torch.cuda.synchronize()
        ms = begin.elapsed_time(end) / repeats
        nodes, tris = c[1] / c[0], c[2] / c[0]
        per_ray = RAY_IO_BYTES + nodes * NODE_BYTES + tris * TRIANGLE_BYTES
        gbs = rays * per_ray / (ms * 1e-3) / 1e9
        out.append({"triangles": n, "rays": rays, "build_ms": float(blas.stats().buildMilliseconds),
                    "mrays_per_s": rays / (ms * 1e-3) / 1e6, "launch_ms": ms, "nodes_per_ray": nodes,
                    "triangles_per_ray": tris, "bytes_per_ray": per_ray, "frac": gbs / HBM_PEAK_GBS,
                    "achieved_gbs": gbs})
        del top, blas, positions, indices, o, d, hits, inst
        torch.cuda.empty_cache()
    return out


def launch_ranks(args):
    """Starts args.gpus ranks of this script as child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
    their environment, as torch.distributed.run would set them), relays rank 0's stdout (the one JSON line) to
    this process's stdout and every other stream to stderr.  Returns the exit status: non-zero when any rank
    failed; the ranks still running at that point are ended.  This process never initialises the GPU."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ranks = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        ranks.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if rank == 0 else sys.stderr.fileno()))
    # rank 0 writes its line at the very end; it is read beside the watch over the ranks, so that a rank that
    # dies early cannot leave this process waiting for a line that will never come
    import threading
    lines = []
    reader = threading.Thread(target=lambda: lines.append(ranks[0].stdout.read()), daemon=True)
    reader.start()
    status = 0
    while any(child.poll() is None for child in ranks):
        for rank, child in enumerate(ranks):
            code = child.poll()
            if code not in (None, 0) and status == 0:
                print(f"bench.py: rank {rank} ended with status {code}", file=sys.stderr)
                status = code if code > 0 else 1
                for other in ranks:   # a rank that lost a peer would wait in its next collective for ever
                    if other.poll() is None:
                        other.kill()
        time.sleep(0.05)
    for rank, child in enumerate(ranks):
        if child.returncode != 0 and status == 0:
            print(f"bench.py: rank {rank} ended with status {child.returncode}", file=sys.stderr)
            status = child.returncode if child.returncode > 0 else 1
    reader.join(timeout=10.0)
    line = lines[0] if lines else b""
    if status == 0 and not line.strip():
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        status = 1
    if status == 0:
        sys.stdout.buffer.write(line)
        sys.stdout.flush()
    return status


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    # Libraries under this process write to the C stdout (RCCL prints a version banner there when a
    # communicator comes up).  The one JSON line must be all that stdout carries: everything else goes to
    # stderr, and the line is written to the saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from raytracer_amd import distributed as D
    from raytracer_amd import engine as eng
    from raytracer_amd import renderer as R
    from raytracer_amd import workloads

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    local %= torch.cuda.device_count()  # more ranks than GPUs only in a gloo rehearsal
    torch.cuda.set_device(local)
    wire = "cuda" if args.backend == "nccl" else "cpu"  # where the bench's own small collectives live
    if world > 1 or args.shared:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    R.load()
    m = run_workload(args, args.workload, args.steps, args.warmup, world, rank, wire)
    profile, shared = m["profile"], m["shared"]
    elapsed, rays, total_rays, total_gathers, per_rank = m["elapsed"], m["rays"], m["total_rays"], m["total_gathers"], m["per_rank"]
    stats, stages, staged_steps, staged_rays = m["stats"], m["stages"], m["staged_steps"], m["staged_rays"]
    batch, in_flight, exchanges, tile_allgather_ms = m["batch"], m["in_flight"], m["exchanges"], m["tile_allgather_ms"]
    bytes_per_ray, nodes_per_ray, tris_per_ray = m["bytes_per_ray"], m["nodes_per_ray"], m["tris_per_ray"]

    # Beside the headline, outside its timed region and at N = 1 only: the other workload north_star names
    # (caustic-lab, final profile, photon map at the reference's sizes) and the triangle soups of the last
    # configuration, whose build time is a reported metric.
    legs = {}
    if world == 1 and not args.shared and not args.no_extra_legs and args.workload == "dragon-cornell":
        legs["caustic_lab"] = caustic_lab_leg(args, wire)
        legs["soup"] = soup_leg(eng, args.soup_sizes)

    photon = None
    if not args.no_photon_bench:
        cpu_seconds = 3.0 if (not args.no_cpu_baseline and world == 1) else 0.0
        photon = photon_benchmark(eng, workloads, args, rank, max(3, min(args.steps, 10)), cpu_seconds)
    ceiling = copy_ceiling_gbs() if rank == 0 else None

    if rank != 0:
        dist.destroy_process_group()
        return

    trace_ms, trace_launches = stages["trace"]
    profiled_workload = args.workload == "dragon-cornell" and (args.width, args.height) == (1920, 1080) and world == 1
    traffic = profiled_traffic() if profiled_workload else None
    achieved = staged_rays * bytes_per_ray / (trace_ms * 1e-3) / 1e9
    avg_launch_s = trace_ms * 1e-3 / max(trace_launches, 1)
    if shared:
        parallelism = (f"replicated scene; one frame shared by {world} ranks, bands of 8 pixel rows dealt out in turn"
                       + ("; photon emission ranges per rank, lists exchanged rank to rank over RCCL every 4th step"
                          if profile else "; no collective inside a step"))
    else:
        parallelism = "one GPU"
    out = {
        "metric": "closest_hit_mrays_per_s",
        "value": total_rays / elapsed / 1e6,
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",  # the total work is fixed: N GPUs share the one frame a single GPU renders alone
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload} stand-in, {args.width}x{args.height}, 1 sample per pixel per step, "
                        f"{'final (32 bounces, photon map)' if profile else 'preview (5 bounces)'}, wavefront "
                        f"generate/trace/shade/accumulate through rt::Renderer::render()",
            "rays_per_step": total_rays / args.steps,
            "rays_per_step_per_gpu": rays / args.steps,
            "extend_rays_per_step": stats["extend_rays"] / args.steps,
            "shadow_rays_per_step": stats["shadow_rays"] / args.steps,
            "triangles": m["scene_triangles"],
            "instances": m["instances"],
            "light_count": m["light_count"],
            "parallelism": parallelism,
        },
        "stage_ms_per_step": {k: v[0] / staged_steps for k, v in stages.items()},
        "trace_kernel_mrays_per_s": staged_rays / (trace_ms * 1e-3) / 1e6,
        "concurrent_tiles": args.tiles,
        "samples_in_flight": in_flight,
        "samples_per_pass": batch,
        "gather_queries_per_step": total_gathers / args.steps,
        "roofline": {
            "bound": "hbm",
            "kernel": "traceKernel (closest-hit and shadow rays of a bounce)",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic[0] if traffic else None,
            "traffic_source": (f"bytes per launch, rocprofv3 --pmc FETCH_SIZE (x2) + WRITE_SIZE, {traffic[1]}, taken with "
                               f"the library this run loaded" if traffic else
                               "no committed PMC summary was taken with the library this run loaded" if profiled_workload else
                               "the committed PMC summaries are of the default workload on one GPU"),
            "traffic_gbs": traffic[0] / avg_launch_s / 1e9 if traffic else None,
            "traffic_frac_of_peak": traffic[0] / avg_launch_s / 1e9 / HBM_PEAK_GBS if traffic else None,
            "copy_ceiling_gbs": ceiling,
            "frac_of_copy_ceiling": achieved / ceiling if ceiling else None,
            "algorithmic_bytes_per_launch": staged_rays * bytes_per_ray / max(trace_launches, 1),
            "bytes_per_ray": bytes_per_ray,
            "nodes_per_ray": nodes_per_ray,
            "triangles_per_ray": tris_per_ray,
            "avg_launch_ms": trace_ms / max(trace_launches, 1),
            "launches": trace_launches,
            "library_sha256": library_hash(),
        },
    }
    if shared:
        mean = sum(per_rank) / world
        out["multi_gpu"] = {
            "rays_per_rank": per_rank,
            "imbalance_max_over_mean": max(per_rank) / mean if mean else None,
            "tile_allgather_ms": tile_allgather_ms,
            "tile_allgather_bytes": args.width * args.height * 16,
            "photon_exchanges": len(exchanges),
            "photon_allgather_ms": float(np.mean([e["ms"] for e in exchanges])) if exchanges else None,
            "photon_allgather_bytes_received": float(np.mean([e["bytes_received"] for e in exchanges])) if exchanges else None,
            "photons_per_pass": float(np.mean([e["photons"] for e in exchanges])) if exchanges else None,
        }
    if photon is not None:
        out["photon_gathers_per_s"] = photon["gathers_per_s"]
        out["photon"] = photon
    if m["photon_pass_ms"] is not None:
        out["photon_pass_ms"] = m["photon_pass_ms"]
        out["photons_stored_per_pass"] = m["stored_photons"]
    if legs:
        out.update(legs)
    if not args.no_cpu_baseline and world == 1:  # the CPU leg is reported at N = 1 only
        out["cpu_baseline"] = cpu_baseline(m["scene"], args, args.width, args.height, profile)
        if photon is not None and photon.get("cpu_baseline"):
            out["cpu_baseline"]["gather"] = photon.pop("cpu_baseline")
    if photon is not None:
        photon.pop("cpu_baseline", None)
    sys.stdout.flush()
    os.write(result_fd, (json.dumps(out) + "\n").encode())
    os.close(result_fd)
    if shared:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
