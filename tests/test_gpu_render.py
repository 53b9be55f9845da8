"""GPU tests of the wavefront integrator and the photon tracer behind rt::Renderer,
against the oracle's scalar restatement of the reference's device programs.

Tolerance: libm (host) and the device math library round sinf/cosf/powf/... differently in the last
place, so a path can take a different branch than its oracle twin; those are rare, the rest agree to
float rounding.  Stated per test: the share of pixels within 1e-3 relative error, and the relative
RMSE of the whole image at equal sample count."""
import numpy as np
import pytest

from raytracer_amd import workloads
from helpers import HostParams, oracle_scene, rel_rmse

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R(engine):
    from raytracer_amd import renderer
    renderer.load()
    return renderer


def render_both(R, oracle, scene, width, height, samples, profile=0, threads=8):
    r = R.Renderer(scene, width, height, profile)
    ref = oracle_scene(oracle, scene)
    host = HostParams(r, scene)
    for s in range(samples):
        r.render()
        host.params.sample = s
        ref.render_sample(host.params, 0, width * height, threads)
    r.synchronize()
    gpu = r.linear_pixels().reshape(-1, 4)
    return r, host, gpu, host.aov[0]


def close_fraction(a, b, rel=1e-3, floor=1e-4):
    err = np.abs(a[:, :3].astype(np.float64) - b[:, :3])
    tol = rel * np.maximum(np.abs(b[:, :3]), floor)
    return float((err <= tol).all(axis=1).mean())


def test_cornell_point_light_matches_oracle(R, oracle):
    scene = workloads.cornell_box()
    r, host, gpu, cpu = render_both(R, oracle, scene, 256, 256, 1)
    assert r.samples() == 1
    assert (gpu[:, 3] == 1.0).all()
    assert cpu[:, :3].mean() > 0.01
    assert close_fraction(gpu, cpu) > 0.995
    assert rel_rmse(gpu[:, :3], cpu[:, :3]) < 2e-2
    # the byte image follows the same tone curve
    diff = np.abs(r.pixels().reshape(-1, 4).astype(int) - host.output.astype(int))
    assert (diff <= 1).mean() > 0.995
    stats = r.statistics()
    assert stats["extend_rays"] >= 256 * 256 and stats["shadow_rays"] > 0 and stats["gather_queries"] == 0


def test_ragged_frame_sizes_match_oracle(R, oracle):
    """Widths that are not a multiple of 8 and pixel counts that are not a multiple of the workgroup size
    (the tile numbering of paths falls back to row order; the last workgroups are partly empty)."""
    scene = workloads.cornell_box()
    for w, h in ((50, 37), (72, 13), (1, 1)):
        r, host, gpu, cpu = render_both(R, oracle, scene, w, h, 1)
        assert close_fraction(gpu, cpu) > 0.99, (w, h)
        assert (gpu[:, 3] == 1.0).all()


def test_cornell_golden_image(R):
    """The committed oracle image of config 1 (Cornell, 1 spp, depth 5; 128 x 128 to keep the fixture small)."""
    import os
    golden = np.load(os.path.join(os.path.dirname(__file__), "golden", "cornell_128_1spp.npz"))["accumulation"]
    r = R.Renderer(workloads.cornell_box(), 128, 128, 0)
    r.render()
    gpu = r.linear_pixels().reshape(-1, 4)
    assert close_fraction(gpu, golden.reshape(-1, 4)) > 0.995


def test_area_light_mis_several_samples(R, oracle):
    scene = workloads.cornell_box(light="panel")
    r, host, gpu, cpu = render_both(R, oracle, scene, 96, 96, 4)
    assert close_fraction(gpu, cpu) > 0.98
    assert rel_rmse(gpu[:, :3], cpu[:, :3]) < 5e-2
    for k in range(1, 7):  # every accumulation buffer, not only the beauty image
        from raytracer_amd import renderer as Rm
        dev = Rm.device_to_host(r.accumulation_buffer(k), 96 * 96 * 16).view(np.float32).reshape(-1, 4)
        assert close_fraction(dev, host.aov[k]) > 0.97, k


def test_textures_environment_lens_glass_metal(R, oracle):
    scene = workloads.textured_sky_scene()
    r, host, gpu, cpu = render_both(R, oracle, scene, 128, 72, 2)
    assert cpu[:, :3].mean() > 0.05
    assert close_fraction(gpu, cpu, rel=2e-3) > 0.95
    # bright sun samples make single pixels heavy; compare the robust bulk
    keep = cpu[:, :3].max(axis=1) < np.percentile(cpu[:, :3].max(axis=1), 99)
    assert rel_rmse(gpu[keep, :3], cpu[keep, :3]) < 0.1


def test_photon_pass_matches_oracle(R, oracle):
    """The whole photon list against the oracle's, aligned by emission index.  The engine appends in
    emission order, so tracing the pass in blocks of 128 emissions and reading the stored count after each
    block tells which photons came from which block; the oracle reports the emission of each of its
    photons.  Blocks in which both sides stored the same number of photons are compared photon by photon
    (a path that forks on a last-place difference of sinf / cosf / powf changes a block's count, or one
    photon of it).  Tolerance: at least 99 % of the blocks agree in count, at least 99 % of the photons
    in those blocks agree to 1e-3 relative in position, power and normal, and the totals within 0.2 %."""
    scene = workloads.caustic_lab()
    emissions, block = 1 << 17, 128
    r = R.Renderer(scene, 64, 36, 1)
    r.set_photon_budget(emissions, 1 << 16, 1 << 14)
    ref = oracle_scene(oracle, scene)
    host = HostParams(r, scene)
    r.begin_photon_pass()
    ends = [r.trace_photon_range(first, block) for first in range(0, emissions, block)]
    stored = ends[-1]
    gpu = R.device_to_host(r.photon_buffer(), stored * 36).view(np.float32).reshape(-1, 9)
    host.params.photonRadius = r.caustic_radius()
    cpu, emission, wanted = ref.trace_photons_indexed(host.params, 0, emissions, 1 << 16, threads=8)
    assert wanted > 1000 and wanted <= 1 << 16
    assert abs(stored - wanted) <= max(4, wanted // 500)
    assert (np.diff(emission.astype(np.int64)) > 0).all()
    gpu_counts = np.diff(np.array([0] + ends))
    cpu_counts = np.bincount(emission // block, minlength=emissions // block)
    agree = gpu_counts == cpu_counts
    assert agree.mean() > 0.99
    gpu_start = np.concatenate([[0], np.cumsum(gpu_counts)[:-1]])
    cpu_start = np.concatenate([[0], np.cumsum(cpu_counts)[:-1]])
    close = total = 0
    for b in np.nonzero(agree & (gpu_counts > 0))[0]:
        g = gpu[gpu_start[b]:gpu_start[b] + gpu_counts[b]]
        c = cpu[cpu_start[b]:cpu_start[b] + cpu_counts[b]]
        close += int(np.isclose(g, c, rtol=1e-3, atol=1e-6).all(axis=1).sum())
        total += len(g)
    assert total > 0.98 * wanted and close > 0.99 * total, (close, total, wanted)
    # sharded emission ranges append in emission order: two halves equal one pass
    r.begin_photon_pass()
    a = r.trace_photon_range(0, 1 << 16)
    b = r.trace_photon_range(1 << 16, 1 << 16)
    assert b == stored and a <= b
    again = R.device_to_host(r.photon_buffer(), stored * 36).view(np.float32).reshape(-1, 9)
    assert np.array_equal(again, gpu)
    # the staging area takes the same list (what a multi-GPU driver sends from)
    r.begin_photon_pass()
    assert r.trace_photon_range(0, emissions, staging=True) == stored
    staged = R.device_to_host(r.photon_staging_buffer(), stored * 36).view(np.float32).reshape(-1, 9)
    assert np.array_equal(staged, gpu)


def test_host_tables_match_the_independent_set_up(R):
    """What the host layer derives from a scene - material records, light list and total weight,
    environment distribution and power, scene sphere, camera frame, caustic radius - against the numpy
    restatement of the reference's renderer.cpp in tests/ref_setup.py.  Sums, products, divisions and
    square roots are the same IEEE operations on both sides and must agree bit for bit; values that go
    through sin / cos / tan (environment rows, spot cones, field of view) may differ in the last place."""
    from ref_setup import Setup
    import copy
    lit = workloads.cornell_box()
    lit["lights"] = [
        dict(transform=workloads.translation(-0.6, 1.7, 0.4), color=(1.0, 0.3, 0.2), intensity=2.0, range=1.2,
             innerCone=0.0, outerCone=0.7853982, type=2),
        dict(transform=workloads.look_at([0.5, 1.9, 0.5], [0.3, 0.0, -0.2]), color=(0.2, 0.6, 1.0), intensity=30.0,
             range=2.5, innerCone=0.2, outerCone=0.4, type=1),
        dict(transform=workloads.look_at([0.0, 1.9, 0.9], [0.0, 0.0, -0.9]), color=(1.0, 1.0, 0.8), intensity=0.6,
             range=float("inf"), innerCone=0.0, outerCone=0.7853982, type=0)]
    scenes = [("cornell panel", workloads.cornell_box(light="panel"), 0), ("punctual lights", lit, 0),
              ("sky, textures, lens", workloads.textured_sky_scene(), 0), ("sphere field", workloads.sphere_field(sky=True), 0),
              ("material maps", workloads.material_map_scene(), 0), ("caustic lab", workloads.caustic_lab(), 1),
              ("emissive mesh", copy.deepcopy(workloads.dragon_cornell(triangles=20000)), 0)]
    scenes[-1][1]["materials"][-1].update(emissive=(0.2, 0.5, 1.0), emissiveStrength=3.0)  # 20 k emissive triangles
    for name, scene, profile in scenes:
        w, h = 160, 90
        r = R.Renderer(scene, w, h, profile)
        p = r.launch_params()
        s = Setup(scene, w, h, final=bool(profile))
        q = s.params
        bits = lambda a: np.asarray(a, dtype=np.float32).view(np.uint32)
        vec = lambda v: np.array([v.x, v.y, v.z], dtype=np.float32)
        # materials: byte for byte
        mats = R.device_to_host(p.materials, 272 * len(scene["materials"]))
        assert bytes(mats) == bytes(s.materials), name
        # lights
        assert p.lightCount == q.lightCount, name
        if p.lightCount:
            mine = R.device_to_host(p.lights, 92 * p.lightCount).view(np.uint32).reshape(-1, 23)
            want = np.frombuffer(bytes((R.GpuLight * len(s.lights))(*s.lights)), dtype=np.uint32).reshape(-1, 23)
            assert np.array_equal(mine[:, 20:], want[:, 20:]), name                       # instance, primitive, type
            exact = np.ones(23, dtype=bool)
            exact[16] = False                                                              # weight: solid angle has a cosine
            exact[20:] = False
            punctual = want[:, 22] < 3
            emissive = want[:, 22] == 3
            assert np.array_equal(mine[~punctual][:, exact], want[~punctual][:, exact]), name
            assert np.array_equal(mine[emissive][:, 16], want[emissive][:, 16]), name       # area x luminance: exact
            sky = want[:, 22] == 4                                                          # the map's power has sines in it
            assert np.allclose(mine[sky].view(np.float32)[:, 16], want[sky].view(np.float32)[:, 16], rtol=2e-6), name
            assert np.allclose(mine[punctual].view(np.float32)[:, :20], want[punctual].view(np.float32)[:, :20],
                               rtol=3e-7, atol=0, equal_nan=True), name
        assert np.isclose(p.lightWeight, q.lightWeight, rtol=1e-6), name
        # environment
        assert (p.environmentWidth, p.environmentHeight) == (q.environmentWidth, q.environmentHeight), name
        if p.environmentWidth:
            n_env = p.environmentWidth * p.environmentHeight
            cdf = R.device_to_host(p.environmentCdf, 4 * n_env).view(np.float32)
            assert np.allclose(cdf, s.environment_cdf, rtol=2e-6, atol=1e-9), name
            assert cdf[-1] == 1.0 and (np.diff(cdf) >= 0).all(), name
            texels = R.device_to_host(p.environment, 16 * n_env).view(np.float32)
            assert np.array_equal(texels, np.ascontiguousarray(scene["environment"]["pixels"], dtype=np.float32).reshape(-1)), name
            assert np.isclose(p.environmentWeight, q.environmentWeight, rtol=2e-6), name
        # scene sphere: exact
        assert np.array_equal(bits(vec(p.sceneCenter)), bits(vec(q.sceneCenter))), name
        assert bits([p.sceneRadius])[0] == bits([q.sceneRadius])[0], name
        # camera frame and caustic radius (tan of the field of view)
        for field in ("eye", "cameraW", "lensU", "lensV"):
            assert np.array_equal(bits(vec(getattr(p, field))), bits(vec(getattr(q, field)))), (name, field)
        for field in ("cameraU", "cameraV"):
            assert np.allclose(vec(getattr(p, field)), vec(getattr(q, field)), rtol=3e-7, atol=0), (name, field)
        assert np.isclose(r.caustic_radius(), s.caustic_radius, rtol=1e-6), name
        assert (p.photonRadius > 0) == bool(profile) and p.maxDepth == (32 if profile else 5), name
        assert (p.photonEmissions, p.photonCapacity, p.photonBucketCount) == (1 << 24, 1 << 23, 1 << 22), name


def test_final_profile_caustics(R, oracle):
    scene = workloads.caustic_lab()
    w, h = 96, 54
    r = R.Renderer(scene, w, h, 1)
    r.set_photon_budget(1 << 18, 1 << 17, 1 << 15)
    ref = oracle_scene(oracle, scene)
    host = HostParams(r, scene)
    radius = r.caustic_radius()
    for s in range(2):
        r.render()
        if s % 4 == 0:  # renderer.cpp:780: a photon pass before samples 0, 4, 8, ...
            host.params.photonRadius = radius
            host.params.photonPass = s // 4
            ph, wanted = ref.trace_photons(host.params, 0, 1 << 18, 1 << 17, threads=8)
            offsets, srt = oracle.build_photon_grid(ph, radius, 1 << 15)
            host.sorted[: len(srt)] = srt
            host.cell_start[:] = offsets
        host.params.sample = s
        ref.render_sample(host.params, 0, w * h, 8)
    r.synchronize()
    gpu = r.linear_pixels().reshape(-1, 4)
    from raytracer_amd import renderer as Rm
    caustic_gpu = Rm.device_to_host(r.accumulation_buffer(4), w * h * 16).view(np.float32).reshape(-1, 4)
    caustic_cpu = host.aov[4]
    assert caustic_cpu[:, :3].sum() > 0 and caustic_gpu[:, :3].sum() > 0
    # photon sets differ in a few members (transcendental rounding), so the density estimate is
    # compared in aggregate: total caustic energy within 3 %, image relRMSE bounded
    assert abs(caustic_gpu[:, :3].sum() / caustic_cpu[:, :3].sum() - 1.0) < 0.03
    assert rel_rmse(gpu[:, :3], host.aov[0][:, :3]) < 0.15
    stats = r.statistics()
    assert stats["gather_queries"] > 0


def test_heavy_photon_queries_give_the_same_floats(R, oracle):
    """A small frame has a wide gather radius: with a dense photon map most queries read thousands of
    photons and go to a wave each instead of staying in their lane.  Both ways add the photons inside
    the radius in index order, so the images agree bit for bit, the counters too; the oracle's caustic
    buffer is matched as closely as the photon sets allow (they differ in a few members)."""
    scene = workloads.caustic_lab()
    w, h = 48, 27
    images, stats = [], []
    for per_lane in (False, True):
        r = R.Renderer(scene, w, h, 1)
        r.set_photon_budget(1 << 22, 1 << 21, 1 << 14)
        r.set_measurement(gather_per_lane=per_lane)
        for _ in range(2):
            r.render()
        r.synchronize()
        images.append(np.stack([R.device_to_host(r.accumulation_buffer(k), w * h * 16).view(np.uint32) for k in range(7)]))
        stats.append(r.statistics())
    assert np.array_equal(images[0], images[1])
    assert stats[0] == stats[1]
    assert stats[0]["gathered_photons"] > 600 * stats[0]["gather_queries"]  # well over the hand-over threshold
    caustic = images[0][4].view(np.float32).reshape(-1, 4)
    assert caustic[:, :3].sum() > 0 and np.isfinite(caustic).all()


def test_sample_launch_through_the_c_abi(R, engine):
    """rt_render_sample called directly on a wavefront state of the caller's own, with the launch
    parameters of a Renderer: without bucket bits the gather reads every range (the 27-step walk, heavy
    queries included) and the sample is the one the Renderer adds itself; launch parameters that switch
    the gather on without a photon grid are refused before anything is launched."""
    import ctypes as C
    w, h = 48, 27
    r = R.Renderer(workloads.caustic_lab(), w, h, 1)
    r.set_photon_budget(1 << 21, 1 << 20, 1 << 14)
    r.set_concurrent_tiles(1)
    r.render()                     # photon pass 0 and sample 0
    r.synchronize()
    first = [R.device_to_host(r.accumulation_buffer(k), w * h * 16).copy() for k in range(7)]
    lib = engine.load()
    state = C.c_void_p(0)
    engine.check(lib.rt_wavefront_create(w * h, C.byref(state)))
    try:
        params = r.launch_params()   # sample index 0 again: the running mean of x and x is x
        params.sample = 0
        engine.check(lib.rt_render_sample(state, C.byref(params), 0, w * h, None))
        import torch
        torch.cuda.synchronize()
        again = [R.device_to_host(r.accumulation_buffer(k), w * h * 16) for k in range(7)]
        for k in range(7):
            assert np.array_equal(first[k], again[k]), k
        for field, value in (("sortedPhotons", None), ("photonCellStart", None), ("photonBucketCount", 0),
                             ("photonBucketCount", 1000)):
            broken = R.LaunchParams.from_buffer_copy(params)
            setattr(broken, field, value)
            with pytest.raises(engine.EngineError, match="photon grid"):
                engine.check(lib.rt_render_sample(state, C.byref(broken), 0, w * h, None))
    finally:
        lib.rt_wavefront_destroy(state)


def test_camera_state_without_the_photon_query_arrays(R, engine):
    """A state created with RT_WAVEFRONT_FOR_CAMERA_NO_CAUSTICS leaves out the four query arrays, the heavy-query
    list and the caustic sum (84 bytes a path), renders a sample without a photon map bit for bit as a full state
    does, and refuses launch parameters with a gather radius.  The Renderer takes such states in its preview
    profile and full ones in the final profile."""
    import ctypes as C
    import torch
    lib = engine.load()
    camera, photons, plain = 1, 2, 4
    paths = 1 << 20
    sizes = {}
    for uses in (camera, photons, plain, camera | plain, camera | photons, camera | photons | plain):
        n = C.c_size_t(0)
        engine.check(lib.rt_wavefront_state_bytes(paths, uses, C.byref(n)))
        sizes[uses] = n.value
    assert sizes[camera] - sizes[plain] == 84 * paths
    assert sizes[camera | plain] == sizes[camera] and sizes[camera | photons | plain] == sizes[camera | photons]
    for uses in (0, 8, 15):
        with pytest.raises(engine.EngineError, match="uses"):
            engine.check(lib.rt_wavefront_state_bytes(paths, uses, C.byref(C.c_size_t(0))))
    w, h = 48, 27
    r = R.Renderer(workloads.caustic_lab(), w, h, R.PREVIEW)
    r.set_concurrent_tiles(1)
    r.render()
    r.synchronize()
    first = [R.device_to_host(r.accumulation_buffer(k), w * h * 16).copy() for k in range(7)]
    state = C.c_void_p(0)
    engine.check(lib.rt_wavefront_create_for(w * h, plain, C.byref(state)))
    try:
        params = r.launch_params()
        params.sample = 0
        assert params.photonRadius == 0.0
        # an empty launch ahead of the first pass (the queue's scratch set-up) changes nothing
        engine.check(lib.rt_wavefront_prepare_stream(state, None))
        engine.check(lib.rt_render_sample(state, C.byref(params), 0, w * h, None))
        torch.cuda.synchronize()
        for k in range(7):
            assert np.array_equal(first[k], R.device_to_host(r.accumulation_buffer(k), w * h * 16)), k
        with_map = R.LaunchParams.from_buffer_copy(params)
        with_map.photonRadius = 0.01
        with pytest.raises(engine.EngineError, match="photonRadius"):
            engine.check(lib.rt_render_sample(state, C.byref(with_map), 0, w * h, None))
    finally:
        lib.rt_wavefront_destroy(state)
    # the same renderer in the final profile: its states grow by the arrays of the gather, and a photon pass
    # started by hand in the preview profile gets them as well
    r.set_profile(R.FINAL)
    r.render()
    r.synchronize()
    assert r.statistics()["gather_queries"] > 0
    r.set_profile(R.PREVIEW)
    r.render()
    r.synchronize()
    r.begin_photon_pass()
    stored = r.trace_photon_range(0, r.photon_emissions())
    r.finish_photon_pass(stored)
    r.render()
    r.synchronize()
    assert r.statistics()["gather_queries"] > 0


def test_default_schedule_follows_the_state_budget(R):
    """What the renderer picks by itself on a GPU with memory to spare: the most samples per pass (64 at most) whose
    state fits 24 GiB; two passes in flight where two passes of 64 samples fit (a quarter of a 1080p frame), else -
    without a photon map - one pass of twice the samples (the whole frame: 32 x 1, half of it: 64 x 1); with a
    photon map two passes stay (16 x 2, four photon passes traced together)."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < (80 << 30):
        pytest.skip("the defaults are cut to half of the free memory; this test states them for a free GPU")
    scene = workloads.cornell_box()
    for shares, expected in ((1, (32, 1)), (2, (64, 1)), (4, (64, 2)), (8, (64, 2))):
        r = R.Renderer(scene, 1920, 1080, R.PREVIEW)
        r.set_row_interleave(0, shares)   # the state is made for the share, with the first question about it
        assert (r.sample_batch(), r.samples_in_flight()) == expected, shares
        del r
    r = R.Renderer(scene, 1920, 1080, R.PREVIEW)
    r.set_concurrent_tiles(1, 2)   # a number the caller names is kept
    assert (r.sample_batch(), r.samples_in_flight()) == (16, 2)
    del r
    final = R.Renderer(workloads.caustic_lab(), 1920, 1080, R.FINAL)
    assert (final.sample_batch(), final.samples_in_flight(), final.photon_passes_together()) == (16, 2, 4)


def test_tiny_photon_budgets(R, oracle):
    """Photon budgets at the lower end: one bucket (every neighbour cell is the same bucket), two buckets,
    a capacity of one photon (the buffer overflows and the map keeps what fits), one emission.  The
    buffers stay finite, no reference is refused, and the dense one- and two-bucket maps still light the
    caustic buffer."""
    scene = workloads.caustic_lab()
    w, h = 40, 24
    for emissions, capacity, buckets in ((1 << 16, 1 << 15, 1), (1 << 16, 1 << 15, 2), (1 << 14, 1, 64), (1, 1, 1)):
        r = R.Renderer(scene, w, h, 1)
        r.set_photon_budget(emissions, capacity, buckets)
        for _ in range(2):
            r.render()
        r.synchronize()
        buffers = np.stack([R.device_to_host(r.accumulation_buffer(k), w * h * 16).view(np.float32) for k in range(7)])
        assert np.isfinite(buffers).all(), (emissions, capacity, buckets)
        stats = r.statistics()
        assert stats["refused_references"] == 0 and stats["gather_queries"] > 0
        if buckets <= 2 and emissions > 1:
            assert stats["gathered_photons"] > 0 and buffers[4].reshape(-1, 4)[:, :3].sum() > 0


def test_reference_error_behaviour(R):
    scene = workloads.cornell_box()
    empty = dict(scene, instances=[])
    with pytest.raises(Exception, match="no geometry"):
        R.Renderer(empty, 8, 8)
    nocam = dict(scene, cameras=[])
    with pytest.raises(Exception, match="no perspective camera"):
        R.Renderer(nocam, 8, 8)
    r = R.Renderer(scene, 8, 8)
    with pytest.raises(Exception, match="denois"):
        r.denoise()
    # references from one table of the scene into another are checked before anything reaches the device
    import copy
    broken = copy.deepcopy(scene)
    broken["instances"][0] = (99, broken["instances"][0][1])
    with pytest.raises(Exception, match="geometry the scene does not have"):
        R.Renderer(broken, 8, 8)
    broken = copy.deepcopy(scene)
    broken["geometries"][0]["material"] = 99
    with pytest.raises(Exception, match="material the scene does not have"):
        R.Renderer(broken, 8, 8)
    broken = copy.deepcopy(scene)
    broken["geometries"][0]["indices"] = broken["geometries"][0]["indices"].copy()
    broken["geometries"][0]["indices"].reshape(-1)[1] = 1 << 20
    with pytest.raises(Exception, match="vertex its geometry does not have"):
        R.Renderer(broken, 8, 8)
    broken = copy.deepcopy(scene)
    broken["materials"][0] = dict(broken["materials"][0], baseColorTexture=dict(texture=3, texCoord=0))
    with pytest.raises(Exception, match="texture the scene does not have"):
        R.Renderer(broken, 8, 8)


def test_set_camera_and_profile_reset_accumulation(R):
    scene = workloads.cornell_box()
    r = R.Renderer(scene, 32, 32)
    r.render()
    r.render()
    assert r.samples() == 2
    r.set_camera(scene["cameras"][0])
    assert r.samples() == 0 and not r.linear_pixels().any()
    r.render()
    r.set_profile(1)
    assert r.samples() == 0 and not r.linear_pixels().any()


def test_pixel_ranges_compose_to_the_full_frame(R):
    """Tile sharding: rendering two pixel ranges gives the same buffers as one full-frame launch."""
    scene = workloads.cornell_box(light="panel")
    w, h = 64, 64
    full = R.Renderer(scene, w, h)
    full.render()
    full.render()
    parts = R.Renderer(scene, w, h)
    for s in range(2):
        # both halves of sample s: the sample counter advances per render(), so drive it per range
        parts.set_pixel_range(0, w * h // 3)
        parts.render()
    a = parts.linear_pixels().reshape(-1, 4)[: w * h // 3]
    assert np.array_equal(a, full.linear_pixels().reshape(-1, 4)[: w * h // 3])


def test_interleaved_bands_compose_to_the_full_frame(R):
    """The bands of eight rows dealt out to 1, 2, 3 and 8 takers, each rendering its share (as the ranks of
    a multi-GPU job do), give the full frame bit for bit: heights with a short last band, fewer bands
    than takers, a width that is no multiple of 8 (row order inside the bands), 1 and 3 concurrent parts."""
    scene = workloads.cornell_box(light="panel")
    for w, h in ((64, 72), (96, 54), (50, 37), (64, 12)):
        full = R.Renderer(scene, w, h)
        for _ in range(2):
            full.render()
        want = full.linear_pixels().reshape(-1, 4)
        want_bytes = full.pixels().reshape(-1, 4)
        rays = full.statistics()["extend_rays"]
        for takers, tiles in ((2, 3), (3, 1), (8, 2)):
            got = np.zeros_like(want)
            got_bytes = np.zeros_like(want_bytes)
            total = 0
            for first in range(takers):
                part = R.Renderer(scene, w, h)
                part.set_concurrent_tiles(tiles)
                part.set_row_interleave(first, takers)
                for _ in range(2):
                    part.render()
                rows = np.zeros(h, dtype=bool)
                for band in range(first, (h + 7) // 8, takers):
                    rows[band * 8:band * 8 + 8] = True
                mask = np.repeat(rows, w)
                image = part.linear_pixels().reshape(-1, 4)
                assert not image[~mask].any(), "a taker wrote outside its bands"
                got[mask] = image[mask]
                got_bytes[mask] = part.pixels().reshape(-1, 4)[mask]
                total += part.statistics()["extend_rays"]
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (w, h, takers)
            assert np.array_equal(got_bytes, want_bytes) and total == rays
    r = R.Renderer(scene, 32, 32)
    from raytracer_amd import engine
    for bad in ((0, 0), (2, 2), (5, 3)):
        with pytest.raises(engine.EngineError):
            r.set_row_interleave(*bad)


def test_render_into_and_copy_to_caller_owned_device_buffers(R):
    """render(output) writes the byte image into a caller-owned device buffer (the interactive loop of the
    reference hands its mapped display buffer over, main.cpp:248-249) and copyOutput() copies the
    renderer's own byte image device to device (renderer.cpp:888-891): both equal pixels()."""
    import torch
    scene = workloads.cornell_box(light="panel")
    w, h = 80, 48
    own = R.Renderer(scene, w, h)
    for _ in range(3):
        own.render()
    want = own.pixels()
    copied = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    own.copy_output(copied)
    assert np.array_equal(copied.cpu().numpy(), want)
    target = torch.full((h, w, 4), 7, dtype=torch.uint8, device="cuda")
    other = R.Renderer(scene, w, h)
    for _ in range(3):
        other.render(target)
    other.synchronize()
    assert np.array_equal(target.cpu().numpy(), want)
    assert np.array_equal(other.linear_pixels(), own.linear_pixels())
    # the renderer's own byte image was never written by the second renderer; its view is the zero-copy one
    view = R.device_view(own.output_buffer(), w * h * 4).view(h, w, 4)
    assert np.array_equal(view.cpu().numpy(), want)


def test_concurrent_tiles_do_not_change_the_image(R):
    """render() cut into 1, 2 or 5 concurrent parts (own streams, own wavefront state), with one to eight
    samples of a part in flight (their accumulation steps chained by events), gives bit-identical
    buffers; 72 rows exercise full 8-row bands, leftover rows and parts that start inside a band."""
    scene = workloads.cornell_box(light="panel")
    w, h = 64, 72
    images = []
    default = R.Renderer(scene, w, h)
    # the renderer's own choice: two passes in flight (a frame this small has room for two passes of 64 samples)
    assert (default.concurrent_tiles(), default.samples_in_flight()) == (1, 2)
    assert R.Renderer(scene, w, h, R.FINAL).samples_in_flight() == 2
    default.set_concurrent_tiles(1, 3)
    default.set_concurrent_tiles(1, 0)
    assert default.samples_in_flight() == 2
    for tiles, depth in ((1, 1), (2, 1), (5, 1), (1, 4), (2, 3), (1, 8)):
        r = R.Renderer(scene, w, h)
        r.set_concurrent_tiles(tiles, depth)
        assert r.concurrent_tiles() == tiles and r.samples_in_flight() == depth
        for _ in range(7):
            r.render()
        images.append((r.linear_pixels().copy(), r.pixels().copy(), r.statistics()))
    for pixels, bytes8, stats in images[1:]:
        assert np.array_equal(pixels, images[0][0])
        assert np.array_equal(bytes8, images[0][1])
        assert stats["extend_rays"] == images[0][2]["extend_rays"] and stats["refused_references"] == 0
    # the final profile: every part has photon query queues of its own and shares the photon map
    scene = workloads.caustic_lab()
    w, h = 64, 40
    images = []
    for tiles, depth in ((1, 1), (3, 1), (4, 1), (1, 4), (1, 3), (2, 2)):
        r = R.Renderer(scene, w, h, 1)
        r.set_photon_budget(1 << 19, 1 << 18, 1 << 14)
        r.set_concurrent_tiles(tiles, depth)
        for _ in range(6):           # photon passes before samples 0 and 4
            r.render()
        r.synchronize()
        images.append((np.stack([R.device_to_host(r.accumulation_buffer(k), w * h * 16) for k in range(7)]),
                       r.statistics()))
    for buffers, stats in images[1:]:
        assert np.array_equal(buffers, images[0][0])
        assert stats == images[0][1]
    assert images[0][1]["gather_queries"] > 0


def test_sample_batches_do_not_change_the_image(R):
    """render() calls collected into passes of 1, 2, 3, 4 or 8 samples (one wavefront pass over pixels x samples
    paths, the means taking the samples in order), alone and with passes in flight and a frame cut into parts:
    bit-identical buffers and byte image after 11 samples, the same ray counts, samples() counting calls
    that still wait; the final profile, whose photon passes fall between samples 3 | 4 and 7 | 8 and cut a
    pass short; a band-interleaved share and a pixel range."""
    import torch
    scene = workloads.cornell_box(light="panel")
    w, h = 64, 72
    shots = []
    for batch, tiles, depth in ((1, 1, 1), (2, 1, 1), (3, 1, 1), (4, 1, 1), (8, 1, 1), (4, 1, 3), (2, 2, 2)):
        r = R.Renderer(scene, w, h)
        r.set_concurrent_tiles(tiles, depth)
        r.set_sample_batch(batch)
        assert r.sample_batch() == batch
        for k in range(11):
            r.render()
            assert r.samples() == k + 1
        shots.append((r.linear_pixels().copy(), r.pixels().copy(), r.statistics()))
    for pixels, bytes8, stats in shots[1:]:
        assert np.array_equal(pixels, shots[0][0]) and np.array_equal(bytes8, shots[0][1])
        assert stats["extend_rays"] == shots[0][2]["extend_rays"] and stats["shadow_rays"] == shots[0][2]["shadow_rays"]
    # a caller's own target closes the pass it belongs to
    r = R.Renderer(scene, w, h)
    r.set_sample_batch(4)
    mine = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    for k in range(11):
        r.render(mine if k == 10 else None)
    torch.cuda.synchronize()
    assert np.array_equal(mine.cpu().numpy().reshape(-1), shots[0][1].reshape(-1))
    # final profile: photon passes before samples 0, 4 and 8
    scene = workloads.caustic_lab()
    w, h = 64, 40
    shots = []
    # (a pass of more than four samples gathers from the map of each group of four it holds; passes of 6 start
    # inside a group every other time; 0 is the renderer's own choice)
    for batch, depth in ((1, 1), (4, 1), (3, 2), (8, 4), (16, 2), (6, 1), (6, 3), (32, 1), (0, 0)):
        r = R.Renderer(scene, w, h, 1)
        r.set_photon_budget(1 << 19, 1 << 18, 1 << 14)
        r.set_concurrent_tiles(1, depth)
        r.set_sample_batch(batch)
        for _ in range(23):
            r.render()
        r.synchronize()
        shots.append(np.stack([R.device_to_host(r.accumulation_buffer(k), w * h * 16) for k in range(7)]))
        assert r.statistics()["gather_queries"] > 0
    for buffers in shots[1:]:
        assert np.array_equal(buffers, shots[0])
    # a pass of 64 samples gathers from sixteen maps, the photon passes of which are traced together; 137 samples are
    # two such passes and a short one
    shots = []
    for batch, depth in ((4, 1), (64, 2), (0, 0)):
        r = R.Renderer(scene, w, h, 1)
        r.set_photon_budget(1 << 17, 1 << 16, 1 << 12)
        r.set_concurrent_tiles(1, depth)
        r.set_sample_batch(batch)
        for _ in range(137):
            r.render()
        r.synchronize()
        if batch == 64:
            assert r.photon_passes_together() == 16
        shots.append(np.stack([R.device_to_host(r.accumulation_buffer(k), w * h * 16) for k in range(7)]))
    for buffers in shots[1:]:
        assert np.array_equal(buffers, shots[0])
    # shares of a frame
    scene = workloads.cornell_box(light="panel")
    w, h = 64, 72
    for share in ("bands", "range"):
        shots = []
        for batch in (1, 4):
            r = R.Renderer(scene, w, h)
            r.set_sample_batch(batch)
            if share == "bands":
                r.set_row_interleave(1, 3)
            else:
                r.set_pixel_range(5 * w + 3, 40 * w + 9)
            for _ in range(6):
                r.render()
            shots.append(r.linear_pixels().copy())
        assert np.array_equal(shots[0], shots[1]) and shots[0].any()


def test_ray_share_does_not_change_the_image(R):
    """The share of the wave slots a ray launch takes is a scheduling matter only: a small share makes
    fewer waves walk the same rays.  Fractions outside (0, 1] are refused."""
    from raytracer_amd import engine
    scene = workloads.cornell_box(light="panel")
    images = []
    for tiles, share in ((1, (1, 1)), (3, (1, 2)), (2, (1, 7)), (1, (1, 1000))):
        r = R.Renderer(scene, 96, 80)
        r.set_concurrent_tiles(tiles)
        r.set_ray_share(*share)
        for _ in range(2):
            r.render()
        images.append(r.linear_pixels().copy())
        assert r.statistics()["refused_references"] == 0
    for pixels in images[1:]:
        assert np.array_equal(pixels, images[0])
    r = R.Renderer(scene, 32, 32)
    for bad in ((0, 1), (1, 0), (3, 2)):
        with pytest.raises(engine.EngineError):
            r.set_ray_share(*bad)


def test_light_lists_none_many_and_finite_ranges(R, oracle):
    """The corners of light selection: a scene with no light at all (nothing to sample, no shadow ray;
    what remains is the constant background seen through the open front of the box), and one with five lights of all three kinds, finite ranges that cut some of them
    off before the walls, a spot cone whose inner and outer angles coincide, and a light of zero
    intensity (it is still selected with its share of the probability and adds nothing)."""
    dark = workloads.cornell_box()
    dark["lights"] = []
    r, host, gpu, cpu = render_both(R, oracle, dark, 64, 48, 2)
    assert np.isfinite(gpu).all() and (gpu[:, 3] == 1.0).all()
    assert close_fraction(gpu, cpu) > 0.99
    assert r.statistics()["shadow_rays"] == 0

    scene = workloads.cornell_box()
    inf = float("inf")
    scene["lights"] = [
        dict(transform=workloads.translation(-0.6, 1.7, 0.4), color=(1.0, 0.3, 0.2), intensity=2.0,
             range=1.2, innerCone=0.0, outerCone=0.7853982, type=2),
        dict(transform=workloads.look_at([0.5, 1.9, 0.5], [0.3, 0.0, -0.2]), color=(0.2, 0.6, 1.0), intensity=30.0,
             range=2.5, innerCone=0.3, outerCone=0.3, type=1),
        dict(transform=workloads.look_at([0.0, 1.9, 0.9], [0.0, 0.0, -0.9]), color=(1.0, 1.0, 0.8), intensity=0.6,
             range=inf, innerCone=0.0, outerCone=0.7853982, type=0),
        dict(transform=workloads.translation(0.0, 1.0, 0.0), color=(1.0, 1.0, 1.0), intensity=0.0,
             range=inf, innerCone=0.0, outerCone=0.7853982, type=2),
        dict(transform=workloads.translation(0.7, 0.3, 0.7), color=(0.4, 1.0, 0.4), intensity=1.0,
             range=0.05, innerCone=0.0, outerCone=0.7853982, type=2),
    ]
    r, host, gpu, cpu = render_both(R, oracle, scene, 96, 72, 4)
    assert np.isfinite(gpu).all() and cpu[:, :3].mean() > 0.005
    assert close_fraction(gpu, cpu, rel=2e-3) > 0.98
    assert rel_rmse(gpu[:, :3], cpu[:, :3]) < 5e-2


def test_every_material_map_matches_oracle(R, oracle):
    """Colour, normal, metallic-roughness, emissive, transmission and thickness maps together, through
    both texture coordinate sets, with offset / scale / rotation transforms and the clamp and mirror
    wrap modes (device.cu:86-98); the first-hit buffers (albedo, normal) are compared as well, since a
    wrong map lookup shows there without path noise."""
    scene = workloads.material_map_scene()
    w, h = 112, 80
    r, host, gpu, cpu = render_both(R, oracle, scene, w, h, 3)
    assert np.isfinite(gpu).all() and cpu[:, :3].mean() > 0.01
    assert close_fraction(gpu, cpu, rel=2e-3) > 0.97
    keep = cpu[:, :3].max(axis=1) < np.percentile(cpu[:, :3].max(axis=1), 99.5)
    assert rel_rmse(gpu[keep, :3], cpu[keep, :3]) < 5e-2
    for k in range(1, 7):
        dev = R.device_to_host(r.accumulation_buffer(k), w * h * 16).view(np.float32).reshape(-1, 4)
        assert close_fraction(dev, host.aov[k], rel=2e-3) > 0.97, k
    # every map takes part: without it the picture changes (the thickness map is looked up,
    # device.cu:164, but nothing downstream reads the product, so it is left out here)
    import copy
    for slot in ("baseColorTexture", "normalTexture", "metallicRoughnessTexture", "emissiveTexture",
                 "transmissionTexture"):
        plain = copy.deepcopy(scene)
        for m in plain["materials"]:
            m.pop(slot, None)
        q = R.Renderer(plain, w, h)
        for _ in range(3):
            q.render()
        changed = (np.abs(q.linear_pixels().reshape(-1, 4)[:, :3] - gpu[:, :3]) > 1e-3).any(axis=1).mean()
        assert changed > 0.01, (slot, changed)


def test_texture_wrap_with_sizes_that_are_no_power_of_two(R, oracle):
    """The same maps at edge lengths 29 and 45: repeat, mirror and clamp then go through the general residue
    (a power of two only needs a mask), with coordinates that run below zero; first-hit buffers and image
    against the oracle's device.cu:86-98."""
    scene = workloads.material_map_scene(size=29, checker=45)
    w, h = 112, 80
    r, host, gpu, cpu = render_both(R, oracle, scene, w, h, 2)
    assert np.isfinite(gpu).all() and cpu[:, :3].mean() > 0.01
    assert close_fraction(gpu, cpu, rel=2e-3) > 0.97
    for k in (5, 6):  # albedo and normal guides: what the maps say at the first hit, without path noise
        dev = R.device_to_host(r.accumulation_buffer(k), w * h * 16).view(np.float32).reshape(-1, 4)
        assert close_fraction(dev, host.aov[k], rel=2e-3) > 0.97, k


def test_scene_loaded_from_gltf_renders_the_same(R, tmp_path):
    """A scene that went through the glTF writer and rt::loadScene renders bit-identically to the scene
    built directly (PNG texture and Radiance environment included; the environment is compared after the
    same RGBE rounding)."""
    from gltf_writer import encode_hdr, float_to_rgbe, rgbe_to_float, write_scene
    scene = workloads.cornell_box(light="panel")
    path = write_scene(scene, tmp_path / "cornell.glb", "glb")
    a, b = R.Renderer(scene, 64, 64), R.Renderer(R.load_scene(path), 64, 64)
    for _ in range(2):
        a.render()
        b.render()
    assert np.array_equal(a.linear_pixels(), b.linear_pixels())

    scene = workloads.textured_sky_scene()
    rgbe = float_to_rgbe(np.asarray(scene["environment"]["pixels"], dtype=np.float32)[..., :3])
    scene["environment"]["pixels"] = rgbe_to_float(rgbe)
    (tmp_path / "sky.hdr").write_bytes(encode_hdr(rgbe))
    path = write_scene(scene, tmp_path / "sky.gltf", "gltf", environment_file="sky.hdr")
    a, b = R.Renderer(scene, 96, 54), R.Renderer(R.load_scene(path), 96, 54)
    a.render()
    b.render()
    assert np.array_equal(a.linear_pixels(), b.linear_pixels())
    assert a.linear_pixels()[..., :3].max() > 0.0

    # every material map, the second coordinate set, tangents, texture transforms and sampler wrap modes
    scene = workloads.material_map_scene()
    for kind in ("glb", "gltf"):
        path = write_scene(scene, tmp_path / f"maps.{kind}", kind)
        a, b = R.Renderer(scene, 80, 56), R.Renderer(R.load_scene(path), 80, 56)
        for _ in range(2):
            a.render()
            b.render()
        assert np.array_equal(a.linear_pixels(), b.linear_pixels()), kind


def test_command_line_renderer_writes_the_renderer_s_image(R, tmp_path):
    """raytracer scene.glb --output x.png / x.exr: the files hold what Renderer::pixels() /
    linearPixels() return for the same scene, size, profile and sample count (rows flipped top-down)."""
    import subprocess
    from gltf_writer import write_scene
    from test_scene_loader import png_reference, read_exr
    scene = workloads.cornell_box(light="panel")
    path = write_scene(scene, tmp_path / "cornell.glb", "glb")
    w, h, spp = 80, 48, 3
    for name in ("out.png", "out.exr"):
        p = subprocess.run([R.CLI_PATH, str(path), "--profile", "preview", "--width", str(w), "--height", str(h),
                            "--samples", str(spp), "--denoise", "off", "--output", str(tmp_path / name)],
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        assert "1 cameras, 0 lights" in p.stdout and f"Done at {spp} spp" in p.stdout and "Mrays/s" in p.stdout
    r = R.Renderer(R.load_scene(path), w, h, 0)
    for _ in range(spp):
        r.render()
    assert np.array_equal(png_reference((tmp_path / "out.png").read_bytes()), r.pixels().reshape(h, w, 4)[::-1])
    with np.errstate(over="ignore"):
        want = r.linear_pixels().reshape(h, w, 4)[::-1].astype(np.float16).astype(np.float32)
    assert np.array_equal(read_exr(tmp_path / "out.exr"), want)
    p = subprocess.run([R.CLI_PATH, str(path), "--output", str(tmp_path / "x.png"), "--denoise", "on"],
                       capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "denoiser is not part of this build" in p.stderr
