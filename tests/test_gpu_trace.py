"""GPU parity tests for the BVH build and the closest-hit / any-hit kernels:
the HIP path, through the C ABI, against the CPU oracle.  t, barycentrics,
primitive and instance ids are compared bit for bit."""
import numpy as np
import pytest

from raytracer_amd import workloads
from helpers import compare_closest, engine_scene, oracle_scene

pytestmark = pytest.mark.gpu


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def check_scene(engine, oracle, scene, rays, any_hit=True, brute=False):
    top, blases = engine_scene(engine, scene)
    ref = oracle_scene(oracle, scene)
    for i in range(len(scene["instances"])):
        assert np.array_equal(top.instance_inverse(i), ref.inverse(i))
    o, d = rays
    hits, inst = top.trace_closest(dev(o), dev(d))
    tuv, prim, rinst = ref.trace_closest(o, d, use_bvh=not brute, threads=8)
    bad, report = compare_closest(hits, inst, tuv, prim, rinst)
    assert bad == 0, f"{bad} of {len(o)} rays differ: {report}"
    if any_hit:
        occ = top.trace_any(dev(o), dev(d)).cpu().numpy()
        assert np.array_equal(occ, (prim != 0xffffffff).astype(np.uint8))
    return top, (tuv, prim, rinst)


def test_cornell_camera_and_random_rays(engine, oracle):
    scene = workloads.cornell_box()
    o1, d1 = workloads.camera_rays(scene, 256, 256)
    o2, d2 = workloads.random_rays(scene, 100000, seed=1)
    top, (tuv, prim, inst) = check_scene(engine, oracle, scene, (np.concatenate([o1, o2]), np.concatenate([d1, d2])),
                                         brute=True)
    assert (prim[:65536] != 0xffffffff).all()
    s = top.stats()
    assert s.triangleCount == 30 and s.instanceCount == 5


def test_single_triangle_and_tiny_meshes(engine, oracle):
    for n in (1, 2, 3, 4, 5, 9, 33):
        scene = workloads.triangle_soup(n, seed=n)
        o, d = workloads.random_rays(scene, 20000, seed=n)
        check_scene(engine, oracle, scene, (o, d), brute=True)


def test_axis_aligned_and_degenerate_rays(engine, oracle):
    scene = workloads.cornell_box()
    rng = np.random.default_rng(3)
    n = 30000
    o = rng.uniform(-0.9, 0.9, (n, 3)) * [1, 1, 1] + [0, 1, 0]
    o[:, 1] = rng.uniform(0.05, 1.95, n)
    axes = np.eye(3)[rng.integers(0, 3, n)] * rng.choice([-1.0, 1.0], n)[:, None]
    # rays along the axes, some starting exactly on box-edge coordinates
    o[: n // 3, 0] = rng.choice([-0.3, 0.3, 0.35, -0.35, 0.0], n // 3)
    org = np.concatenate([o, np.full((n, 1), 0.001)], 1).astype(np.float32)
    dr = np.concatenate([axes, np.full((n, 1), 1e16)], 1).astype(np.float32)
    # a zero direction and a huge tmin: both must miss everywhere
    org[-1] = [0, 1, 0, 0.001]
    dr[-1] = [0, 0, 0, 1e16]
    org[-2, 3] = 1e10
    check_scene(engine, oracle, scene, (org, dr), brute=True)


def test_hostile_rays(engine, oracle):
    """Rays a caller should not send but may: NaN and infinite components, an empty or reversed interval,
    a negative tmin, denormal directions.  Nothing may fault, no reference may be refused, and whatever
    the walk reports must be what the scalar restatement reports."""
    scene = workloads.sphere_field(count=30, seed=4, subdivisions=2)
    o, d = workloads.random_rays(scene, 40000, seed=8)
    nan, inf = np.float32("nan"), np.float32("inf")
    k = 0
    for values in ((nan, 0, 0), (0, nan, 0), (inf, 0, 0), (-inf, 1, 1)):
        for target in (o, d):
            target[k * 50:(k + 1) * 50, :3] = np.where(np.isfinite(values) & (np.array(values) == 0), target[k * 50:(k + 1) * 50, :3],
                                                       np.array(values, dtype=np.float32))
            k += 1
    base = k * 50
    o[base:base + 200, 3] = 5.0          # tmin beyond tmax
    d[base:base + 200, 3] = 1.0
    o[base + 200:base + 400, 3] = -10.0  # interval that starts behind the origin
    d[base + 400:base + 600, 3] = 0.0    # empty interval
    d[base + 600:base + 800, 3] = nan
    o[base + 800:base + 1000, 3] = nan
    d[base + 1000:base + 1200, :3] *= np.float32(1e-40)  # denormal direction
    d[base + 1200:base + 1400, :3] *= np.float32(1e30)   # very long direction
    top, _ = check_scene(engine, oracle, scene, (o, d))
    assert top is not None


def test_short_rays_tmax(engine, oracle):
    scene = workloads.cornell_box(light="panel")
    o, d = workloads.random_rays(scene, 100000, seed=9)
    d[:, 3] = np.random.default_rng(1).uniform(0.0, 2.0, len(d)).astype(np.float32)
    check_scene(engine, oracle, scene, (o, d), brute=True)


def test_soup_incoherent(engine, oracle):
    scene = workloads.triangle_soup(100000)
    o, d = workloads.random_rays(scene, 300000, seed=2)
    top, _ = check_scene(engine, oracle, scene, (o, d))
    s = top.stats()
    assert s.triangleCount == 100000 and s.nodeCount > 1000


def test_clustering_builder_gives_the_same_hits(engine, oracle, monkeypatch):
    """RT_BLAS_BUILDER=ploc selects the locally-ordered clustering build; hits do not depend on the tree."""
    monkeypatch.setenv("RT_BLAS_BUILDER", "ploc")
    scene = workloads.triangle_soup(50000, seed=9)
    check_scene(engine, oracle, scene, workloads.random_rays(scene, 100000, seed=3))
    scene = workloads.sphere_field(count=50, subdivisions=2)
    check_scene(engine, oracle, scene, workloads.random_rays(scene, 50000, seed=6))
    for n in (1, 2, 3, 5, 9):
        scene = workloads.triangle_soup(n, seed=n)
        check_scene(engine, oracle, scene, workloads.random_rays(scene, 2000, seed=n), brute=True)


def test_instanced_sphere_field(engine, oracle):
    scene = workloads.sphere_field(count=300, subdivisions=3)
    o1, d1 = workloads.camera_rays(scene, 320, 180)
    o2, d2 = workloads.random_rays(scene, 100000, seed=4)
    top, (tuv, prim, inst) = check_scene(engine, oracle, scene,
                                         (np.concatenate([o1, o2]), np.concatenate([d1, d2])))
    assert len(np.unique(inst)) > 100  # many different instances are hit


def test_dragon_standin_primary_and_bounce_rays(engine, oracle):
    scene = workloads.dragon_cornell()
    o1, d1 = workloads.camera_rays(scene, 480, 270)
    ref = oracle_scene(oracle, scene)
    tuv, prim, inst = ref.trace_closest(o1, d1, threads=8)
    # secondary rays leaving the primary hits in random directions
    rng = np.random.default_rng(6)
    hit = prim != 0xffffffff
    p = o1[hit, :3] + d1[hit, :3] * tuv[hit, :1]
    nd = rng.normal(0, 1, p.shape)
    nd /= np.linalg.norm(nd, axis=1, keepdims=True)
    o2 = np.concatenate([p - d1[hit, :3] * 1e-3, np.full((len(p), 1), 0.001)], 1).astype(np.float32)
    d2 = np.concatenate([nd, np.full((len(p), 1), 1e16)], 1).astype(np.float32)
    check_scene(engine, oracle, scene, (np.concatenate([o1, o2]), np.concatenate([d1, d2])))


def test_full_frame_properties_without_the_oracle(engine):
    """The benchmark's size (one ray per pixel of a 1080p frame plus as many incoherent rays, 871 k
    triangles), checked through properties that need no CPU answer: the same call twice gives the same
    bits; a ray's result does not depend on where it sits in the batch (rays are handed to persistent
    waves in whatever order they ask); any-hit agrees with closest-hit; the hit interval is (tmin, tmax]."""
    import torch
    scene = workloads.dragon_cornell()
    top, blases = engine_scene(engine, scene)
    o1, d1 = workloads.camera_rays(scene, 1920, 1080)
    o2, d2 = workloads.random_rays(scene, len(o1), seed=8)
    o, d = dev(np.concatenate([o1, o2])), dev(np.concatenate([d1, d2]))
    hits, inst = top.trace_closest(o, d)
    again, inst2 = top.trace_closest(o, d)
    assert torch.equal(hits.view(torch.int32), again.view(torch.int32)) and torch.equal(inst, inst2)
    order = torch.randperm(o.shape[0], device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    shuffled, sinst = top.trace_closest(o[order].contiguous(), d[order].contiguous())
    assert torch.equal(shuffled.view(torch.int32), hits[order].view(torch.int32)) and torch.equal(sinst, inst[order])
    found = hits[:, 3].view(torch.int32) != -1
    assert 0.5 < float(found.float().mean()) <= 1.0
    assert torch.equal(top.trace_any(o, d).bool(), found)
    # tmax = t keeps the hit (closed end); the next float below t loses it or finds nothing nearer or equal
    t = hits[:, 0]
    closed = d.clone()
    closed[found, 3] = t[found]
    same, _ = top.trace_closest(o, closed)
    assert torch.equal(same[found].view(torch.int32), hits[found].view(torch.int32))
    opened = d.clone()
    opened[found, 3] = torch.nextafter(t[found], torch.zeros_like(t[found]))
    before, _ = top.trace_closest(o, opened)
    missed = before[found][:, 3].view(torch.int32) == -1
    assert bool((missed | (before[found][:, 0] < t[found])).all())
    s = top.stats()
    assert s.triangleCount == 871212 and s.instanceCount == len(scene["instances"])


def test_ragged_batch_sizes(engine, oracle):
    """Batches around the sizes the ray feed is cut at: one ray, one wave's worth and one more, chunks
    that are partly or wholly empty, fewer rays than the launch has waves (most of which then leave at
    once) and a batch every wave draws from."""
    scene = workloads.sphere_field(count=40, seed=9, subdivisions=2)
    top, blases = engine_scene(engine, scene)
    ref = oracle_scene(oracle, scene)
    o, d = workloads.random_rays(scene, 400003, seed=12)
    tuv, prim, rinst = ref.trace_closest(o, d, use_bvh=True, threads=8)
    for n in (1, 2, 63, 64, 65, 129, 511, 513, 1000, 4097, 70001, 400003):
        hits, inst = top.trace_closest(dev(o[:n]), dev(d[:n]))
        bad, report = compare_closest(hits, inst, tuv[:n], prim[:n], rinst[:n])
        assert bad == 0, f"{n} rays: {bad} differ: {report}"
        occ = top.trace_any(dev(o[:n]), dev(d[:n])).cpu().numpy()
        assert np.array_equal(occ, (prim[:n] != 0xffffffff).astype(np.uint8)), n


def soup_scene(points):
    """A one-geometry scene from (n, 3, 3) triangle corners."""
    p = np.asarray(points, dtype=np.float32).reshape(-1, 3)
    idx = np.arange(len(p), dtype=np.uint32).reshape(-1, 3)
    return dict(geometries=[{"vertices": workloads.make_vertices(p), "indices": idx, "material": 0}],
                instances=[(0, np.eye(4, dtype=np.float32).reshape(-1))], materials=[workloads.default_material()],
                lights=[], cameras=[], environment=None)


def test_degenerate_geometry(engine, oracle):
    """Geometry that stresses the builder rather than the rays: many copies of one triangle (equal Morton
    codes, equal distances: the tie policy decides), zero-area and needle triangles, a mesh far from the
    origin, and triangles of very different sizes in one geometry."""
    rng = np.random.default_rng(21)
    base = np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
    copies = np.repeat(base[None], 3000, axis=0)
    needles = rng.uniform(-1, 1, (500, 1, 3)) + np.array([[0, 0, 0], [1e-6, 0, 0], [2.0, 2e-6, 0]])[None]
    points = rng.uniform(-1, 1, (300, 1, 3)) + np.zeros((1, 3, 3))  # three equal corners
    lines = rng.uniform(-1, 1, (300, 1, 3)) + np.array([[0, 0, 0], [0.5, 0.5, 0.5], [1.0, 1.0, 1.0]])[None]
    tiny = rng.uniform(-1, 1, (2000, 1, 3)) + rng.normal(0, 1e-4, (2000, 3, 3))
    huge = rng.uniform(-50, 50, (20, 3, 3))
    far = workloads.triangle_soup(2000, seed=5)["geometries"][0]["vertices"][:, :3].reshape(-1, 3, 3) + 1.0e5
    for name, tris in (("copies", copies), ("needles + points + lines", np.concatenate([needles, points, lines, tiny[:200]])),
                       ("mixed sizes", np.concatenate([tiny, huge])), ("far from the origin", far)):
        scene = soup_scene(tris)
        o, d = workloads.random_rays(scene, 60000, seed=3)
        # some rays straight at the shared triangle / along coordinate planes
        o[:1000, :3] = [0.25, 0.25, 1.0] + (tris.reshape(-1, 3).mean(0) - [1 / 3, 1 / 3, 0]) * (name == "far from the origin")
        d[:1000, :3] = [0.0, 0.0, -1.0]
        check_scene(engine, oracle, scene, (o, d), brute=len(tris) <= 4000)


def test_counters_and_empty_launch(engine, oracle):
    import torch
    scene = workloads.cornell_box()
    top, blases = engine_scene(engine, scene)
    o, d = workloads.camera_rays(scene, 64, 64)
    counters = torch.zeros(4, dtype=torch.int64, device="cuda")
    top.trace_closest(dev(o), dev(d), counters=counters)
    c = counters.cpu().numpy()
    assert c[0] == 4096 and c[1] >= 4096 and c[2] > 0 and c[3] > 0
    hits, inst = top.trace_closest(torch.empty((0, 4), device="cuda"), torch.empty((0, 4), device="cuda"))
    assert hits.shape[0] == 0


def test_empty_geometry_is_rejected(engine):
    import torch
    with pytest.raises(engine.EngineError):
        engine.Blas(torch.zeros((3, 16), device="cuda"), torch.zeros((0, 3), dtype=torch.int32, device="cuda"))
    with pytest.raises(engine.EngineError):
        engine.Scene([])


def test_index_beyond_the_vertices_is_rejected(engine):
    """The build reads vertices through the indices; an index that names no vertex is an error message,
    not a read outside the vertex buffer."""
    import torch
    scene = workloads.triangle_soup(5000, seed=2)
    v = dev(scene["geometries"][0]["vertices"])
    idx = scene["geometries"][0]["indices"].astype(np.int32).copy()
    engine.Blas(v, dev(idx))
    for bad in (len(scene["geometries"][0]["vertices"]), 0x7fffffff, -1):
        broken = idx.copy()
        broken[4321, 1] = bad
        with pytest.raises(engine.EngineError, match="index"):
            engine.Blas(v, dev(broken))
    torch.cuda.synchronize()


def test_chain_of_nested_instances_fits_the_traversal_stack(engine, oracle):
    """Instances whose sizes and distances grow geometrically make the top-level sweep peel a few instances
    off per cut: left to the surface-area cuts alone that is a chain of some forty wide nodes, deeper than the
    traversal stack can follow.  The builder cuts by count where the levels left would not hold a group
    otherwise, so the scene builds within the stack's depth, traces bit for bit like the oracle, and renders;
    the reference (optixAccelBuild, renderer.cpp:414-432) takes such a scene too."""
    from raytracer_amd import renderer as R
    R.load()
    box = workloads.box_mesh([-0.5, -0.5, -0.5], [0.5, 0.5, 0.5])

    def chain(count):
        scene = workloads.cornell_box()
        scene["geometries"] = [{"vertices": box[0], "indices": box[1], "material": 1}]
        scene["instances"] = [(0, workloads.translation(3.0 * 1.3 ** k, 0.0, 0.0, 1.3 ** k))
                              for k in range(-(count // 2), count - count // 2)]
        return scene

    for count in (40, 300, 3000):
        scene = chain(min(count, 300))
        if count == 3000:  # 1.3^1500 is no float: a slower progression
            scene["instances"] = [(0, workloads.translation(3.0 * 1.01 ** k, 0.0, 0.0, 1.01 ** k)) for k in range(-1500, 1500)]
        rays = workloads.random_rays(scene, 20000, seed=count)
        top, _ = check_scene(engine, oracle, scene, rays, brute=count <= 300)
        assert top.stats().maxDepth <= 18, top.stats().maxDepth
    deep = chain(300)
    r = R.Renderer(deep, 64, 64)
    r.render()
    r.synchronize()
    stats = r.statistics()
    assert stats["refused_references"] == 0 and stats["extend_rays"] >= 64 * 64
    assert np.isfinite(r.linear_pixels()).all()


def test_many_launches_in_flight_on_two_streams(engine):
    """More batched queries in flight than the scene keeps cursor sets for (16), alternating between two
    streams with nothing waited for in between: every launch still draws its rays through a set of its
    own, so every batch equals what the same batch gives alone.  Long batches next to short ones make
    the launches overlap in time."""
    import torch
    scene = workloads.dragon_cornell()
    top, blases = engine_scene(engine, scene)
    o, d = workloads.random_rays(scene, 600000, seed=14)
    o, d = dev(o), dev(d)
    cuts = [0]
    rng = np.random.default_rng(2)
    while cuts[-1] < o.shape[0]:
        cuts.append(min(o.shape[0], cuts[-1] + int(rng.choice([300, 5000, 25000]))))
    batches = [(o[a:b].contiguous(), d[a:b].contiguous()) for a, b in zip(cuts, cuts[1:])]
    assert len(batches) > 40
    want = [top.trace_closest(bo, bd) for bo, bd in batches]
    want_any = [top.trace_any(bo, bd) for bo, bd in batches]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got, got_any = [], []
    for k, (bo, bd) in enumerate(batches):
        hits = torch.empty((bo.shape[0], 4), dtype=torch.float32, device="cuda")
        inst = torch.empty(bo.shape[0], dtype=torch.int32, device="cuda")
        occ = torch.empty(bo.shape[0], dtype=torch.uint8, device="cuda")
        top.trace_closest(bo, bd, stream=streams[k % 2], hits=hits, inst=inst)
        top.trace_any(bo, bd, stream=streams[(k + 1) % 2], out=occ)
        got.append((hits, inst))
        got_any.append(occ)
    torch.cuda.synchronize()
    for k in range(len(batches)):
        assert torch.equal(got[k][0].view(torch.int32), want[k][0].view(torch.int32)), k
        assert torch.equal(got[k][1], want[k][1]), k
        assert torch.equal(got_any[k], want_any[k]), k


def test_surface_area_refinement_keeps_hits_and_saves_node_visits(engine, oracle, monkeypatch):
    """The upper levels of a bottom-level tree are rebuilt with binned surface-area splits over the radix
    tree's clusters.  Hits do not depend on the tree: the plain radix tree (RT_BLAS_REFINE=0), the default
    and a rebuild down to clusters of four triangles all match the oracle bit for bit; on the dense mesh
    the refined trees visit no more nodes per ray than the plain one."""
    import torch
    scene = workloads.dragon_cornell(triangles=120000)
    rays = workloads.random_rays(scene, 150000, seed=17)
    visits = {}
    for name, env in (("plain", {"RT_BLAS_REFINE": "0"}), ("default", {}), ("deep", {"RT_BLAS_CLUSTERS": "1000000"})):
        monkeypatch.delenv("RT_BLAS_REFINE", raising=False)
        monkeypatch.delenv("RT_BLAS_CLUSTERS", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        top, _ = check_scene(engine, oracle, scene, rays)
        counters = torch.zeros(4, dtype=torch.int64, device="cuda")
        top.trace_closest(dev(rays[0]), dev(rays[1]), counters=counters)
        c = counters.cpu().numpy()
        visits[name] = c[1] / c[0]
    assert visits["default"] <= visits["plain"] * 1.005 and visits["deep"] <= visits["plain"] * 1.005, visits
    for n in (65, 200, 5000):  # sizes around the point where the rebuild starts
        soup = workloads.triangle_soup(n, seed=n)
        monkeypatch.setenv("RT_BLAS_CLUSTERS", "64")
        check_scene(engine, oracle, soup, workloads.random_rays(soup, 20000, seed=n), brute=n <= 200)


def test_axis_parallel_rays_stay_cheap(engine, oracle):
    """Rays with one or two direction components of exactly zero (probe rays along the axes, rays inside an
    axis plane - also what a normal generator yields now and then, a few in 2^24) on a 200 k soup: the
    same bits as the oracle, origins on the coordinates of triangle corners included, and about as many
    node visits as their oblique neighbours, where a slab test that leaves the parallel axis open visits
    a hundred times as many."""
    import torch
    scene = workloads.triangle_soup(200000, seed=11)
    rng = np.random.default_rng(12)
    n = 60000
    o, d = workloads.random_rays(scene, n, seed=13)
    oblique = (o.copy(), d.copy())
    kind = rng.integers(0, 3, n)
    axis = rng.integers(0, 3, n)
    for k in range(3):
        one = (kind == 0) & (axis == k)           # one component zero: the ray lies in a plane
        d[one, k] = 0.0
        two = (kind == 1) & (axis == k)           # two components zero: along axis k
        d[two, (k + 1) % 3] = 0.0
        d[two, (k + 2) % 3] = 0.0
    d[kind == 2, 1] = -0.0                        # minus zero counts as zero
    d[:, :3] /= np.linalg.norm(d[:, :3], axis=1, keepdims=True)
    d = d.astype(np.float32)
    # a third of the origins sit exactly on corner coordinates of the mesh along the parallel axis
    corners = scene["geometries"][0]["vertices"][:, :3]
    picks = rng.integers(0, len(corners), n)
    snap = rng.random(n) < 0.33
    nudge = rng.integers(-1, 2, n)                 # ... or one float beside them, on either side
    for k in range(3):
        rows = snap & (d[:, k] == 0.0)
        on = corners[picks[rows], k]
        o[rows, k] = np.where(nudge[rows] < 0, np.nextafter(on, np.float32(-np.inf)),
                              np.where(nudge[rows] > 0, np.nextafter(on, np.float32(np.inf)), on))
    assert (d[:, :3] == 0).any(axis=1).mean() > 0.9
    top, _ = check_scene(engine, oracle, scene, (o, d))

    def visits(rays):
        c = torch.zeros(4, dtype=torch.int64, device="cuda")
        top.trace_closest(dev(rays[0]), dev(rays[1]), counters=c)
        c = c.cpu().numpy()
        return c[1] / c[0]
    parallel, ordinary = visits((o, d)), visits(oblique)
    print(f"nodes per ray: axis-parallel {parallel:.1f}, oblique {ordinary:.1f}")
    assert parallel < 1.5 * ordinary


def test_small_mesh_seen_from_far_away(engine, oracle):
    """A mesh whose nodes are many orders of magnitude smaller than their distance from the ray origins,
    both as it is and behind an instance that scales it down further: the plane distances of a node's
    children then differ by less than their rounding, the empty slots of a node (an inverted box that the
    slab test is meant to reject) can pass the test, and the walk has to drop their references instead of
    following them.  Hits as the oracle's, bit for bit, from every distance."""
    soup = workloads.triangle_soup(3000, seed=21)
    g = soup["geometries"][0]
    g["vertices"] = g["vertices"].copy()
    g["vertices"][:, :3] *= np.float32(1e-3)            # the mesh fills a cube of one millimetre
    shrink = np.eye(4, dtype=np.float32)
    shrink[0, 0] = shrink[1, 1] = shrink[2, 2] = 1e-3
    shrink[:3, 3] = [5e-3, 0.0, 0.0]
    scene = dict(soup)
    scene["instances"] = [(0, np.eye(4, dtype=np.float32).reshape(-1)), (0, shrink.T.reshape(-1))]
    rng = np.random.default_rng(22)
    n = 40000
    targets = rng.random((n, 3)) * 1e-3
    targets[n // 2:] = targets[n // 2:] * 1e-3 + [5e-3, 0.0, 0.0]
    away = rng.normal(size=(n, 3))
    away /= np.linalg.norm(away, axis=1, keepdims=True)
    distance = 10.0 ** rng.uniform(-1.0, 7.0, (n, 1))   # ten centimetres to ten thousand kilometres
    o = targets + away * distance
    d = -away
    org = np.concatenate([o, np.full((n, 1), 0.001)], axis=1).astype(np.float32)
    dr = np.concatenate([d, np.full((n, 1), 1e16)], axis=1).astype(np.float32)
    _, (tuv, prim, _) = check_scene(engine, oracle, scene, (org, dr))
    assert (prim != 0xffffffff).sum() > 1000             # near origins resolve the mesh, far ones mostly do not


def test_geometry_records_must_cover_every_instance(engine):
    """The shading side looks a hit's vertices up through its instance's geometry record: a table of records
    shorter than the geometry numbers the instances carry is refused, not read past its end."""
    import ctypes as C
    scene = workloads.cornell_box()
    top, blases = engine_scene(engine, scene)
    geometries = max(g for g, _ in scene["instances"]) + 1
    assert geometries >= 2
    records = (C.c_uint8 * (24 * geometries))()
    lib = engine.load()
    engine.check(lib.rt_scene_set_geometry(top.handle, records, geometries))
    with pytest.raises(engine.EngineError, match="geometry beyond the records"):
        engine.check(lib.rt_scene_set_geometry(top.handle, records, geometries - 1))
    # the refused call changed nothing: the scene still has the records of the call before it
    import ctypes
    params = (ctypes.c_uint8 * 344)()
    ctypes.cast(params, ctypes.POINTER(ctypes.c_uint64))[168 // 8] = lib.rt_scene_handle(top.handle)
    state = ctypes.c_void_p(0)
    engine.check(lib.rt_wavefront_create(64, ctypes.byref(state)))
    try:
        with pytest.raises(engine.EngineError) as failure:   # the launch gets past the record check to the next one
            engine.check(lib.rt_render_sample(state, params, 0, 1, None))
        assert "rt_scene_set_geometry has not been called" not in str(failure.value)
        assert "materials" in str(failure.value)
    finally:
        lib.rt_wavefront_destroy(state)


def test_rays_ordered_by_origin_report_the_same_hits(engine, oracle):
    """A large batch over a large scene is traced through an index sorted by origin cell (RT_TRACE_ORDER=1 does
    it for any batch): every ray's result stays at the ray's own place and is what it was, NaN origins and ragged
    sizes included, for both kinds of query and from two streams at once."""
    import os
    import torch
    scene = workloads.triangle_soup(100000)
    top, blases = engine_scene(engine, scene)
    ref = oracle_scene(oracle, scene)
    o, d = workloads.random_rays(scene, 300001, seed=31)
    o[5:40, 0] = np.float32("nan")
    o[40:80, :3] += np.float32(1e6)   # far outside the scene's box
    od, dd = dev(o), dev(d)
    plain, pinst = top.trace_closest(od, dd)
    shadow = top.trace_any(od, dd)
    os.environ["RT_TRACE_ORDER"] = "1"
    try:
        for n in (300001, 65, 64, 2):
            hits, inst = top.trace_closest(od[:n].contiguous(), dd[:n].contiguous())
            assert torch.equal(hits.view(torch.int32), plain[:n].view(torch.int32)) and torch.equal(inst, pinst[:n])
            assert torch.equal(top.trace_any(od[:n].contiguous(), dd[:n].contiguous()), shadow[:n])
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        results = []
        for k in range(6):
            with torch.cuda.stream(streams[k & 1]):
                results.append(top.trace_closest(od, dd, stream=streams[k & 1]))
        torch.cuda.synchronize()
        for hits, inst in results:
            assert torch.equal(hits.view(torch.int32), plain.view(torch.int32)) and torch.equal(inst, pinst)
    finally:
        del os.environ["RT_TRACE_ORDER"]
    tuv, prim, rinst = ref.trace_closest(o, d, use_bvh=True, threads=8)
    bad, report = compare_closest(plain, pinst, tuv, prim, rinst)
    assert bad == 0, report
