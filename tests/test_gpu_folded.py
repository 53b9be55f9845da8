"""Folded instances: rt_scene_build puts the triangles of a small mesh whose instance transform is the identity
into the tree over the instances themselves.  Every hit must be the hit of the instance, bit for bit: against
the oracle (which knows nothing of the folding), against the same scene built with RT_SCENE_FOLD=0, and through
the interval properties that need no second answer."""
import os

import numpy as np
import pytest

from raytracer_amd import workloads
from helpers import engine_scene, oracle_scene, compare_closest

pytestmark = pytest.mark.gpu


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def unfolded(engine, scene):
    os.environ["RT_SCENE_FOLD"] = "0"
    try:
        return engine_scene(engine, scene)
    finally:
        del os.environ["RT_SCENE_FOLD"]


def hostile(o, d):
    """Rays nobody should send: non-finite components, negative zeros, reversed, empty and NaN intervals."""
    nan, inf = np.float32("nan"), np.float32("inf")
    o, d = o.copy(), d.copy()
    o[0:50, 0] = nan
    d[50:100, 1] = nan
    o[100:150, 2] = inf
    d[150:200, 0] = -inf
    o[200:250, :3] *= np.float32(-0.0)          # origins of negative zeros
    d[250:300, 1] = np.float32(-0.0)            # a direction component of negative zero
    d[300:350, 0] = 0.0
    d[300:350, 2] = 0.0                          # parallel to the y axis
    o[350:550, 3] = -10.0                        # an interval that starts behind the origin
    o[550:600, 3] = nan
    d[600:650, 3] = nan
    o[650:700, 3] = 5.0
    d[650:700, 3] = 1.0                          # tmin beyond tmax
    d[700:750, :3] *= np.float32(1e-40)          # denormal direction
    return o, d


def test_which_instances_are_folded(engine):
    import torch
    scene = workloads.cornell_box(light="panel")
    top, blases = engine_scene(engine, scene)
    # three walls and the light panel stand where their vertices say; the two boxes are rotated
    identity = [np.array_equal(np.asarray(m, dtype=np.float32).reshape(-1), np.eye(4, dtype=np.float32).reshape(-1))
                for _, m in scene["instances"]]
    triangles = sum(len(scene["geometries"][g]["indices"]) for (g, _), same in zip(scene["instances"], identity) if same)
    assert top.folded() == (sum(identity), triangles) and sum(identity) == 4
    assert top.stats().instanceCount == len(scene["instances"])
    plain, _ = unfolded(engine, scene)
    assert plain.folded() == (0, 0)
    # a translation, a negative zero in the matrix, a mesh of more than 64 triangles: entered as instances
    moved = np.eye(4, dtype=np.float32)  # column-major, as the scene description has its matrices
    moved[3, 0] = 0.5
    signed = np.eye(4, dtype=np.float32)
    signed[0, 1] = -0.0
    quad = workloads.quad([-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1])
    big = workloads.triangle_soup(65, seed=3)["geometries"][0]
    geometries = [{"vertices": quad[0], "indices": quad[1], "material": 0}, big]
    meshes = []
    for g in geometries:
        meshes.append(engine.Blas(torch.from_numpy(np.ascontiguousarray(g["vertices"], dtype=np.float32)).cuda(),
                                  torch.from_numpy(np.ascontiguousarray(g["indices"].astype(np.int32))).cuda()))
    for transform, mesh, expected in ((moved, 0, (0, 0)), (signed, 0, (0, 0)), (np.eye(4, dtype=np.float32), 1, (0, 0)),
                                      (np.eye(4, dtype=np.float32), 0, (1, 2))):
        assert engine.Scene([(meshes[mesh], transform.reshape(-1))]).folded() == expected


@pytest.mark.parametrize("name", ["cornell", "dragon", "spheres"])
def test_folded_and_entered_instances_report_the_same_hits(engine, oracle, name):
    import torch
    scene = {"cornell": lambda: workloads.cornell_box(light="panel"), "dragon": workloads.dragon_cornell,
             "spheres": lambda: workloads.sphere_field(count=30, seed=4, subdivisions=2)}[name]()
    top, blases = engine_scene(engine, scene)
    plain, plain_blases = unfolded(engine, scene)
    assert top.folded()[0] >= 1
    o1, d1 = workloads.camera_rays(scene, 320, 180)
    o2, d2 = hostile(*workloads.random_rays(scene, 200000, seed=21))
    o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    hits, inst = top.trace_closest(dev(o), dev(d))
    same, sinst = plain.trace_closest(dev(o), dev(d))
    assert torch.equal(hits.view(torch.int32), same.view(torch.int32)) and torch.equal(inst, sinst)
    assert torch.equal(top.trace_any(dev(o), dev(d)), plain.trace_any(dev(o), dev(d)))
    if name != "dragon":  # the oracle walks every triangle of every instance here: small scenes only
        ref = oracle_scene(oracle, scene)
        tuv, prim, rinst = ref.trace_closest(o, d, use_bvh=False, threads=8)
        bad, report = compare_closest(hits, inst, tuv, prim, rinst)
        assert bad == 0, f"{bad} of {len(o)} rays differ: {report}"


@pytest.mark.parametrize("fold", [True, False])
def test_interval_that_ends_at_a_hit_on_a_flat_mesh(engine, fold):
    """A floor at y = 0 has no extent along y: its box must still hold the hit points the triangle test computes,
    whose rounding is a few ulps of the floor's size and not of the distance.  tmax = t keeps the hit."""
    import torch
    quad = workloads.quad([-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1])
    scene = dict(geometries=[{"vertices": quad[0], "indices": quad[1], "material": 0}],
                 instances=[(0, np.eye(4, dtype=np.float32).reshape(-1))])
    top, blases = engine_scene(engine, scene) if fold else unfolded(engine, scene)
    assert top.folded() == ((1, 2) if fold else (0, 0))
    rng = np.random.default_rng(5)
    n = 400000
    o = np.zeros((n, 4), dtype=np.float32)
    d = np.zeros((n, 4), dtype=np.float32)
    o[:, 0] = rng.uniform(-0.9, 0.9, n)
    o[:, 2] = rng.uniform(-0.9, 0.9, n)
    o[:, 1] = rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-4.0, -0.5, n)  # close to the floor, either side
    o[:, 3] = 1e-5
    direction = rng.normal(size=(n, 3))
    direction[:, 1] = -np.sign(o[:, 1]) * np.abs(direction[:, 1])  # towards the floor
    d[:, :3] = direction / np.linalg.norm(direction, axis=1, keepdims=True)
    d[:, 3] = 1e16
    od, dd = dev(o), dev(d)
    hits, inst = top.trace_closest(od, dd)
    found = hits[:, 3].view(torch.int32) != -1
    assert float(found.float().mean()) > 0.3
    closed = dd.clone()
    closed[found, 3] = hits[:, 0][found]
    same, _ = top.trace_closest(od, closed)
    assert torch.equal(same[found].view(torch.int32), hits[found].view(torch.int32))


def test_origins_on_the_geometry(engine, oracle):
    """Rays that start on the plane of a flat mesh or on a face of an instance's box, tmin = 0, against flat meshes
    that are folded (identity) and entered (shifted along their own plane): the oracle's brute force decides."""
    rng = np.random.default_rng(11)
    floor = workloads.quad([-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1])
    wall = workloads.quad([0, -1, -1], [0, 1, -1], [0, 1, 1], [0, -1, 1])
    box = workloads.box_mesh([-0.5, -0.25, -0.75], [0.25, 0.5, 0.125])
    geometries = [{"vertices": m[0], "indices": m[1], "material": 0} for m in (floor, wall, box)]
    shift = np.eye(4, dtype=np.float32)
    shift[3, :3] = (0.375, 0.0, -0.25)
    scene = dict(geometries=geometries, instances=[(0, np.eye(4, dtype=np.float32).reshape(-1)), (1, shift.reshape(-1)),
                                                   (2, np.eye(4, dtype=np.float32).reshape(-1)), (0, shift.reshape(-1))])
    top, blases = engine_scene(engine, scene)
    assert top.folded() == (2, 14)
    ref = oracle_scene(oracle, scene)
    n = 200000
    o, d = workloads.random_rays(scene, n, seed=3)
    planes = np.array([0.0, 0.375, -0.5, 0.25, -0.25, 0.5, -0.75, 0.125, 1.0, -1.0], dtype=np.float32)
    axis = rng.integers(0, 3, n)
    o[np.arange(n), axis] = planes[rng.integers(0, len(planes), n)]
    o[:, 3] = 0.0
    hits, inst = top.trace_closest(dev(o), dev(d))
    tuv, prim, rinst = ref.trace_closest(o, d, use_bvh=False, threads=8)
    bad, report = compare_closest(hits, inst, tuv, prim, rinst)
    assert bad == 0, f"{bad} of {n} rays differ: {report}"
    plain, _ = unfolded(engine, scene)
    same, sinst = plain.trace_closest(dev(o), dev(d))
    bad, report = compare_closest(same, sinst, tuv, prim, rinst)
    assert bad == 0, f"not folded: {bad} of {n} rays differ: {report}"
