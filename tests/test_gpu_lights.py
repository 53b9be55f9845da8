"""GPU tests of the light index (rt_light_index_build): the thresholds select the light the reference's
walk over the weights selects (device.cu:433-447) for every target, the table finds the light the walk
over the list finds for a triangle (device.cu:607-621), and a render through the index equals the render
through the walks bit for bit and the oracle's within the integrator tolerance."""
import os

import numpy as np
import pytest

from raytracer_amd import workloads
from helpers import oracle_scene, rel_rmse
from ref_setup import Setup

pytestmark = pytest.mark.gpu

LIGHT = np.dtype([("a", "<f4", 3), ("b", "<f4", 3), ("c", "<f4", 3), ("normal", "<f4", 3), ("emission", "<f4", 3),
                  ("area", "<f4"), ("weight", "<f4"), ("range", "<f4"), ("innerCone", "<f4"), ("outerCone", "<f4"),
                  ("instance", "<u4"), ("primitive", "<u4"), ("type", "<u4")])
assert LIGHT.itemsize == 92


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def light_list(weights, rng, emissive_share=0.9, instances=7, primitives=5000):
    lights = np.zeros(len(weights), dtype=LIGHT)
    lights["weight"] = weights
    lights["type"] = np.where(rng.random(len(weights)) < emissive_share, 3, 2)
    lights["instance"] = rng.integers(0, instances, len(weights))
    lights["primitive"] = rng.integers(0, primitives, len(weights))   # pairs repeat: several lights per triangle
    lights["area"] = 1.0
    return lights


def sequential_sum(weights):
    total = np.float32(0.0)
    for w in weights.astype(np.float32):
        total = np.float32(total + w)
    return float(total)


WEIGHTS = {
    "even": lambda rng, n: rng.uniform(0.5, 1.5, n),
    "wide": lambda rng, n: np.exp(rng.normal(0.0, 6.0, n)),                       # 1e-8 .. 1e8
    "zeros and giants": lambda rng, n: np.where(rng.random(n) < 0.3, 0.0, np.exp(rng.normal(0.0, 3.0, n)))
    * np.where(rng.random(n) < 0.01, 1e6, 1.0),
    "tiny behind a giant": lambda rng, n: np.concatenate([[1e7], rng.uniform(1e-3, 1.0, n - 1)]),
}


@pytest.mark.parametrize("kind", list(WEIGHTS))
@pytest.mark.parametrize("count", [9, 1000, 20000])
def test_thresholds_select_the_walks_light(engine, oracle, kind, count):
    """Against the oracle's walk: dense random numbers with the list's own total weight (float sum in
    list order, as the host builds it), and, with a total of one so that a target can be named exactly,
    every threshold itself, its two neighbouring floats, zero and values beyond the last threshold."""
    import torch
    rng = np.random.default_rng(count + len(kind))
    weights = WEIGHTS[kind](rng, count).astype(np.float32)
    lights = light_list(weights, rng)
    raw = dev(lights.view(np.uint8).reshape(-1))
    index = engine.LightIndex(raw)
    thresholds = index.thresholds().cpu().numpy()
    assert (np.diff(thresholds) >= 0).all() and thresholds[0] >= 0
    total = sequential_sum(weights)
    uniform = rng.random(1 << 17, dtype=np.float32)
    uniform[:4] = [0.0, np.float32(1.0) - np.float32(2.0 ** -24), 0.5, np.float32(2.0 ** -24)]
    chosen, _ = engine.light_query(raw, total, index, uniform=dev(uniform))
    walked, _ = engine.light_query(raw, total, None, uniform=dev(uniform))
    expect, _ = oracle.light_walks(lights.view(np.uint8).reshape(-1), total, uniform=uniform, threads=16)
    assert np.array_equal(chosen.cpu().numpy().view(np.uint32), expect)
    assert np.array_equal(walked.cpu().numpy().view(np.uint32), expect)
    if kind == "even":
        assert len(np.unique(expect)) > min(count, 1000) // 2
    # exact targets around every threshold (lightWeight 1: target = the number handed in)
    picks = np.arange(count) if count <= 1000 else rng.choice(count, 1000, replace=False)
    around = np.concatenate([thresholds[picks], np.nextafter(thresholds[picks], np.float32(np.inf)),
                             np.nextafter(thresholds[picks], np.float32(-np.inf)),
                             [0.0, thresholds[-1] * 2 + 1, np.float32(3e38)]]).astype(np.float32)
    chosen, _ = engine.light_query(raw, 1.0, index, uniform=dev(around))
    expect, _ = oracle.light_walks(lights.view(np.uint8).reshape(-1), 1.0, uniform=around, threads=16)
    assert np.array_equal(chosen.cpu().numpy().view(np.uint32), expect)
    torch.cuda.synchronize()


def test_triangle_table_finds_the_first_light_of_a_triangle(engine, oracle):
    import torch
    rng = np.random.default_rng(5)
    count = 30000
    lights = light_list(rng.uniform(0.1, 1.0, count).astype(np.float32), rng, emissive_share=0.8, instances=5, primitives=4000)
    raw = dev(lights.view(np.uint8).reshape(-1))
    index = engine.LightIndex(raw)
    # every pair of the list (punctual lights' pairs included: they must not be found), then pairs at random
    instance = np.concatenate([lights["instance"], rng.integers(0, 8, 50000)]).astype(np.uint32)
    primitive = np.concatenate([lights["primitive"], rng.integers(0, 6000, 50000)]).astype(np.uint32)
    _, found = engine.light_query(raw, 1.0, index, instance=dev(instance.view(np.int32)), primitive=dev(primitive.view(np.int32)))
    _, walked = engine.light_query(raw, 1.0, None, instance=dev(instance.view(np.int32)), primitive=dev(primitive.view(np.int32)))
    _, expect = oracle.light_walks(lights.view(np.uint8).reshape(-1), 1.0, instance=instance, primitive=primitive, threads=16)
    assert np.array_equal(found.cpu().numpy().view(np.uint32), expect)
    assert np.array_equal(walked.cpu().numpy().view(np.uint32), expect)
    hit = expect != 0xffffffff
    assert 0.3 < hit.mean() < 0.95 and (expect[:count][lights["type"] == 3] <= np.arange(count)[lights["type"] == 3]).all()
    torch.cuda.synchronize()


def test_a_list_with_a_bad_weight_keeps_the_walk(engine, oracle):
    rng = np.random.default_rng(6)
    weights = rng.uniform(0.5, 1.5, 100).astype(np.float32)
    weights[40] = np.nan
    lights = light_list(weights, rng)
    raw = dev(lights.view(np.uint8).reshape(-1))
    index = engine.LightIndex(raw)
    assert int(index.words()[2]) == 0
    uniform = rng.random(4096, dtype=np.float32)
    chosen, _ = engine.light_query(raw, 100.0, index, uniform=dev(uniform))
    expect, _ = oracle.light_walks(lights.view(np.uint8).reshape(-1), 100.0, uniform=uniform)
    assert np.array_equal(chosen.cpu().numpy().view(np.uint32), expect)


def test_render_with_forty_thousand_emissive_triangles(engine, oracle, monkeypatch):
    """Two emissive icospheres of 20,480 triangles each light the Cornell box: 40,960 entries in the light
    list.  The render through the index equals the render through the walks bit for bit (two samples, all
    seven buffers), and the oracle's render - which walks - within the integrator tolerance."""
    from raytracer_amd import renderer as R
    R.load()
    scene = workloads.glowing_spheres_room()
    w, h = 96, 96

    def gpu(samples):
        r = R.Renderer(scene, w, h, 0)
        for _ in range(samples):
            r.render()
        r.synchronize()
        buffers = [R.device_to_host(r.accumulation_buffer(k), w * h * 16).view(np.float32).reshape(-1, 4).copy() for k in range(7)]
        return r, buffers

    r, through_index = gpu(2)
    assert r.launch_params().lightCount == 40960
    stats = r.statistics()
    assert stats["shadow_rays"] > 0 and stats["refused_references"] == 0
    monkeypatch.setenv("RT_LIGHT_INDEX", "0")
    _, through_walks = gpu(2)
    monkeypatch.delenv("RT_LIGHT_INDEX")
    for a, b in zip(through_index, through_walks):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    ref = oracle_scene(oracle, scene)
    setup = Setup(scene, w, h, final=False)
    for s in range(2):
        setup.params.sample = s
        ref.render_sample(setup.params, 0, w * h, 16)
    cpu, image = setup.aov[0], through_index[0]
    assert cpu[:, :3].mean() > 0.01 and np.isfinite(image).all()
    err = np.abs(image[:, :3].astype(np.float64) - cpu[:, :3])
    close = (err <= 2e-3 * np.maximum(np.abs(cpu[:, :3]), 1e-4)).all(axis=1).mean()
    assert close > 0.97, close
    keep = cpu[:, :3].max(axis=1) < np.percentile(cpu[:, :3].max(axis=1), 99.5)
    assert rel_rmse(image[keep, :3], cpu[keep, :3]) < 5e-2


def test_an_index_of_another_light_array_is_not_used(engine):
    """A wavefront state keeps the light index it was given; launch parameters whose light array is another one
    (here: a copy with other weights at another address, same count) must not be served from it.  Such a launch
    walks the list, so its buffers equal those of a state that never had an index, bit for bit."""
    import ctypes as C
    import torch
    from raytracer_amd import renderer as R
    R.load()
    scene = workloads.glowing_spheres_room()
    w, h = 64, 64
    r = R.Renderer(scene, w, h, 0)
    r.render()
    r.synchronize()
    params = r.launch_params()
    count = params.lightCount
    assert count == 40960
    lights = torch.from_numpy(R.device_to_host(params.lights, 92 * count).copy()).cuda()
    table = lights.view(torch.float32).view(count, 23)
    table[: count // 2, 16] *= 3.0                      # the weight (offset 64) of half the lights
    other = R.LaunchParams.from_buffer_copy(params)
    other.lights = lights.data_ptr()
    total = np.float32(0.0)
    for weight in table[:, 16].cpu().numpy():
        total = np.float32(total + weight)
    other.lightWeight = float(total)
    other.sample = 0
    index = engine.LightIndex(torch.from_numpy(R.device_to_host(params.lights, 92 * count).copy()).cuda())  # of the OLD weights
    lib = engine.load()
    images = []
    for with_stale_index in (False, True):
        state = C.c_void_p(0)
        engine.check(lib.rt_wavefront_create(w * h, C.byref(state)))
        try:
            if with_stale_index:
                engine.check(lib.rt_wavefront_set_light_index(state, C.c_void_p(index.memory.data_ptr()), count))
            for k in range(7):  # the running mean of sample 0 is old + (new - old): start every run from zero
                R.device_view(r.accumulation_buffer(k), w * h * 16).zero_()
            torch.cuda.synchronize()
            engine.check(lib.rt_render_sample(state, C.byref(other), 0, w * h, None))
            torch.cuda.synchronize()
            images.append(np.stack([R.device_to_host(r.accumulation_buffer(k), w * h * 16).copy() for k in range(7)]))
        finally:
            lib.rt_wavefront_destroy(state)
    differing = [int((images[0].view(np.uint32) != image.view(np.uint32)).sum()) for image in images]
    assert differing == [0, 0], differing
    assert images[0].view(np.float32).reshape(7, -1, 4)[0, :, :3].sum() > 0
