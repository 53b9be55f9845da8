"""CPU tests of the oracle's light walks (device.cu:433-447, 607-621 restated in oracle_render.cpp) against a
plain numpy float32 evaluation of the same loops, and of the size the engine states for a light index."""
import ctypes as C

import numpy as np

LIGHT = np.dtype([("a", "<f4", 3), ("b", "<f4", 3), ("c", "<f4", 3), ("normal", "<f4", 3), ("emission", "<f4", 3),
                  ("area", "<f4"), ("weight", "<f4"), ("range", "<f4"), ("innerCone", "<f4"), ("outerCone", "<f4"),
                  ("instance", "<u4"), ("primitive", "<u4"), ("type", "<u4")])


def walk(weights, target):
    """The reference's loop in float32: subtract, stop at the first remainder that is not positive."""
    rest = np.float32(target)
    for i, w in enumerate(weights):
        rest = np.float32(rest - w)
        if rest <= 0:
            return i
    return len(weights) - 1


def test_light_walks_follow_the_reference_loops(oracle):
    rng = np.random.default_rng(3)
    n = 300
    lights = np.zeros(n, dtype=LIGHT)
    lights["weight"] = np.exp(rng.normal(0.0, 4.0, n)).astype(np.float32)
    lights["weight"][rng.random(n) < 0.2] = 0.0
    lights["type"] = np.where(rng.random(n) < 0.7, 3, 1)
    lights["instance"] = rng.integers(0, 4, n)
    lights["primitive"] = rng.integers(0, 40, n)
    total = np.float32(0)
    for w in lights["weight"]:
        total = np.float32(total + w)
    uniform = rng.random(2000, dtype=np.float32)
    uniform[:3] = [0.0, 1.0 - 2.0 ** -24, 2.0 ** -24]
    raw = lights.view(np.uint8).reshape(-1)
    chosen, _ = oracle.light_walks(raw, float(total), uniform=uniform, threads=4)
    expect = [walk(lights["weight"], np.float32(u) * total) for u in uniform]
    assert np.array_equal(chosen, np.array(expect, dtype=np.uint32))
    instance = rng.integers(0, 5, 3000).astype(np.uint32)
    primitive = rng.integers(0, 45, 3000).astype(np.uint32)
    _, found = oracle.light_walks(raw, float(total), instance=instance, primitive=primitive, threads=3)
    for i, p, f in zip(instance, primitive, found):
        match = np.nonzero((lights["type"] == 3) & (lights["instance"] == i) & (lights["primitive"] == p))[0]
        assert f == (match[0] if len(match) else 0xffffffff)


def test_light_index_size_without_a_gpu():
    from raytracer_amd import engine
    lib = engine.load()
    size = C.c_size_t()
    for count, capacity in ((0, 2), (1, 2), (3, 8), (1000, 2048), (40960, 131072)):
        assert lib.rt_light_index_bytes(count, C.byref(size)) == 0
        words = 4 + 2 * count
        words += words & 1
        assert size.value == 4 * (words + 3 * capacity)
    assert lib.rt_light_index_bytes(5, None) != 0
    assert lib.rt_light_index_build(None, 0, None, 0, None) != 0   # no index memory
