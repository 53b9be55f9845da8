"""The elementary functions the shading code shares between the device and the CPU oracle (include/rt_math.h):
on the host against libm evaluated in binary64 (within one unit in the last place, special values as libm's),
and on the GPU against the host build of the same header, bit for bit."""
import numpy as np
import pytest

RNG = np.random.default_rng(5)


def ulps(a, b):
    """Distance in representable float32 values between a and b (same-sign finite values)."""
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia)
    ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


def samples():
    n = 200000
    angles = np.concatenate([RNG.uniform(-7.0, 7.0, n), RNG.uniform(-1000.0, 1000.0, n), 10.0 ** RNG.uniform(-30, 5, n),
                             np.arange(-64, 65) * (np.pi / 4), [0.0, -0.0]]).astype(np.float32)
    bases = np.concatenate([RNG.uniform(0.0, 1.0, n), 10.0 ** RNG.uniform(-30, 30, n), [0.0, 1.0, 1e-4, 0.5]]).astype(np.float32)
    exponents = np.concatenate([RNG.choice(np.float32([5.0, 2.4, 1.0 / 2.2, 0.5, 2.0]), n), RNG.uniform(-8.0, 8.0, n),
                                [0.0, 1.0, 5.0, 2.4]]).astype(np.float32)
    units = np.concatenate([RNG.uniform(-1.0, 1.0, n), 1.0 - 10.0 ** RNG.uniform(-8, 0, n), -1.0 + 10.0 ** RNG.uniform(-8, 0, n),
                            [1.0, -1.0, 0.0, -0.0]]).astype(np.float32)
    ys = np.concatenate([RNG.normal(size=n), 10.0 ** RNG.uniform(-20, 20, n) * RNG.choice([-1.0, 1.0], n),
                         [0.0, -0.0, 0.0, -0.0, 1.0, -1.0]]).astype(np.float32)
    xs = np.concatenate([RNG.normal(size=n), 10.0 ** RNG.uniform(-20, 20, n) * RNG.choice([-1.0, 1.0], n),
                         [0.0, 0.0, -0.0, -0.0, 0.0, 0.0]]).astype(np.float32)
    powers = RNG.uniform(-160.0, 130.0, n).astype(np.float32)
    return {"sine": (angles, None), "cosine": (angles, None), "power": (bases, exponents), "arc_tangent2": (ys, xs),
            "arc_cosine": (units, None), "exp2": (powers, None)}


LIBM = {"sine": lambda x, y: np.sin(x), "cosine": lambda x, y: np.cos(x), "power": lambda x, y: np.power(x, y),
        "arc_tangent2": lambda x, y: np.arctan2(x, y), "arc_cosine": lambda x, y: np.arccos(x), "exp2": lambda x, y: np.exp2(x)}


@pytest.mark.parametrize("name", list(LIBM))
def test_host_build_is_within_one_ulp_of_libm(oracle, name):
    x, y = samples()[name]
    got = oracle.shared_math(name, x, y)
    with np.errstate(all="ignore"):
        want = LIBM[name](x.astype(np.float64), None if y is None else y.astype(np.float64)).astype(np.float32)
    finite = np.isfinite(want) & np.isfinite(got)
    assert np.array_equal(np.isnan(want), np.isnan(got))
    assert np.array_equal(np.isinf(want), np.isinf(got)) and np.array_equal(want[np.isinf(want)], got[np.isinf(got)])
    distance = ulps(got[finite], want[finite])
    assert distance.max() <= 1, (name, int(distance.max()), x[finite][distance.argmax()])
    assert (distance == 0).mean() > 0.999   # binary64 rounded once: almost always the correctly rounded value


def test_values_the_shading_code_relies_on(oracle):
    one = np.float32([1.0])
    assert oracle.shared_math("cosine", np.float32([0.0]))[0] == 1.0 and oracle.shared_math("sine", np.float32([0.0]))[0] == 0.0
    assert oracle.shared_math("power", one, np.float32([2.4]))[0] == 1.0
    assert oracle.shared_math("power", np.float32([0.0]), np.float32([5.0]))[0] == 0.0
    assert oracle.shared_math("power", np.float32([0.0, 0.25]), np.float32([1.0 / 2.2, 0.5])).tolist() == [0.0, 0.5]
    assert oracle.shared_math("exp2", np.float32([0.0, 1.0, -1.0, 10.0])).tolist() == [1.0, 2.0, 0.5, 1024.0]
    assert oracle.shared_math("arc_cosine", np.float32([1.0]))[0] == 0.0
    assert oracle.shared_math("arc_cosine", np.float32([-1.0]))[0] == np.float32(np.pi)
    assert np.isnan(oracle.shared_math("sine", np.float32([np.inf, np.nan]))).all()
    assert np.isnan(oracle.shared_math("power", np.float32([-2.0]), np.float32([2.0]))).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(LIBM))
def test_device_and_host_give_the_same_bits(engine, oracle, name):
    import torch
    x, y = samples()[name]
    dx = torch.from_numpy(x).cuda()
    dy = torch.from_numpy(y).cuda() if y is not None else None
    got = engine.shared_math(name, dx, dy).cpu().numpy()
    want = oracle.shared_math(name, x, y)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
