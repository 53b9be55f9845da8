"""CPU tests of the boundary: both libraries load without a GPU and export every symbol their headers
declare; struct layouts of the Python mirror match the C headers; the oracle never leaks into the product."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_engine_exports_every_declared_symbol():
    from raytracer_amd import engine
    lib = engine.load()  # raises if the library or any bound symbol is missing
    names = declared_functions("rt_engine.h")
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), name
    assert set(names) == set(engine.SYMBOLS), set(names) ^ set(engine.SYMBOLS)


def test_host_library_loads_and_struct_mirror_matches():
    from raytracer_amd import renderer as R
    R.load()
    assert C.sizeof(R.LaunchParams) == 344 and R.LaunchParams.scene.offset == 168
    assert R.LaunchParams.eye.offset == 176 and R.LaunchParams.sceneCenter.offset == 276
    assert R.LaunchParams.photonRadius.offset == 292 and R.LaunchParams.photonPass.offset == 336
    assert C.sizeof(R.GpuMaterial) == 272 and R.GpuMaterial.baseColorTexture.offset == 72
    assert C.sizeof(R.GpuLight) == 92 and R.GpuLight.type.offset == 88


def test_status_strings_and_invalid_arguments_without_a_gpu():
    from raytracer_amd import engine
    lib = engine.load()
    assert lib.rt_status_string(0) == b"success"
    assert lib.rt_status_string(-1) == b"invalid value"
    # argument validation happens before any device call
    assert lib.rt_photon_keys(None, 4, 0.1, 1000, None, None) == -1
    assert b"power of two" in lib.rt_last_error()
    assert lib.rt_photon_scan_storage(1 << 22) == lib.rt_photon_grid_scratch_bytes(1 << 23, 1 << 22)
    assert lib.rt_photon_grid_scratch_bytes(1 << 23, 1 << 22) >= 4 * 4 * (1 << 23)


def test_product_never_touches_the_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle" not in text.lower(), os.path.join(base, f)


def test_byte_over_255_without_the_division():
    """The texel fetch of the engine turns a byte into byte / 255.0f by q = byte * r, q + fma(-q, 255, byte) * r
    with r = 1.0f / 255.0f (raytracer_amd/csrc/shading.h, unitOfByte): evaluated here in exact arithmetic with
    one rounding to binary32 per operation, it equals the correctly rounded quotient for every byte."""
    from fractions import Fraction

    import numpy as np

    def rn32(x):  # round to nearest binary32, ties to even (normal range is all that occurs here)
        if x == 0:
            return Fraction(0)
        sign, x = (-1, -x) if x < 0 else (1, x)
        e = x.numerator.bit_length() - x.denominator.bit_length()
        if Fraction(2) ** e > x:
            e -= 1
        scaled = x / Fraction(2) ** (e - 23)          # in [2^23, 2^24)
        lower = scaled.numerator // scaled.denominator
        rest = scaled - lower
        if rest > Fraction(1, 2) or (rest == Fraction(1, 2) and lower % 2 == 1):
            lower += 1
        return sign * lower * Fraction(2) ** (e - 23)

    r = rn32(Fraction(1, 255))
    assert float(r) == float(np.float32(1.0) / np.float32(255.0))
    for byte in range(256):
        v = Fraction(byte)
        q = rn32(v * r)
        remainder = rn32(-q * 255 + v)
        formula = rn32(remainder * r + q)
        assert formula == rn32(v / 255), byte
        assert float(formula) == float(np.float32(byte) / np.float32(255.0)), byte
