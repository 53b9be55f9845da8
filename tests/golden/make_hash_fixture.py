#!/usr/bin/env python3
"""Writes tests/golden/photon_hash_ref.npz: outputs of the REFERENCE's own photonCell / photonBucket
(/root/reference/src/photon_hash.cuh:10-24, compiled unmodified by `make -C oracle _ref` against NVIDIA's
cuda_runtime.h) for positions that stress the cell assignment.

Run in the build container only (the reference does not travel):

    make -C oracle _ref && python tests/golden/make_hash_fixture.py

Inputs per cell size: raytracer_amd.workloads.hash_edge_positions (random positions, exact cell boundaries
and their neighbouring floats, signed zeros, denormals, huge magnitudes) plus 65,536 random positions over
eleven decades of magnitude.  A position whose quotient p / cellSize is not inside (-2^31, 2^31), or is NaN,
is left out: the header's static_cast<int> of such a value is undefined behaviour in a host build (the GPU
saturates), so the reference run says nothing about it.  Three cell sizes (the reference's minimum radius
0.0005, the survey's 0.0037, and 0.25) x bucket counts 2^4, 2^14, 2^22.

The file holds, per cell size k: cellSize_k (float32), positions_k (n x 3 float32), cells_k (n x 3 int32,
the reference's photonCell), and keys_k_b for b in (4, 14, 22): the reference's photonBucket with 2^b
buckets (uint8 / uint16 / uint32).
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

CELL_SIZES = (0.0005, 0.0037, 0.25)
BUCKET_BITS = (4, 14, 22)


def reference_hash():
    path = os.path.join(ROOT, "oracle", "_ref", "libref_hash.so")
    if not os.path.exists(path):
        raise SystemExit(f"{path} is missing: run `make -C oracle _ref` where /root/reference and cuda_runtime.h exist")
    lib = C.CDLL(path)
    lib.ref_photon_hash.restype = None
    lib.ref_photon_hash.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_uint, C.c_void_p, C.c_void_p]
    return lib


def positions_for(cell_size, seed):
    from raytracer_amd import workloads
    rng = np.random.default_rng(seed)
    edge = workloads.hash_edge_positions(cell_size, n=32768)
    magnitude = np.float32(10.0) ** rng.uniform(-8.0, 3.0, (65536, 3)).astype(np.float32)
    wide = (magnitude * rng.choice(np.float32([-1.0, 1.0]), (65536, 3))).astype(np.float32)
    p = np.concatenate([edge, wide]).astype(np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        q = p / np.float32(cell_size)
    keep = np.all(np.isfinite(q) & (np.abs(q) < np.float32(2147483520.0)), axis=1)  # largest float below 2^31
    return np.ascontiguousarray(p[keep])


def main():
    lib = reference_hash()
    out = {}
    for k, cs in enumerate(CELL_SIZES):
        p = positions_for(cs, seed=100 + k)
        n = p.shape[0]
        cells = np.empty((n, 3), dtype=np.int32)
        out[f"cellSize_{k}"] = np.float32(cs)
        out[f"positions_{k}"] = p
        for bits in BUCKET_BITS:
            keys = np.empty(n, dtype=np.uint32)
            lib.ref_photon_hash(p.ctypes.data, n, np.float32(cs), 1 << bits, cells.ctypes.data, keys.ctypes.data)
            out[f"keys_{k}_{bits}"] = keys.astype(np.uint8 if bits <= 8 else np.uint16 if bits <= 16 else np.uint32)
        out[f"cells_{k}"] = cells
        print(f"cell size {cs}: {n} positions, cells {cells.min()} .. {cells.max()}")
    path = os.path.join(HERE, "photon_hash_ref.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path)} bytes")

    # the four known-answer rows of photon_hash_kat.json, from the same library
    kat = np.float32([[0.0, 0.0, 0.0], [-0.001, 0.5, 2.25], [1.234, -5.678, 9.0], [-1e-8, 1e-8, -0.0037]])
    cells = np.empty((4, 3), dtype=np.int32)
    keys = np.empty(4, dtype=np.uint32)
    lib.ref_photon_hash(kat.ctypes.data, 4, np.float32(0.0037), 1 << 22, cells.ctypes.data, keys.ctypes.data)
    for row in zip(kat.tolist(), cells.tolist(), keys.tolist()):
        print("known answer:", row)


if __name__ == "__main__":
    main()
