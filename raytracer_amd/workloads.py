"""Seeded synthetic inputs for tests and benchmarks.

The reference ships no scenes (its .gitignore excludes them), so every workload
is generated here from a seed: photon sets and gather queries shaped like a
caustic on a table, and triangle scenes (Cornell box, a dense closed mesh that
stands in for the Stanford dragon, an instanced sphere field, a random triangle
soup).  Plain numpy; nothing here touches the GPU.
"""
import math

import numpy as np

# photon map constants of the reference (renderer.cpp:533-535)
PHOTON_BUCKETS = 1 << 22
PHOTON_CAPACITY = 1 << 23
PHOTON_EMISSIONS = 1 << 24


def caustic_radius(focus_distance, near, scene_radius, vertical_fov, height):
    """renderer.cpp:692-700: one-pixel footprint at the focus depth, floored at 0.0005."""
    depth = min(max(focus_distance, near), 2.0 * scene_radius)
    return max(2.0 * 1.0 * depth * math.tan(0.5 * vertical_fov) / height, 0.0005)


def caustic_photons(n, seed=2, extent=1.0, clustered=0.3):
    """n photons on the plane y = 0: uniform over [-extent, extent]^2, with a
    fraction drawn from thin ring-shaped arcs (the focused part of a caustic).
    Returns (n, 9) float32 rows: position, power, normal."""
    rng = np.random.default_rng(seed)
    n_cluster = int(n * clustered)
    n_uniform = n - n_cluster
    pos = np.zeros((n, 3), dtype=np.float32)
    pos[:n_uniform, 0] = rng.uniform(-extent, extent, n_uniform)
    pos[:n_uniform, 2] = rng.uniform(-extent, extent, n_uniform)
    arcs = rng.integers(0, 6, n_cluster)
    centre = np.stack([0.35 * np.cos(arcs * 1.3), 0.35 * np.sin(arcs * 2.1)], axis=1) * extent
    radius = (0.08 + 0.03 * arcs)[:, None] * extent
    angle = rng.uniform(0.0, 2.0 * math.pi, n_cluster)
    width = rng.normal(0.0, 0.002 * extent, n_cluster)
    ring = centre + (radius + width[:, None]) * np.stack([np.cos(angle), np.sin(angle)], axis=1)
    pos[n_uniform:, 0] = ring[:, 0]
    pos[n_uniform:, 2] = ring[:, 1]
    pos[:, 1] = rng.normal(0.0, 1e-5, n).astype(np.float32)
    power = rng.uniform(0.0, 1e-6, (n, 3)).astype(np.float32)
    normal = np.zeros((n, 3), dtype=np.float32)
    tilt = rng.normal(0.0, 0.05, (n, 2)).astype(np.float32)
    normal[:, 0] = tilt[:, 0]
    normal[:, 1] = 1.0
    normal[:, 2] = tilt[:, 1]
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    # a few photons face away (rejected by the 0.7 normal test)
    flip = rng.random(n) < 0.02
    normal[flip] *= -1.0
    perm = rng.permutation(n)  # emission order is unrelated to position
    return np.ascontiguousarray(np.concatenate([pos, power, normal], axis=1)[perm], dtype=np.float32)


def gather_queries(n, seed=7, extent=1.0):
    """n query points (primary hits on the table) with their shading normals."""
    rng = np.random.default_rng(seed)
    qpos = np.zeros((n, 3), dtype=np.float32)
    qpos[:, 0] = rng.uniform(-extent, extent, n)
    qpos[:, 2] = rng.uniform(-extent, extent, n)
    qpos[:, 1] = rng.normal(0.0, 1e-5, n)
    qn = np.zeros((n, 3), dtype=np.float32)
    qn[:, 1] = 1.0
    return qpos, qn


def hash_edge_positions(cell_size, seed=11, n=65536):
    """Positions that stress photonCell: negatives, exact cell boundaries, signed
    zeros, denormals and magnitudes where float->int saturates."""
    rng = np.random.default_rng(seed)
    cs = np.float32(cell_size)
    blocks = []
    blocks.append(rng.uniform(-10.0, 10.0, (n // 2, 3)))
    k = rng.integers(-3000, 3000, (n // 8, 3)).astype(np.float32)
    blocks.append(k * cs)                                   # on cell boundaries
    blocks.append(np.nextafter((k * cs).astype(np.float32), np.float32(-np.inf)))
    blocks.append(np.nextafter((k * cs).astype(np.float32), np.float32(np.inf)))
    special = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-38, -1e-38, 1e30, -1e30, 3e38, -3e38, 7.9e6, -7.9e6,
                        2147483648.0 * float(cs), -2147483648.0 * float(cs), 1e-8, -1e-8], dtype=np.float32)
    grid = np.stack(np.meshgrid(special, special, special, indexing="ij"), axis=-1).reshape(-1, 3)
    blocks.append(grid)
    out = np.concatenate([np.asarray(b, dtype=np.float32) for b in blocks], axis=0)
    return np.ascontiguousarray(out, dtype=np.float32)
