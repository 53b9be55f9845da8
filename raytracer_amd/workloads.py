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


# ---------------------------------------------------------------------------
# Triangle scenes.  A scene is a dict:
#   geometries: list of {"vertices": (nv, 16) float32 rows laid out as the
#               64-byte GPU vertex (position 0:3, normal 3:6, tangent 8:12,
#               uv 12:14, uv1 14:16), "indices": (nt, 3) uint32, "material": int}
#   instances:  list of (geometry index, column-major 4x4 as 16 floats)
#   materials:  list of dicts (keys as the reference's Material, scene.hpp:52-73)
#   lights:     list of dicts (scene.hpp:114-123), cameras: list of dicts (:103-112)
# ---------------------------------------------------------------------------

def make_vertices(position, normal=None, uv=None):
    n = position.shape[0]
    v = np.zeros((n, 16), dtype=np.float32)
    v[:, 0:3] = position
    if normal is not None:
        v[:, 3:6] = normal
    if uv is not None:
        v[:, 12:14] = uv
    return v


def translation(x, y, z, scale=1.0):
    m = np.eye(4, dtype=np.float32) * np.float32(scale)
    m[3, 3] = 1.0
    m[0, 3], m[1, 3], m[2, 3] = x, y, z
    return np.ascontiguousarray(m.T).reshape(-1)  # column-major


def rotation_y(angle, x=0.0, y=0.0, z=0.0, scale=1.0):
    c, s = math.cos(angle), math.sin(angle)
    m = np.array([[c * scale, 0, s * scale, x], [0, scale, 0, y], [-s * scale, 0, c * scale, z], [0, 0, 0, 1]],
                 dtype=np.float32)
    return np.ascontiguousarray(m.T).reshape(-1)


def look_at(eye, target, up=(0.0, 1.0, 0.0)):
    """Camera-to-world, column-major; the camera looks down its -Z (renderer.cpp:672-677)."""
    eye = np.asarray(eye, dtype=np.float64)
    f = np.asarray(target, dtype=np.float64) - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, np.asarray(up, dtype=np.float64))
    r /= np.linalg.norm(r)
    u = np.cross(r, f)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = r, u, -f, eye
    return np.ascontiguousarray(m.T.astype(np.float32)).reshape(-1)


def default_material(**kw):
    m = dict(baseColor=(1.0, 1.0, 1.0, 1.0), emissive=(0.0, 0.0, 0.0), attenuationColor=(1.0, 1.0, 1.0),
             metallic=0.0, roughness=1.0, transmission=0.0, ior=1.5, thickness=0.0,
             attenuationDistance=float("inf"), emissiveStrength=1.0, dispersion=0.0)
    m.update(kw)
    return m


def quad(p0, p1, p2, p3):
    """Two triangles p0 p1 p2 / p0 p2 p3 with a flat normal."""
    p = np.array([p0, p1, p2, p3], dtype=np.float32)
    n = np.cross(p[1] - p[0], p[2] - p[0])
    n = n / np.linalg.norm(n)
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float32)
    return make_vertices(p, np.tile(n, (4, 1)), uv), np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32)


def box_mesh(lo, hi, faces=("x-", "x+", "y-", "y+", "z-", "z+")):
    lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
    c = lambda x, y, z: [hi[0] if x else lo[0], hi[1] if y else lo[1], hi[2] if z else lo[2]]
    sides = {
        "x-": (c(0, 0, 0), c(0, 0, 1), c(0, 1, 1), c(0, 1, 0)),
        "x+": (c(1, 0, 0), c(1, 1, 0), c(1, 1, 1), c(1, 0, 1)),
        "y-": (c(0, 0, 0), c(1, 0, 0), c(1, 0, 1), c(0, 0, 1)),
        "y+": (c(0, 1, 0), c(0, 1, 1), c(1, 1, 1), c(1, 1, 0)),
        "z-": (c(0, 0, 0), c(0, 1, 0), c(1, 1, 0), c(1, 0, 0)),
        "z+": (c(0, 0, 1), c(1, 0, 1), c(1, 1, 1), c(0, 1, 1)),
    }
    vs, idx = [], []
    for f in faces:
        v, i = quad(*sides[f])
        idx.append(i + 4 * len(vs))
        vs.append(v)
    return np.concatenate(vs), np.concatenate(idx)


def merge_meshes(meshes):
    vs, idx, base = [], [], 0
    for v, i in meshes:
        vs.append(v)
        idx.append(i + base)
        base += v.shape[0]
    return np.concatenate(vs), np.concatenate(idx).astype(np.uint32)


def cornell_box(light="point"):
    """~32 triangles: five walls (inward facing), a short and a tall box without
    bottoms, an optional emissive ceiling panel; one point light near the ceiling."""
    white = default_material(baseColor=(0.73, 0.73, 0.73, 1.0))
    red = default_material(baseColor=(0.65, 0.05, 0.05, 1.0))
    green = default_material(baseColor=(0.12, 0.45, 0.15, 1.0))
    materials = [default_material(metallic=1.0), white, red, green]  # material 0 = loader default
    geos, inst = [], []

    def add(mesh, material, transform=None):
        geos.append({"vertices": mesh[0], "indices": mesh[1], "material": material})
        inst.append((len(geos) - 1, np.eye(4, dtype=np.float32).reshape(-1) if transform is None else transform))

    # walls face inwards: build them as box faces seen from inside
    add(merge_meshes([quad([-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1])[::1],          # floor (flipped below)
                      quad([-1, 2, -1], [-1, 2, 1], [1, 2, 1], [1, 2, -1]),                # ceiling
                      quad([-1, 0, -1], [-1, 2, -1], [1, 2, -1], [1, 0, -1])]), 1)         # back wall
    add(quad([-1, 0, -1], [-1, 0, 1], [-1, 2, 1], [-1, 2, -1]), 2)                          # left, red
    add(quad([1, 0, -1], [1, 2, -1], [1, 2, 1], [1, 0, 1]), 3)                              # right, green
    short = box_mesh([-0.3, 0.0, -0.3], [0.3, 0.6, 0.3], faces=("x-", "x+", "y+", "z-", "z+"))
    tall = box_mesh([-0.3, 0.0, -0.3], [0.3, 1.2, 0.3], faces=("x-", "x+", "y+", "z-", "z+"))
    add(short, 1, rotation_y(-0.3, 0.35, 0.0, 0.35))
    add(tall, 1, rotation_y(0.35, -0.35, 0.0, -0.3))
    lights = []
    if light == "point":
        lights.append(dict(transform=translation(0.0, 1.8, 0.0), color=(1.0, 1.0, 1.0), intensity=3.0,
                           range=float("inf"), innerCone=0.0, outerCone=0.7853982, type=2))
    else:
        materials.append(default_material(baseColor=(0, 0, 0, 1), emissive=(1.0, 0.9, 0.7), emissiveStrength=15.0))
        add(quad([-0.25, 1.999, -0.25], [0.25, 1.999, -0.25], [0.25, 1.999, 0.25], [-0.25, 1.999, 0.25]), 4)
    camera = dict(transform=look_at([0.0, 1.0, 3.4], [0.0, 1.0, 0.0]), verticalFov=0.69, aspectRatio=0.0,
                  nearPlane=0.01, aperture=0.0, focusDistance=3.4)
    return dict(geometries=geos, instances=inst, materials=materials, lights=lights, cameras=[camera],
                environment=None)


def tube_mesh(segments_u, segments_v, seed=1, radius=0.16, scale=0.55):
    """Closed, noise-displaced tube swept along a trefoil knot: 2 * u * v triangles.
    Stands in for the Stanford dragon (similar triangle count, curved, self-occluding)."""
    rng = np.random.default_rng(seed)
    u = np.arange(segments_u, dtype=np.float64) / segments_u * 2.0 * math.pi
    v = np.arange(segments_v, dtype=np.float64) / segments_v * 2.0 * math.pi
    curve = np.stack([np.sin(u) + 2.0 * np.sin(2.0 * u), np.cos(u) - 2.0 * np.cos(2.0 * u), -np.sin(3.0 * u)], axis=1)
    curve *= scale / 3.0
    tangent = np.roll(curve, -1, axis=0) - np.roll(curve, 1, axis=0)
    tangent /= np.linalg.norm(tangent, axis=1, keepdims=True)
    ref = np.array([0.0, 0.0, 1.0])
    n1 = np.cross(tangent, ref)
    n1 /= np.linalg.norm(n1, axis=1, keepdims=True) + 1e-12
    n2 = np.cross(tangent, n1)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    # scale-like bumps plus a few low-frequency lobes
    phase = rng.uniform(0, 2 * math.pi, 6)
    bump = (1.0 + 0.10 * np.sin(36.0 * uu + phase[0]) * np.sin(14.0 * vv + phase[1])
            + 0.25 * np.sin(3.0 * uu + phase[2]) + 0.08 * np.sin(7.0 * uu + 2.0 * vv + phase[3]))
    r = radius * scale * bump
    pos = (curve[:, None, :] + r[:, :, None] * (np.cos(vv)[:, :, None] * n1[:, None, :] +
                                                 np.sin(vv)[:, :, None] * n2[:, None, :]))
    pos = pos.reshape(-1, 3)
    iu = np.arange(segments_u)
    iv = np.arange(segments_v)
    a = (iu[:, None] * segments_v + iv[None, :]).reshape(-1)
    b = (((iu + 1) % segments_u)[:, None] * segments_v + iv[None, :]).reshape(-1)
    c = (((iu + 1) % segments_u)[:, None] * segments_v + ((iv + 1) % segments_v)[None, :]).reshape(-1)
    d = (iu[:, None] * segments_v + ((iv + 1) % segments_v)[None, :]).reshape(-1)
    idx = np.concatenate([np.stack([a, b, c], axis=1), np.stack([a, c, d], axis=1)]).astype(np.uint32)
    # smooth normals: area-weighted face normals accumulated on the vertices
    p = pos.astype(np.float32)
    fn = np.cross(p[idx[:, 1]] - p[idx[:, 0]], p[idx[:, 2]] - p[idx[:, 0]])
    nrm = np.zeros_like(p)
    for k in range(3):
        np.add.at(nrm, idx[:, k], fn)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True) + 1e-20
    uv = np.stack([uu.reshape(-1) / (2 * math.pi), vv.reshape(-1) / (2 * math.pi)], axis=1)
    return make_vertices(p, nrm, uv.astype(np.float32)), idx


def dragon_cornell(triangles=871200, seed=1):
    """Cornell box holding a ~870k-triangle absorbing-glass mesh ("dragon-standin")."""
    scene = cornell_box(light="panel")
    # keep the walls, drop the two boxes (instances 3, 4)
    scene["geometries"] = scene["geometries"][:3] + scene["geometries"][5:]
    scene["instances"] = [(0, scene["instances"][0][1]), (1, scene["instances"][1][1]),
                          (2, scene["instances"][2][1]), (3, scene["instances"][5][1])]
    v_seg = max(8, int(round(math.sqrt(triangles / 8.0))))
    u_seg = max(8, triangles // (2 * v_seg))
    mesh = tube_mesh(u_seg, v_seg, seed=seed)
    scene["materials"].append(default_material(baseColor=(0.9, 0.8, 1.0, 1.0), roughness=0.05, transmission=1.0,
                                               attenuationColor=(0.55, 0.25, 0.8), attenuationDistance=0.25))
    scene["geometries"].append({"vertices": mesh[0], "indices": mesh[1], "material": len(scene["materials"]) - 1})
    scene["instances"].append((len(scene["geometries"]) - 1, rotation_y(0.4, 0.0, 1.0, 0.0, 1.3)))
    return scene


def icosphere(subdivisions=4):
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(subdivisions):
        edges = {}
        verts = list(v)

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in edges:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                edges[key] = len(verts) - 1
            return edges[key]

        nf = []
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        v, f = np.array(verts), np.array(nf, dtype=np.int64)
    p = v.astype(np.float32)
    return make_vertices(p, p.copy()), f.astype(np.uint32)


def sphere_field(count=500, seed=3, subdivisions=4, sky=False):
    """One icosphere geometry instanced `count` times on a disc plus three large
    spheres and a ground quad: a two-level (instanced) scene seen through a thin
    lens.  sky=True adds a latitude/longitude sky with a sun as environment light."""
    rng = np.random.default_rng(seed)
    sphere = icosphere(subdivisions)
    ground = quad([-30, 0, -30], [-30, 0, 30], [30, 0, 30], [30, 0, -30])
    materials = [default_material(metallic=1.0),
                 default_material(baseColor=(0.5, 0.5, 0.5, 1.0)),
                 default_material(baseColor=(0.8, 0.3, 0.2, 1.0)),
                 default_material(baseColor=(0.9, 0.9, 0.9, 1.0), metallic=1.0, roughness=0.1),
                 default_material(baseColor=(1.0, 1.0, 1.0, 1.0), transmission=1.0, roughness=0.0)]
    geos = [{"vertices": ground[0], "indices": ground[1], "material": 1},
            {"vertices": sphere[0], "indices": sphere[1], "material": 2},
            {"vertices": sphere[0], "indices": sphere[1], "material": 3},
            {"vertices": sphere[0], "indices": sphere[1], "material": 4}]
    inst = [(0, np.eye(4, dtype=np.float32).reshape(-1))]
    for k, x in enumerate((-4.0, 0.0, 4.0)):
        inst.append((1 + k, translation(x, 1.0, 0.0, 1.0)))
    r = 11.0 * np.sqrt(rng.random(count))
    a = rng.uniform(0, 2 * math.pi, count)
    for i in range(count):
        inst.append((1 + int(rng.integers(0, 3)),
                     rotation_y(float(rng.uniform(0, 6.28)), float(r[i] * math.cos(a[i])), 0.2,
                                float(r[i] * math.sin(a[i])), 0.2)))
    camera = dict(transform=look_at([13.0, 2.0, 3.0], [0.0, 0.0, 0.0]), verticalFov=0.35, aspectRatio=0.0,
                  nearPlane=0.01, aperture=0.1, focusDistance=10.0)
    lights = [dict(transform=look_at([5.0, 12.0, 5.0], [0.0, 0.0, 0.0]), color=(1.0, 1.0, 1.0), intensity=3.0,
                   range=float("inf"), innerCone=0.0, outerCone=0.7853982, type=0)]
    environment = dict(pixels=sky_environment(128, 64), rotation=0.7, strength=1.0, exposure=0.0) if sky else None
    return dict(geometries=geos, instances=inst, materials=materials, lights=lights, cameras=[camera],
                environment=environment)


def glowing_spheres_room(subdivisions=5, spheres=2):
    """The Cornell box lit by emissive icospheres instead of its point light: 20 * 4^subdivisions emissive
    triangles per sphere (20,480 at the default), every one of them an entry of the light list."""
    scene = cornell_box(light="point")
    scene["lights"] = []
    sphere = icosphere(subdivisions)
    scene["materials"].append(default_material(baseColor=(0, 0, 0, 1), emissive=(1.0, 0.8, 0.6), emissiveStrength=6.0))
    scene["materials"].append(default_material(baseColor=(0, 0, 0, 1), emissive=(0.5, 0.7, 1.0), emissiveStrength=9.0))
    first = len(scene["materials"]) - 2
    places = [(-0.45, 1.55, 0.1, 0.22), (0.5, 1.3, -0.2, 0.15), (0.0, 0.35, 0.6, 0.1)]
    for k in range(spheres):
        scene["geometries"].append({"vertices": sphere[0], "indices": sphere[1], "material": first + k % 2})
        x, y, z, r = places[k % len(places)]
        scene["instances"].append((len(scene["geometries"]) - 1, translation(x, y, z, r)))
    return scene


def triangle_soup(n, seed=4):
    """n unshared triangles: centroids uniform in the unit cube, edge length
    ~ U(0.5, 1.5) * n^(-1/3)."""
    rng = np.random.default_rng(seed)
    c = rng.random((n, 3), dtype=np.float32)
    size = (rng.uniform(0.5, 1.5, (n, 1)) * n ** (-1.0 / 3.0)).astype(np.float32)
    offs = rng.normal(0.0, 1.0, (n, 3, 3)).astype(np.float32)
    offs /= np.linalg.norm(offs, axis=2, keepdims=True)
    p = (c[:, None, :] + 0.5 * size[:, None, :] * offs).reshape(-1, 3)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)
    return dict(geometries=[{"vertices": make_vertices(p), "indices": idx, "material": 0}],
                instances=[(0, np.eye(4, dtype=np.float32).reshape(-1))], materials=[default_material()],
                lights=[], cameras=[], environment=None)


def scene_bounds(scene):
    lo = np.full(3, np.inf)
    hi = np.full(3, -np.inf)
    for g, m in scene["instances"]:
        p = scene["geometries"][g]["vertices"][:, :3].astype(np.float64)
        t = np.asarray(m, dtype=np.float64).reshape(4, 4).T
        w = p @ t[:3, :3].T + t[:3, 3]
        lo = np.minimum(lo, w.min(0))
        hi = np.maximum(hi, w.max(0))
    return lo, hi


def random_rays(scene, n, seed=5, tmin=0.001, tmax=1e16):
    """Incoherent rays: origins uniform in the (slightly grown) scene box,
    directions uniform on the sphere.  Returns (org_tmin, dir_tmax), (n, 4) each."""
    rng = np.random.default_rng(seed)
    lo, hi = scene_bounds(scene)
    pad = 0.05 * (hi - lo)
    o = rng.uniform(lo - pad, hi + pad, (n, 3))
    d = rng.normal(0.0, 1.0, (n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    org = np.concatenate([o, np.full((n, 1), tmin)], axis=1).astype(np.float32)
    dr = np.concatenate([d, np.full((n, 1), tmax)], axis=1).astype(np.float32)
    return np.ascontiguousarray(org), np.ascontiguousarray(dr)


def camera_rays(scene, width, height, seed=0, camera=0):
    """Pinhole primary rays through jittered pixel centres, row-major from the bottom row."""
    cam = scene["cameras"][camera]
    m = np.asarray(cam["transform"], dtype=np.float64).reshape(4, 4).T
    right, up, back, eye = m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3]
    scale = math.tan(0.5 * cam["verticalFov"])
    aspect = cam["aspectRatio"] if cam["aspectRatio"] > 0 else width / height
    rng = np.random.default_rng(seed)
    x = (np.arange(width)[None, :] + rng.random((height, width))) / width
    y = (np.arange(height)[:, None] + rng.random((height, width))) / height
    d = (-back[None, None, :] + (2 * x - 1)[:, :, None] * right * scale * aspect
         + (2 * y - 1)[:, :, None] * up * scale)
    d /= np.linalg.norm(d, axis=2, keepdims=True)
    n = width * height
    org = np.concatenate([np.tile(eye, (n, 1)), np.full((n, 1), 0.001)], axis=1).astype(np.float32)
    dr = np.concatenate([d.reshape(-1, 3), np.full((n, 1), 1e16)], axis=1).astype(np.float32)
    return np.ascontiguousarray(org), np.ascontiguousarray(dr)


def torus_mesh(major=0.25, minor=0.03, segments_u=96, segments_v=24):
    u = np.arange(segments_u) / segments_u * 2.0 * math.pi
    v = np.arange(segments_v) / segments_v * 2.0 * math.pi
    uu, vv = np.meshgrid(u, v, indexing="ij")
    ring = major + minor * np.cos(vv)
    pos = np.stack([ring * np.cos(uu), minor * np.sin(vv), ring * np.sin(uu)], axis=-1).reshape(-1, 3)
    nrm = np.stack([np.cos(vv) * np.cos(uu), np.sin(vv), np.cos(vv) * np.sin(uu)], axis=-1).reshape(-1, 3)
    iu, iv = np.arange(segments_u), np.arange(segments_v)
    a = (iu[:, None] * segments_v + iv[None, :]).reshape(-1)
    b = (((iu + 1) % segments_u)[:, None] * segments_v + iv[None, :]).reshape(-1)
    c = (((iu + 1) % segments_u)[:, None] * segments_v + ((iv + 1) % segments_v)[None, :]).reshape(-1)
    d = (iu[:, None] * segments_v + ((iv + 1) % segments_v)[None, :]).reshape(-1)
    idx = np.concatenate([np.stack([a, c, b], axis=1), np.stack([a, d, c], axis=1)]).astype(np.uint32)
    uv = np.stack([uu.reshape(-1) / (2 * math.pi), vv.reshape(-1) / (2 * math.pi)], axis=1)
    return make_vertices(pos.astype(np.float32), nrm.astype(np.float32), uv.astype(np.float32)), idx


def checker_image(size=64, a=(200, 60, 60, 255), b=(240, 240, 240, 255), squares=8):
    y, x = np.mgrid[0:size, 0:size]
    mask = ((x * squares // size) + (y * squares // size)) % 2 == 0
    img = np.where(mask[:, :, None], np.array(a, np.uint8), np.array(b, np.uint8)).astype(np.uint8)
    return np.ascontiguousarray(img)


def sky_environment(width=64, height=32, sun=(0.3, 0.6)):
    """Small latitude/longitude sky: blue-to-white gradient with a bright sun disc."""
    v, u = np.mgrid[0:height, 0:width]
    theta = (v + 0.5) / height * math.pi
    up = np.cos(theta)
    sky = np.stack([0.35 + 0.25 * (1 - up), 0.5 + 0.2 * (1 - up), 0.9 * np.ones_like(up), np.ones_like(up)], -1)
    su, sv = sun[0] * width, (1 - sun[1]) * 0.5 * height
    d2 = (u - su) ** 2 + (v - sv) ** 2
    sky[..., :3] += (40.0 * np.exp(-d2 / 1.5))[..., None]
    return np.ascontiguousarray(sky.astype(np.float32))


def caustic_lab(glass_subdivisions=3):
    """Diffuse table, a gold ring (metallic mirror branch), a faceted glass ball
    (transmission) and one spot light from above: caustics on the table."""
    table = quad([-1, 0, -1], [-1, 0, 1], [1, 0, 1], [1, 0, -1])
    ring = torus_mesh()
    ball = icosphere(glass_subdivisions)
    materials = [default_material(metallic=1.0),
                 default_material(baseColor=(0.8, 0.8, 0.8, 1.0)),
                 default_material(baseColor=(1.0, 0.78, 0.34, 1.0), metallic=1.0, roughness=0.05),
                 default_material(baseColor=(1.0, 1.0, 1.0, 1.0), transmission=1.0, roughness=0.0, ior=1.5,
                                  dispersion=0.02)]
    geos = [{"vertices": table[0], "indices": table[1], "material": 1},
            {"vertices": ring[0], "indices": ring[1], "material": 2},
            {"vertices": ball[0], "indices": ball[1], "material": 3}]
    inst = [(0, np.eye(4, dtype=np.float32).reshape(-1)),
            (1, translation(-0.3, 0.031, 0.0)),
            (2, translation(0.3, 0.15, 0.05, 0.15))]
    lights = [dict(transform=look_at([0.6, 1.5, 0.4], [0.0, 0.0, 0.0]), color=(1.0, 0.97, 0.9), intensity=60.0,
                   range=float("inf"), innerCone=0.25, outerCone=0.45, type=1)]
    camera = dict(transform=look_at([0.0, 0.9, 1.3], [0.0, 0.05, 0.0]), verticalFov=0.6, aspectRatio=0.0,
                  nearPlane=0.01, aperture=0.0, focusDistance=1.5)
    return dict(geometries=geos, instances=inst, materials=materials, lights=lights, cameras=[camera],
                environment=None)


def textured_sky_scene():
    """Checker-textured floor, a rough metal and a glass ball under a sky environment
    with a sun, seen through a thin lens: textures, environment sampling, MIS."""
    floor = quad([-2, 0, -2], [-2, 0, 2], [2, 0, 2], [2, 0, -2])
    ball = icosphere(3)
    materials = [default_material(metallic=1.0),
                 default_material(baseColorTexture=dict(texture=0, scale=(4.0, 4.0), rotation=0.3)),
                 default_material(baseColor=(0.9, 0.9, 0.95, 1.0), metallic=1.0, roughness=0.35),
                 default_material(transmission=0.9, roughness=0.1, ior=1.45, attenuationColor=(0.4, 0.9, 0.5),
                                  attenuationDistance=0.5)]
    geos = [{"vertices": floor[0], "indices": floor[1], "material": 1},
            {"vertices": ball[0], "indices": ball[1], "material": 2},
            {"vertices": ball[0], "indices": ball[1], "material": 3}]
    inst = [(0, np.eye(4, dtype=np.float32).reshape(-1)), (1, translation(-0.6, 0.5, 0.0, 0.5)),
            (2, rotation_y(0.5, 0.6, 0.4, 0.2, 0.4))]
    camera = dict(transform=look_at([0.0, 1.2, 3.0], [0.0, 0.4, 0.0]), verticalFov=0.6, aspectRatio=0.0,
                  nearPlane=0.01, aperture=0.02, focusDistance=3.0)
    return dict(geometries=geos, instances=inst, materials=materials, lights=[], cameras=[camera],
                images=[checker_image()], textures=[dict(image=0)],
                environment=dict(pixels=sky_environment(), rotation=0.4, strength=1.2, exposure=-0.5))


def material_map_scene(seed=11, size=32, checker=64):
    """Every texture slot of a material at once: a floor with colour, normal and metallic-roughness maps
    (the normal map through the second texture coordinate set, clamped and mirrored wrap modes), an
    emissive panel whose emission comes from a striped map, and a glass ball whose transmission and
    thickness are modulated by maps; tangents and both coordinate sets are filled in."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:size, 0:size]  # `size`, `checker`: edge lengths of the maps (powers of two or not)
    bumps = np.stack([128 + 80 * np.sin(x * 0.8), 128 + 80 * np.cos(y * 0.6), np.full_like(x, 230.0),
                      np.full_like(x, 255.0)], -1)
    normal_map = np.ascontiguousarray(np.clip(bumps, 0, 255).astype(np.uint8))
    rough_metal = np.ascontiguousarray(rng.integers(20, 256, (size, size, 4), dtype=np.uint8))
    stripes = np.zeros((size, size, 4), np.uint8)
    stripes[:, (x[0] // 4) % 2 == 0] = (255, 180, 90, 255)
    stripes[..., 3] = 255
    fade = np.zeros((size, size, 4), np.uint8)
    fade[..., 0] = (x * 255 // (size - 1)).astype(np.uint8)          # transmission reads red
    fade[..., 1] = (255 - y * 200 // (size - 1)).astype(np.uint8)    # thickness reads green
    fade[..., 2:] = 255
    images = [checker_image(size=checker), normal_map, rough_metal, np.ascontiguousarray(stripes), np.ascontiguousarray(fade)]
    textures = [dict(image=0, wrapU=33071, wrapV=33071), dict(image=1, wrapU=10497, wrapV=33648),
                dict(image=2, wrapU=33648, wrapV=10497), dict(image=3), dict(image=4)]

    floor = quad([-2, 0, -2], [-2, 0, 2], [2, 0, 2], [2, 0, -2])
    floor[0][:, 8:12] = (0.0, 0.0, 1.0, 1.0)
    floor[0][:, 12:14] = floor[0][:, 12:14] * 1.6 - 0.3         # runs past [0, 1]: the wrap modes matter
    floor[0][:, 14:16] = floor[0][:, 12:14][:, ::-1] * 2.5
    panel = quad([-0.8, 1.6, -1.2], [0.8, 1.6, -1.2], [0.8, 0.4, -1.2], [-0.8, 0.4, -1.2])
    ball = icosphere(3)
    p = ball[0][:, 0:3]
    ball[0][:, 12] = np.arctan2(p[:, 2], p[:, 0]) / (2.0 * math.pi) + 0.5
    ball[0][:, 13] = np.arccos(np.clip(p[:, 1], -1.0, 1.0)) / math.pi
    ball[0][:, 14:16] = ball[0][:, 12:14] * 3.0
    tangent = np.stack([-p[:, 2], np.zeros(len(p)), p[:, 0]], 1)
    tangent /= np.maximum(np.linalg.norm(tangent, axis=1, keepdims=True), 1e-6)
    ball[0][:, 8:11] = tangent
    ball[0][:, 11] = -1.0

    materials = [default_material(metallic=1.0),
                 default_material(metallic=1.0, roughness=1.0,
                                  baseColorTexture=dict(texture=0, scale=(1.0, 1.0)),
                                  normalTexture=dict(texture=1, texCoord=1, strength=0.7, rotation=0.2),
                                  metallicRoughnessTexture=dict(texture=2, offset=(0.25, -0.4), scale=(2.0, 3.0))),
                 default_material(baseColor=(0.1, 0.1, 0.1, 1.0), emissive=(1.0, 1.0, 1.0), emissiveStrength=4.0,
                                  emissiveTexture=dict(texture=3, scale=(2.0, 1.0))),
                 default_material(baseColor=(0.95, 0.95, 1.0, 1.0), roughness=0.15, transmission=1.0, ior=1.5,
                                  thickness=0.4, attenuationColor=(0.3, 0.8, 0.6), attenuationDistance=0.6,
                                  transmissionTexture=dict(texture=4), thicknessTexture=dict(texture=4, texCoord=1),
                                  normalTexture=dict(texture=1, scale=(2.0, 2.0), strength=0.4))]
    geos = [{"vertices": floor[0], "indices": floor[1], "material": 1},
            {"vertices": panel[0], "indices": panel[1], "material": 2},
            {"vertices": ball[0], "indices": ball[1], "material": 3}]
    inst = [(0, np.eye(4, dtype=np.float32).reshape(-1)), (1, np.eye(4, dtype=np.float32).reshape(-1)),
            (2, rotation_y(0.7, 0.3, 0.55, 0.3, 0.55))]
    lights = [dict(transform=translation(-0.8, 2.2, 1.2), color=(1.0, 0.95, 0.9), intensity=6.0,
                   range=float("inf"), innerCone=0.0, outerCone=0.7853982, type=2)]
    camera = dict(transform=look_at([0.2, 1.3, 3.2], [0.0, 0.5, 0.0]), verticalFov=0.62, aspectRatio=0.0,
                  nearPlane=0.01, aperture=0.0, focusDistance=3.0)
    return dict(geometries=geos, instances=inst, materials=materials, lights=lights, cameras=[camera],
                images=images, textures=textures, environment=None)
