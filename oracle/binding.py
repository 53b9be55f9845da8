"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

_lib = None
_u32, _f32, _vp = C.c_uint32, C.c_float, C.c_void_p


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `make -C oracle`")
    lib = C.CDLL(LIB_PATH)
    sig = {
        "orc_photon_cell": (None, [_vp, _f32, _vp]),
        "orc_photon_bucket": (_u32, [_vp, _u32]),
        "orc_photon_keys": (None, [_vp, _u32, _f32, _u32, _vp]),
        "orc_build_photon_grid": (None, [_vp, _u32, _f32, _u32, _vp, _vp]),
        "orc_photon_gather": (None, [_vp, _vp, _u32, _f32, _vp, _vp, _u32, _vp, C.c_int]),
        "orc_shared_math": (None, [C.c_int, _vp, _vp, _u32, _vp]),
        "orc_seeded": (_u32, [_u32, _u32]),
        "orc_random": (_f32, [_vp]),
        "orc_scene_create": (_vp, []),
        "orc_scene_destroy": (None, [_vp]),
        "orc_scene_add_geometry": (C.c_int, [_vp, _vp, _u32, _vp, _u32, _u32]),
        "orc_scene_add_instance": (C.c_int, [_vp, _u32, _vp]),
        "orc_scene_commit": (None, [_vp]),
        "orc_scene_inverse": (None, [_vp, _u32, _vp]),
        "orc_trace_closest": (None, [_vp, _vp, _vp, _u32, C.c_int, C.c_int, _vp, _vp, _vp]),
        "orc_trace_any": (None, [_vp, _vp, _vp, _u32, C.c_int, C.c_int, _vp]),
        "orc_shade_hits": (None, [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp]),
        "orc_render_sample": (None, [_vp, _vp, _u32, _u32, C.c_int]),
        "orc_trace_photons": (_u32, [_vp, _vp, _u32, _u32, _vp, _u32, C.c_int]),
        "orc_trace_photons_indexed": (_u32, [_vp, _vp, _u32, _u32, _vp, _vp, _u32, C.c_int]),
        "orc_ray_counter": (C.c_uint64, [C.c_int]),
        "orc_light_walks": (None, [_vp, _u32, _f32, _vp, _vp, _vp, _u32, _vp, _vp, C.c_int]),
    }
    for name, (restype, argtypes) in sig.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def photon_cell(position, cell_size):
    p = _f32c(position).reshape(3)
    out = np.zeros(3, dtype=np.int32)
    load().orc_photon_cell(_p(p), np.float32(cell_size), _p(out))
    return out


def photon_bucket(cell, bucket_count):
    c = np.ascontiguousarray(cell, dtype=np.int32).reshape(3)
    return int(load().orc_photon_bucket(_p(c), bucket_count))


def photon_keys(positions, cell_size, bucket_count):
    p = _f32c(positions).reshape(-1, 3)
    keys = np.zeros(p.shape[0], dtype=np.uint32)
    load().orc_photon_keys(_p(p), p.shape[0], np.float32(cell_size), bucket_count, _p(keys))
    return keys


def build_photon_grid(photons, cell_size, bucket_count):
    """photons: (n, 9) float32 -> (offsets (B+1,) uint32, sorted (n, 9) float32)."""
    ph = _f32c(photons).reshape(-1, 9)
    offsets = np.zeros(bucket_count + 1, dtype=np.uint32)
    out = np.zeros_like(ph)
    load().orc_build_photon_grid(_p(ph), ph.shape[0], np.float32(cell_size), bucket_count, _p(offsets), _p(out))
    return offsets, out


def photon_gather(sorted_photons, cell_start, bucket_count, radius, qpos, qnormal, threads=1):
    sp = _f32c(sorted_photons).reshape(-1, 9)
    cs = np.ascontiguousarray(cell_start, dtype=np.uint32)
    qp = _f32c(qpos).reshape(-1, 3)
    qn = _f32c(qnormal).reshape(-1, 3)
    flux = np.zeros_like(qp)
    load().orc_photon_gather(_p(sp), _p(cs), bucket_count, np.float32(radius), _p(qp), _p(qn), qp.shape[0], _p(flux),
                             threads)
    return flux


def seeded(a, b):
    return int(load().orc_seeded(a & 0xffffffff, b & 0xffffffff))


def random_stream(state, count):
    """Returns (values, final_state) of `count` draws from orc_random."""
    s = C.c_uint32(state & 0xffffffff)
    lib = load()
    vals = np.array([lib.orc_random(C.byref(s)) for _ in range(count)], dtype=np.float32)
    return vals, s.value


def light_walks(lights, light_weight, uniform=None, instance=None, primitive=None, threads=1):
    """The reference's walks over a light list (device.cu:433-447, 607-621) for arrays of inputs.
    lights: uint8 array of n * 92 bytes (rt_gpu_light records).  Returns (chosen, found) uint32 arrays
    (None where the inputs were not given)."""
    lights = np.ascontiguousarray(lights, dtype=np.uint8)
    count = lights.size // 92
    n = len(uniform) if uniform is not None else len(instance)
    chosen = np.zeros(n, dtype=np.uint32) if uniform is not None else None
    found = np.zeros(n, dtype=np.uint32) if instance is not None else None
    keep = [np.ascontiguousarray(a, dtype=t) if a is not None else None
            for a, t in ((uniform, np.float32), (instance, np.uint32), (primitive, np.uint32))]
    load().orc_light_walks(_p(lights), count, light_weight, *[_p(a) if a is not None else None for a in keep], n,
                           _p(chosen) if chosen is not None else None, _p(found) if found is not None else None, threads)
    return chosen, found


def ray_counter(reset=False):
    return int(load().orc_ray_counter(int(reset)))


class Scene:
    """Oracle scene: geometries of rt_gpu_vertex (n, 16) float32 + (m, 3) uint32 indices."""

    def __init__(self):
        self.lib = load()
        self.handle = C.c_void_p(self.lib.orc_scene_create())
        self.committed = False

    def add_geometry(self, vertices, indices, material=0):
        v = _f32c(vertices)
        if v.shape[1] == 3:
            full = np.zeros((v.shape[0], 16), dtype=np.float32)
            full[:, :3] = v
            v = full
        assert v.shape[1] == 16
        idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        return self.lib.orc_scene_add_geometry(self.handle, _p(v), v.shape[0], _p(idx), idx.shape[0], material)

    def add_instance(self, geometry, transform=None):
        m = np.eye(4, dtype=np.float32).reshape(-1) if transform is None else _f32c(transform).reshape(16)
        return self.lib.orc_scene_add_instance(self.handle, geometry, _p(m))

    def commit(self):
        self.lib.orc_scene_commit(self.handle)
        self.committed = True

    def inverse(self, instance):
        out = np.zeros(12, dtype=np.float32)
        self.lib.orc_scene_inverse(self.handle, instance, _p(out))
        return out

    def trace_closest(self, org_tmin, dir_tmax, use_bvh=True, threads=1):
        o = _f32c(org_tmin).reshape(-1, 4)
        d = _f32c(dir_tmax).reshape(-1, 4)
        n = o.shape[0]
        tuv = np.zeros((n, 3), dtype=np.float32)
        prim = np.zeros(n, dtype=np.uint32)
        inst = np.zeros(n, dtype=np.uint32)
        self.lib.orc_trace_closest(self.handle, _p(o), _p(d), n, int(use_bvh), threads, _p(tuv), _p(prim), _p(inst))
        return tuv, prim, inst

    def trace_any(self, org_tmin, dir_tmax, use_bvh=True, threads=1):
        o = _f32c(org_tmin).reshape(-1, 4)
        d = _f32c(dir_tmax).reshape(-1, 4)
        occluded = np.zeros(o.shape[0], dtype=np.uint8)
        self.lib.orc_trace_any(self.handle, _p(o), _p(d), o.shape[0], int(use_bvh), threads, _p(occluded))
        return occluded

    def shade_hits(self, org_tmin, dir_tmax, tuv, prim, inst):
        o = _f32c(org_tmin).reshape(-1, 4)
        d = _f32c(dir_tmax).reshape(-1, 4)
        t = _f32c(tuv).reshape(-1, 3)
        p = np.ascontiguousarray(prim, dtype=np.uint32)
        i = np.ascontiguousarray(inst, dtype=np.uint32)
        hits = np.zeros((o.shape[0], 96), dtype=np.uint8)
        self.lib.orc_shade_hits(self.handle, _p(o), _p(d), _p(t), _p(p), _p(i), o.shape[0], _p(hits))
        return hits

    def render_sample(self, params, pixel_begin, pixel_end, threads=1):
        """params: a ctypes rt_launch_params whose pointers are HOST addresses."""
        self.lib.orc_render_sample(self.handle, C.byref(params), pixel_begin, pixel_end, threads)

    def trace_photons(self, params, first, count, capacity, threads=1):
        """Returns (photons (n, 9) float32 in emission order, wanted count)."""
        out = np.zeros((capacity, 9), dtype=np.float32)
        wanted = self.lib.orc_trace_photons(self.handle, C.byref(params), first, count, _p(out), capacity, threads)
        return out[:min(wanted, capacity)], int(wanted)

    def trace_photons_indexed(self, params, first, count, capacity, threads=1):
        """Returns (photons (n, 9) float32 in emission order, their emission indices (n,), wanted count)."""
        out = np.zeros((capacity, 9), dtype=np.float32)
        emission = np.zeros(capacity, dtype=np.uint32)
        wanted = self.lib.orc_trace_photons_indexed(self.handle, C.byref(params), first, count, _p(out), _p(emission),
                                                    capacity, threads)
        n = min(wanted, capacity)
        return out[:n], emission[:n], int(wanted)

    def close(self):
        if self.handle:
            self.lib.orc_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


SHARED_MATH = {"sine": 0, "cosine": 1, "power": 2, "arc_tangent2": 3, "arc_cosine": 4, "exp2": 5}


def shared_math(name, x, y=None):
    """include/rt_math.h compiled for the host: float32 arrays in, float32 array out."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y if y is not None else x, dtype=np.float32)
    out = np.empty_like(x)
    load().orc_shared_math(SHARED_MATH[name], _p(x), _p(y), x.size, _p(out))
    return out
