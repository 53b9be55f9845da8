"""ctypes binding of librt_engine.so (the C ABI declared in include/rt_engine.h).

Device memory is handed over as torch tensors on a HIP device (their
data_ptr()); torch is plumbing here, the kernels are the library's own.  There
is no CPU fallback: loading fails loudly when the library has not been built.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librt_engine.so")


class EngineError(RuntimeError):
    pass


class RtBvhStats(C.Structure):
    _fields_ = [
        ("nodeCount", C.c_uint64),
        ("nodeBytes", C.c_uint64),
        ("triangleCount", C.c_uint64),
        ("triangleBytes", C.c_uint64),
        ("instanceCount", C.c_uint32),
        ("maxDepth", C.c_uint32),
        ("buildMilliseconds", C.c_float),
    ]


class RtInstanceDesc(C.Structure):
    _fields_ = [
        ("transform", C.c_float * 12),
        ("instanceId", C.c_uint32),
        ("geometry", C.c_uint32),
        ("blas", C.c_void_p),
    ]


# name -> (restype, argtypes); every symbol include/rt_engine.h declares
_u32, _f32, _vp, _sz = C.c_uint32, C.c_float, C.c_void_p, C.c_size_t
SYMBOLS = {
    "rt_last_error": (C.c_char_p, []),
    "rt_status_string": (C.c_char_p, [C.c_int]),
    "rt_photon_grid_scratch_bytes": (_sz, [_u32, _u32]),
    "rt_photon_scan_storage": (_sz, [_u32]),
    "rt_build_photon_grid": (C.c_int, [_vp, _u32, _f32, _u32, _vp, _vp, _vp, _sz, _vp]),
    "rt_build_photon_grid_async": (C.c_int, [_vp, _u32, _f32, _u32, _vp, _vp, _vp, _sz, _vp, _vp]),
    "rt_build_photon_grid_static": (C.c_int, [_vp, _u32, _f32, _u32, _vp, _vp, _vp]),
    "rt_build_photon_grid_counted": (C.c_int, [_vp, _vp, _u32, _f32, _u32, _vp, _vp, _sz, _vp, C.c_int, _vp]),
    "rt_photon_grid_keys": (_vp, [_vp, _sz, _u32, _u32]),
    "rt_wavefront_set_photon_keys": (C.c_int, [_vp, _vp]),
    "rt_photon_keys": (C.c_int, [_vp, _u32, _f32, _u32, _vp, _vp]),
    "rt_photon_gather": (C.c_int, [_vp, _vp, _u32, _f32, _vp, _vp, _u32, _vp, _vp]),
    "rt_blas_build": (C.c_int, [_vp, _u32, _u32, _vp, _u32, _vp, C.POINTER(_vp)]),
    "rt_blas_destroy": (None, [_vp]),
    "rt_scene_build": (C.c_int, [C.POINTER(RtInstanceDesc), _u32, _vp, C.POINTER(_vp)]),
    "rt_scene_destroy": (None, [_vp]),
    "rt_scene_handle": (C.c_uint64, [_vp]),
    "rt_scene_instance_inverse": (C.c_int, [_vp, _u32, C.POINTER(C.c_float)]),
    "rt_blas_stats": (C.c_int, [_vp, C.POINTER(RtBvhStats)]),
    "rt_scene_stats": (C.c_int, [_vp, C.POINTER(RtBvhStats)]),
    "rt_scene_folded": (C.c_int, [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rt_trace_closest": (C.c_int, [_vp, _vp, _vp, _u32, _vp, _vp, _vp]),
    "rt_trace_any": (C.c_int, [_vp, _vp, _vp, _u32, _vp, _vp]),
    "rt_trace_closest_counted": (C.c_int, [_vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp]),
    "rt_photon_gather_counted": (C.c_int, [_vp, _vp, _u32, _f32, _vp, _vp, _u32, _vp, _vp, _vp]),
    "rt_photon_bucket_bitmap": (C.c_int, [_vp, _u32, _vp, _vp]),
    "rt_photon_gather_masked": (C.c_int, [_vp, _vp, _vp, _u32, _f32, _vp, _vp, _u32, _vp, _vp]),
    "rt_scene_set_geometry": (C.c_int, [_vp, _vp, _u32]),
    "rt_wavefront_create": (C.c_int, [_u32, C.POINTER(_vp)]),
    "rt_wavefront_destroy": (None, [_vp]),
    "rt_wavefront_create_for": (C.c_int, [_u32, _u32, C.POINTER(_vp)]),
    "rt_wavefront_state_bytes": (C.c_int, [_u32, _u32, C.POINTER(C.c_size_t)]),
    "rt_wavefront_set_photon_maps": (C.c_int, [_vp, _vp, _u32, _u32, _u32]),
    "rt_render_sample": (C.c_int, [_vp, _vp, _u32, _u32, _vp]),
    "rt_render_sample_seeded": (C.c_int, [_vp, _vp, _u32, _u32, _u32, _vp]),
    "rt_render_sample_bands": (C.c_int, [_vp, _vp, _u32, _u32, _u32, _u32, _vp]),
    "rt_trace_photons": (C.c_int, [_vp, _vp, _u32, _u32, _vp]),
    "rt_trace_photon_passes": (C.c_int, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp]),
    "rt_wavefront_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.c_int, _vp]),
    "rt_wavefront_configure": (C.c_int, [_vp, _u32]),
    "rt_wavefront_set_ray_share": (C.c_int, [_vp, _u32, _u32]),
    "rt_wavefront_prepare_stream": (C.c_int, [_vp, _vp]),
    "rt_wavefront_set_bucket_bitmap": (C.c_int, [_vp, _vp]),
    "rt_light_index_bytes": (C.c_int, [_u32, C.POINTER(C.c_size_t)]),
    "rt_light_index_build": (C.c_int, [_vp, _u32, _vp, C.c_size_t, _vp]),
    "rt_wavefront_set_light_index": (C.c_int, [_vp, _vp, _u32]),
    "rt_light_query": (C.c_int, [_vp, _u32, C.c_float, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp]),
    "rt_wavefront_set_accumulate_events": (C.c_int, [_vp, _vp, _vp]),
    "rt_wavefront_set_sample_batch": (C.c_int, [_vp, _u32, _u32]),
    "rt_shared_math": (C.c_int, [C.c_int, _vp, _vp, _u32, _vp, _vp]),
    "rt_streams_run_concurrently": (C.c_int, [_vp, _u32, _vp]),
    "rt_wavefront_timings": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(_u32), _vp]),
}

_lib = None


def load():
    """Loads the engine library; raises EngineError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` "
            "(or __graft_entry__.build()); there is no fallback path")
    # torch first: its wheel carries a HIP runtime of its own under the same library name, and a
    # process that loads the system's copy before torch's finds no device from torch afterwards
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if os.environ.get("RT_ENGINE_ALLOW_MISSING") == "1":  # development aid only
                continue
            raise EngineError(f"{LIB_PATH} does not export {name}: rebuild it")
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status):
    if status != 0:
        lib = load()
        raise EngineError(
            f"rt_status {status} ({lib.rt_status_string(status).decode()}): {lib.rt_last_error().decode()}")


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream(stream):
    if stream is None:
        return C.c_void_p(0)
    return C.c_void_p(stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream))


PHOTON_FLOATS = 9  # rt_photon: position, power, normal
STATIC_GRID_MAX_PHOTONS = 1 << 20  # RT_PHOTON_GRID_STATIC_MAX_PHOTONS


def photon_keys(positions, cell_size, bucket_count, stream=None):
    """positions: (n, 3) float32 device tensor -> (n,) int32 bucket keys (uint32 bit patterns)."""
    import torch
    assert positions.is_cuda and positions.dtype == torch.float32 and positions.is_contiguous()
    n = positions.shape[0]
    keys = torch.empty(n, dtype=torch.int32, device=positions.device)
    check(load().rt_photon_keys(_ptr(positions), n, float(cell_size), bucket_count, _ptr(keys), _stream(stream)))
    return keys


class PhotonGrid:
    """Buffers of one hash grid, allocated once the way renderer.cpp:537-545 does."""

    def __init__(self, capacity, bucket_count, device="cuda", own_scratch=True):
        import torch
        self.capacity = capacity
        self.bucket_count = bucket_count
        self.counts = torch.empty(bucket_count + 1, dtype=torch.int32, device=device)
        self.offsets = torch.empty(bucket_count + 1, dtype=torch.int32, device=device)
        self.sorted = torch.empty((max(capacity, 1), PHOTON_FLOATS), dtype=torch.float32, device=device)
        nbytes = load().rt_photon_grid_scratch_bytes(capacity, bucket_count) if own_scratch else 0
        self.scratch = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device) if own_scratch else None
        self.scratch_bytes = nbytes
        self.count = 0
        self.cell_size = 0.0
        self.bitmap = None

    def mark_buckets(self, stream=None):
        """Builds the bit-per-bucket map of the grid as it stands (rt_photon_bucket_bitmap)."""
        import torch
        if self.bitmap is None:
            self.bitmap = torch.empty(max(1, self.bucket_count // 32), dtype=torch.int32, device=self.offsets.device)
        check(load().rt_photon_bucket_bitmap(_ptr(self.offsets), self.bucket_count, _ptr(self.bitmap), _stream(stream)))
        return self.bitmap

    def build(self, photons, cell_size, stream=None, sync=False):
        """photons: (n, 9) float32 device tensor."""
        import torch
        assert photons.is_cuda and photons.dtype == torch.float32 and photons.is_contiguous()
        n = photons.shape[0]
        assert n <= self.capacity
        lib = load()
        args = (_ptr(photons), n, float(cell_size), self.bucket_count, _ptr(self.counts), _ptr(self.offsets),
                _ptr(self.scratch), self.scratch_bytes, _ptr(self.sorted))
        if sync:
            check(lib.rt_build_photon_grid(*args))
        else:
            check(lib.rt_build_photon_grid_async(*args, _stream(stream)))
        self.count = n
        self.cell_size = float(cell_size)

    def build_counted(self, photons, count, cell_size, keys_ready=False, stream=None):
        """rt_build_photon_grid_counted: photons is the whole (capacity, 9) list, `count` a (1,) int32 device
        tensor with its length, which the host never reads.  keys_ready: the caller has put the bucket keys
        at keys_view() already."""
        import torch
        assert photons.is_cuda and photons.dtype == torch.float32 and photons.is_contiguous()
        assert count.is_cuda and count.dtype == torch.int32 and count.numel() == 1
        most = photons.shape[0]
        assert most <= self.capacity
        check(load().rt_build_photon_grid_counted(_ptr(photons), _ptr(count), most, float(cell_size), self.bucket_count,
                                                  _ptr(self.offsets), _ptr(self.scratch), self.scratch_bytes,
                                                  _ptr(self.sorted), int(keys_ready), _stream(stream)))
        self.count = None  # only the device knows
        self.cell_size = float(cell_size)

    def keys_view(self, most):
        """The `most` bucket keys a counted build over this grid's scratch starts from, as an int32 tensor."""
        import torch
        address = load().rt_photon_grid_keys(_ptr(self.scratch), self.scratch_bytes, most, self.bucket_count)
        assert address, "the grid's scratch is too small for that many photons"
        offset = address - self.scratch.data_ptr()
        return self.scratch[offset:offset + 4 * most].view(torch.int32)

    def build_static(self, photons, cell_size, stream=None):
        """rt_build_photon_grid_static: the build through the library's static scratch (allocates nothing)."""
        import torch
        assert photons.is_cuda and photons.dtype == torch.float32 and photons.is_contiguous()
        n = photons.shape[0]
        assert n <= self.capacity
        check(load().rt_build_photon_grid_static(_ptr(photons), n, float(cell_size), self.bucket_count,
                                                 _ptr(self.offsets), _ptr(self.sorted), _stream(stream)))
        self.count = n
        self.cell_size = float(cell_size)

    def gather(self, qpos, qnormal, radius=None, stream=None, out=None, visited=None, masked=False):
        """visited: optional (1,) int64 device tensor that receives the number of photons read;
        masked: ask the bucket bits (mark_buckets, called after the last build) before a range."""
        import torch
        assert qpos.is_cuda and qpos.dtype == torch.float32 and qpos.is_contiguous()
        assert qnormal.is_cuda and qnormal.dtype == torch.float32 and qnormal.is_contiguous()
        n = qpos.shape[0]
        flux = out if out is not None else torch.empty((n, 3), dtype=torch.float32, device=qpos.device)
        r = float(self.cell_size if radius is None else radius)
        if masked:
            assert visited is None and self.bitmap is not None
            check(load().rt_photon_gather_masked(_ptr(self.sorted), _ptr(self.offsets), _ptr(self.bitmap),
                                                 self.bucket_count, r, _ptr(qpos), _ptr(qnormal), n, _ptr(flux),
                                                 _stream(stream)))
        elif visited is None:
            check(load().rt_photon_gather(_ptr(self.sorted), _ptr(self.offsets), self.bucket_count, r, _ptr(qpos),
                                          _ptr(qnormal), n, _ptr(flux), _stream(stream)))
        else:
            check(load().rt_photon_gather_counted(_ptr(self.sorted), _ptr(self.offsets), self.bucket_count, r,
                                                  _ptr(qpos), _ptr(qnormal), n, _ptr(flux), _ptr(visited),
                                                  _stream(stream)))
        return flux


VERTEX_FLOATS = 16  # rt_gpu_vertex is 64 bytes


class Blas:
    """Per-geometry BVH (the GAS of renderer.cpp:343-396)."""

    def __init__(self, vertices, indices, stream=None):
        """vertices: (nv, 16) float32 device tensor laid out as rt_gpu_vertex, or (nv, 3);
        indices: (nt, 3) int32 device tensor."""
        import torch
        assert vertices.is_cuda and vertices.dtype == torch.float32 and vertices.is_contiguous()
        assert indices.is_cuda and indices.dtype == torch.int32 and indices.is_contiguous()
        self.vertices, self.indices = vertices, indices
        handle = C.c_void_p(0)
        check(load().rt_blas_build(_ptr(vertices), vertices.shape[0], vertices.shape[1] * 4, _ptr(indices),
                                   indices.shape[0], _stream(stream), C.byref(handle)))
        self.handle = handle

    def stats(self):
        s = RtBvhStats()
        check(load().rt_blas_stats(self.handle, C.byref(s)))
        return s

    def close(self):
        if self.handle:
            load().rt_blas_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene:
    """Top level over instances (the IAS of renderer.cpp:398-434).

    instances: list of (blas, transform) with transform a column-major 4x4
    (16 floats, scene.hpp:31-33) or None for identity; instanceId = list index,
    geometry = `geometry_ids[i]` (default: the index of the blas among distinct
    blases)."""

    def __init__(self, instances, stream=None):
        descs = (RtInstanceDesc * max(len(instances), 1))()
        self.blases = []
        for i, (blas, transform) in enumerate(instances):
            m = np.eye(4, dtype=np.float32).reshape(-1) if transform is None else np.asarray(
                transform, dtype=np.float32).reshape(16)
            # renderer.cpp:403-406: row-major 3x4 out of the column-major 4x4
            for row in range(3):
                for col in range(4):
                    descs[i].transform[row * 4 + col] = float(m[col * 4 + row])
            descs[i].instanceId = i
            if blas not in self.blases:
                self.blases.append(blas)
            descs[i].geometry = self.blases.index(blas)
            descs[i].blas = blas.handle
        handle = C.c_void_p(0)
        check(load().rt_scene_build(descs, len(instances), _stream(stream), C.byref(handle)))
        self.handle = handle
        self.instance_count = len(instances)

    def stats(self):
        s = RtBvhStats()
        check(load().rt_scene_stats(self.handle, C.byref(s)))
        return s

    def folded(self):
        """(instances, triangles) that rt_scene_build put into the tree over the instances themselves."""
        instances, triangles = C.c_uint32(0), C.c_uint32(0)
        check(load().rt_scene_folded(self.handle, C.byref(instances), C.byref(triangles)))
        return instances.value, triangles.value

    def instance_inverse(self, index):
        out = (C.c_float * 12)()
        check(load().rt_scene_instance_inverse(self.handle, index, out))
        return np.array(out, dtype=np.float32)

    def trace_closest(self, org_tmin, dir_tmax, stream=None, counters=None, hits=None, inst=None):
        """org_tmin, dir_tmax: (n, 4) float32 device tensors.  Returns (hits (n, 4) float32 whose
        last column holds the primitive's bit pattern, instance (n,) int32)."""
        import torch
        assert org_tmin.is_cuda and org_tmin.dtype == torch.float32 and org_tmin.is_contiguous()
        assert dir_tmax.is_cuda and dir_tmax.dtype == torch.float32 and dir_tmax.is_contiguous()
        n = org_tmin.shape[0]
        if hits is None:
            hits = torch.empty((n, 4), dtype=torch.float32, device=org_tmin.device)
        if inst is None:
            inst = torch.empty(n, dtype=torch.int32, device=org_tmin.device)
        if counters is None:
            check(load().rt_trace_closest(self.handle, _ptr(org_tmin), _ptr(dir_tmax), n, _ptr(hits), _ptr(inst),
                                          _stream(stream)))
        else:
            check(load().rt_trace_closest_counted(self.handle, _ptr(org_tmin), _ptr(dir_tmax), n, _ptr(hits),
                                                  _ptr(inst), _ptr(counters), _stream(stream)))
        return hits, inst

    def trace_any(self, org_tmin, dir_tmax, stream=None, out=None):
        import torch
        n = org_tmin.shape[0]
        occluded = out if out is not None else torch.empty(n, dtype=torch.uint8, device=org_tmin.device)
        check(load().rt_trace_any(self.handle, _ptr(org_tmin), _ptr(dir_tmax), n, _ptr(occluded), _stream(stream)))
        return occluded

    def close(self):
        if self.handle:
            load().rt_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


LIGHT_BYTES = 92  # sizeof(rt_gpu_light)


class LightIndex:
    """Thresholds and triangle table of a light list (rt_light_index_build): selects the light the
    reference's walk over the list selects (device.cu:433-447, 607-621) without the walk."""

    def __init__(self, lights, stream=None):
        """lights: device uint8 tensor of n * 92 bytes, the rt_gpu_light records."""
        import torch
        assert lights.is_cuda and lights.dtype == torch.uint8 and lights.is_contiguous()
        assert lights.numel() % LIGHT_BYTES == 0
        self.lights = lights
        self.count = lights.numel() // LIGHT_BYTES
        size = C.c_size_t()
        check(load().rt_light_index_bytes(self.count, C.byref(size)))
        self.memory = torch.empty((size.value + 7) // 8, dtype=torch.int64, device=lights.device)
        check(load().rt_light_index_build(_ptr(lights), self.count, _ptr(self.memory), size.value, _stream(stream)))

    def words(self):
        """The index as 32-bit words (header, thresholds, weights, table)."""
        import torch
        return self.memory.view(torch.int32)

    def thresholds(self):
        import torch
        w = self.words()
        assert int(w[2]) == 1, "this light list has no thresholds"
        return w[4:4 + self.count].view(torch.float32)


def light_query(lights, light_weight, index=None, uniform=None, instance=None, primitive=None, stream=None):
    """rt_light_query: (chosen, found) int32 tensors (None where the inputs were not given); through
    `index` (a LightIndex) or by the walks themselves."""
    import torch
    count = lights.numel() // LIGHT_BYTES
    n = uniform.numel() if uniform is not None else instance.numel()
    chosen = torch.empty(n, dtype=torch.int32, device=lights.device) if uniform is not None else None
    found = torch.empty(n, dtype=torch.int32, device=lights.device) if instance is not None else None
    check(load().rt_light_query(_ptr(lights), count, float(light_weight), _ptr(index.memory) if index else None,
                                _ptr(uniform) if uniform is not None else None,
                                _ptr(instance) if instance is not None else None,
                                _ptr(primitive) if primitive is not None else None, n,
                                _ptr(chosen) if chosen is not None else None,
                                _ptr(found) if found is not None else None, _stream(stream)))
    return chosen, found


SHARED_MATH = {"sine": 0, "cosine": 1, "power": 2, "arc_tangent2": 3, "arc_cosine": 4, "exp2": 5}


def shared_math(name, x, y=None, stream=None):
    """include/rt_math.h on the device: x (and y for power / arc_tangent2) float32 device tensors -> float32."""
    import torch
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty_like(x)
    check(load().rt_shared_math(SHARED_MATH[name], _ptr(x), _ptr(y), x.numel(), _ptr(out), _stream(stream)))
    return out
