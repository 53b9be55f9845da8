"""Python mirror of the host layer: rt::Renderer (raytracer_amd/host/renderer.hpp) bound through
librt_host.so.  Same methods as the reference's src/renderer.hpp:16-37, built from the scene dicts of
raytracer_amd.workloads.  No CPU fallback: loading fails loudly when the libraries are missing."""
import ctypes as C
import os

import numpy as np

from . import engine

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "librt_host.so")

PREVIEW, FINAL = 0, 1
MAX_PHOTON_PASSES = 16  # RT_MAX_PHOTON_PASSES: photon passes the renderer may trace together


class Float2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class Float3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Float4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class GpuTextureRef(C.Structure):
    _fields_ = [("texture", C.c_int32), ("texCoord", C.c_uint32), ("offset", Float2), ("scale", Float2),
                ("rotation", C.c_float), ("strength", C.c_float)]


class GpuMaterial(C.Structure):
    _fields_ = [("baseColor", Float4), ("emissive", Float3), ("attenuationColor", Float3), ("metallic", C.c_float),
                ("roughness", C.c_float), ("transmission", C.c_float), ("ior", C.c_float), ("thickness", C.c_float),
                ("attenuationDistance", C.c_float), ("emissiveStrength", C.c_float), ("dispersion", C.c_float),
                ("baseColorTexture", GpuTextureRef), ("metallicRoughnessTexture", GpuTextureRef),
                ("normalTexture", GpuTextureRef), ("emissiveTexture", GpuTextureRef),
                ("transmissionTexture", GpuTextureRef), ("thicknessTexture", GpuTextureRef),
                ("_tail", C.c_uint8 * 8)]  # the C struct is 16-byte aligned (float4 member)


class GpuTexture(C.Structure):
    _fields_ = [("pixels", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("wrapU", C.c_uint32),
                ("wrapV", C.c_uint32)]


class GpuLight(C.Structure):
    _fields_ = [("a", Float3), ("b", Float3), ("c", Float3), ("normal", Float3), ("emission", Float3),
                ("area", C.c_float), ("weight", C.c_float), ("range", C.c_float), ("innerCone", C.c_float),
                ("outerCone", C.c_float), ("instance", C.c_uint32), ("primitive", C.c_uint32), ("type", C.c_uint32)]


class LaunchParams(C.Structure):
    """rt_launch_params (include/rt_types.h), 344 bytes."""
    _fields_ = ([(n, C.c_void_p) for n in (
        "output", "accumulation", "diffuse", "reflection", "refraction", "caustics", "albedoGuide", "normalGuide",
        "display", "display2", "display3", "composite", "materials", "textures", "lights", "environment",
        "environmentCdf", "photons", "photonCount", "sortedPhotons", "photonCellStart")] +
        [("scene", C.c_uint64)] +
        [(n, Float3) for n in ("eye", "cameraU", "cameraV", "cameraW", "lensU", "lensV")] +
        [(n, C.c_float) for n in ("aperture", "focusDistance", "lightWeight", "environmentWeight",
                                  "environmentRotation", "environmentStrength", "exposure")] +
        [("sceneCenter", Float3), ("sceneRadius", C.c_float), ("photonRadius", C.c_float)] +
        [(n, C.c_uint32) for n in ("lightCount", "environmentWidth", "environmentHeight", "width", "height", "sample",
                                   "maxDepth", "photonBucketCount", "photonCapacity", "photonEmissions", "photonPass")])


assert C.sizeof(GpuMaterial) == 272 and C.sizeof(GpuLight) == 92 and C.sizeof(LaunchParams) == 344
assert C.sizeof(GpuTexture) == 24

_lib = None
_vp, _u32, _f32 = C.c_void_p, C.c_uint32, C.c_float
_SIG = {
    "rth_last_error": (C.c_char_p, []),
    "rth_scene_create": (_vp, []),
    "rth_scene_destroy": (None, [_vp]),
    "rth_scene_add_material": (C.c_int, [_vp, C.POINTER(GpuMaterial)]),
    "rth_scene_add_geometry": (C.c_int, [_vp, _vp, _u32, _vp, _u32, _u32]),
    "rth_scene_add_instance": (C.c_int, [_vp, _u32, _vp]),
    "rth_scene_add_camera": (C.c_int, [_vp, _vp, _f32, _f32, _f32, _f32, _f32]),
    "rth_scene_add_light": (C.c_int, [_vp, _vp, _vp, _f32, _f32, _f32, _f32, _u32]),
    "rth_scene_add_image": (C.c_int, [_vp, _vp, _u32, _u32]),
    "rth_scene_add_texture": (C.c_int, [_vp, C.c_int32, _u32, _u32]),
    "rth_scene_set_environment": (None, [_vp, _vp, _u32, _u32, _f32, _f32, _f32]),
    "rth_scene_load": (_vp, [C.c_char_p]),
    "rth_scene_counts": (None, [_vp, C.POINTER(_u32)]),
    "rth_scene_get_geometry": (C.c_int, [_vp, _u32, C.POINTER(_u32), _vp, _vp, C.c_char_p, _u32]),
    "rth_scene_get_instance": (C.c_int, [_vp, _u32, _vp, C.POINTER(_u32)]),
    "rth_scene_get_material": (C.c_int, [_vp, _u32, _vp, C.POINTER(_f32), C.POINTER(_u32)]),
    "rth_scene_get_camera": (C.c_int, [_vp, _u32, _vp, _vp]),
    "rth_scene_get_light": (C.c_int, [_vp, _u32, _vp, _vp]),
    "rth_scene_get_image": (C.c_int, [_vp, _u32, C.POINTER(_u32), _vp]),
    "rth_scene_get_texture": (C.c_int, [_vp, _u32, _vp]),
    "rth_scene_get_environment": (C.c_int, [_vp, _vp, _vp]),
    "rth_decode_png": (C.c_int, [_vp, C.c_size_t, C.POINTER(_u32), C.POINTER(_u32), _vp]),
    "rth_decode_jpeg": (C.c_int, [_vp, C.c_size_t, C.POINTER(_u32), C.POINTER(_u32), _vp]),
    "rth_decode_hdr": (C.c_int, [_vp, C.c_size_t, C.POINTER(_u32), C.POINTER(_u32), _vp]),
    "rth_write_png": (C.c_int, [C.c_char_p, _u32, _u32, _vp]),
    "rth_write_exr": (C.c_int, [C.c_char_p, _u32, _u32, _vp]),
    "rth_renderer_create": (_vp, [_vp, _u32, _u32, C.c_int]),
    "rth_renderer_destroy": (None, [_vp]),
    "rth_renderer_render": (C.c_int, [_vp, _vp]),
    "rth_renderer_denoise": (C.c_int, [_vp]),
    "rth_renderer_set_camera": (C.c_int, [_vp, _vp, _f32, _f32, _f32, _f32, _f32]),
    "rth_renderer_set_profile": (C.c_int, [_vp, C.c_int]),
    "rth_renderer_copy_output": (C.c_int, [_vp, _vp]),
    "rth_renderer_pixels": (C.c_int, [_vp, _vp]),
    "rth_renderer_linear_pixels": (C.c_int, [_vp, _vp]),
    "rth_renderer_samples": (_u32, [_vp]),
    "rth_renderer_set_photon_budget": (C.c_int, [_vp, _u32, _u32, _u32]),
    "rth_renderer_set_sample_sequence": (C.c_int, [_vp, _u32, _u32]),
    "rth_renderer_set_pixel_range": (C.c_int, [_vp, _u32, _u32]),
    "rth_renderer_set_row_interleave": (C.c_int, [_vp, _u32, _u32]),
    "rth_renderer_set_concurrent_tiles": (C.c_int, [_vp, _u32]),
    "rth_renderer_concurrent_tiles": (_u32, [_vp]),
    "rth_renderer_set_parts_and_depth": (C.c_int, [_vp, _u32, _u32]),
    "rth_renderer_samples_in_flight": (_u32, [_vp]),
    "rth_renderer_set_sample_batch": (C.c_int, [_vp, _u32]),
    "rth_renderer_sample_batch": (_u32, [_vp]),
    "rth_renderer_set_ray_share": (C.c_int, [_vp, _u32, _u32]),
    "rth_renderer_set_manual_photon_passes": (None, [_vp, C.c_int]),
    "rth_renderer_photon_pass_due": (C.c_int, [_vp]),
    "rth_renderer_begin_photon_pass": (C.c_int, [_vp]),
    "rth_renderer_trace_photon_range": (C.c_int, [_vp, _u32, _u32, C.c_int, C.POINTER(_u32)]),
    "rth_renderer_photon_passes_together": (_u32, [_vp]),
    "rth_renderer_begin_photon_passes": (C.c_int, [_vp, _u32, C.POINTER(_u32)]),
    "rth_renderer_trace_photon_range_passes": (C.c_int, [_vp, _u32, _u32, C.c_int, C.POINTER(_u32)]),
    "rth_renderer_finish_photon_passes": (C.c_int, [_vp, C.POINTER(_u32)]),
    "rth_renderer_photon_buffer_of": (_vp, [_vp, _u32]),
    "rth_renderer_photon_staging_buffer_of": (_vp, [_vp, _u32]),
    "rth_renderer_photon_buffer": (_vp, [_vp]),
    "rth_renderer_photon_staging_buffer": (_vp, [_vp]),
    "rth_renderer_output_buffer": (_vp, [_vp]),
    "rth_renderer_photon_capacity": (_u32, [_vp]),
    "rth_renderer_photon_emissions": (_u32, [_vp]),
    "rth_renderer_finish_photon_pass": (C.c_int, [_vp, _u32]),
    "rth_renderer_accumulation_buffer": (_vp, [_vp, C.c_int]),
    "rth_renderer_statistics": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "rth_renderer_set_measurement": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int]),
    "rth_renderer_stage_timings": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    "rth_renderer_synchronize": (C.c_int, [_vp]),
    "rth_renderer_launch_params": (None, [_vp, C.POINTER(LaunchParams)]),
    "rth_renderer_caustic_radius": (_f32, [_vp]),
}


def load():
    global _lib
    if _lib is not None:
        return _lib
    engine.load()
    if not os.path.exists(HOST_LIB_PATH):
        raise engine.EngineError(f"{HOST_LIB_PATH} is missing: build it with `make -C raytracer_amd/host`")
    lib = C.CDLL(HOST_LIB_PATH)
    for name, (restype, argtypes) in _SIG.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _check(status):
    if status != 0:
        raise engine.EngineError(load().rth_last_error().decode())


def _added(index):
    """rth_scene_add_* return the index of the new entry, or -1 when the entry could not be stored."""
    if index < 0:
        raise engine.EngineError(load().rth_last_error().decode())
    return index


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def gpu_material(m):
    """dict (workloads.default_material) -> GpuMaterial."""
    g = GpuMaterial()
    g.baseColor = Float4(*m["baseColor"])
    g.emissive = Float3(*m["emissive"])
    g.attenuationColor = Float3(*m["attenuationColor"])
    for k in ("metallic", "roughness", "transmission", "ior", "thickness", "attenuationDistance", "emissiveStrength",
              "dispersion"):
        setattr(g, k, m[k])
    for k in ("baseColorTexture", "metallicRoughnessTexture", "normalTexture", "emissiveTexture",
              "transmissionTexture", "thicknessTexture"):
        t = m.get(k)
        ref = GpuTextureRef(-1, 0, Float2(0.0, 0.0), Float2(1.0, 1.0), 0.0, 1.0)
        if t is not None:
            ref = GpuTextureRef(t.get("texture", -1), t.get("texCoord", 0), Float2(*t.get("offset", (0.0, 0.0))),
                                Float2(*t.get("scale", (1.0, 1.0))), t.get("rotation", 0.0), t.get("strength", 1.0))
        setattr(g, k, ref)
    return g


class HostScene:
    """rt::Scene built from a workloads scene dict."""

    def __init__(self, scene):
        lib = load()
        self.handle = C.c_void_p(lib.rth_scene_create())
        for m in scene["materials"]:
            _added(lib.rth_scene_add_material(self.handle, C.byref(gpu_material(m))))
        for g in scene["geometries"]:
            v = np.ascontiguousarray(g["vertices"], dtype=np.float32)
            i = np.ascontiguousarray(g["indices"], dtype=np.uint32)
            _added(lib.rth_scene_add_geometry(self.handle, _p(v), v.shape[0], _p(i), i.shape[0], g["material"]))
        for gi, m in scene["instances"]:
            mm = np.ascontiguousarray(m, dtype=np.float32)
            _added(lib.rth_scene_add_instance(self.handle, gi, _p(mm)))
        for c in scene["cameras"]:
            mm = np.ascontiguousarray(c["transform"], dtype=np.float32)
            _added(lib.rth_scene_add_camera(self.handle, _p(mm), c["verticalFov"], c["aspectRatio"], c["nearPlane"],
                                            c["aperture"], c["focusDistance"]))
        for l in scene["lights"]:
            mm = np.ascontiguousarray(l["transform"], dtype=np.float32)
            col = np.ascontiguousarray(l["color"], dtype=np.float32)
            _added(lib.rth_scene_add_light(self.handle, _p(mm), _p(col), l["intensity"], l["range"], l["innerCone"],
                                           l["outerCone"], l["type"]))
        for img in scene.get("images", []):
            px = np.ascontiguousarray(img, dtype=np.uint8)
            _added(lib.rth_scene_add_image(self.handle, _p(px), px.shape[1], px.shape[0]))
        for t in scene.get("textures", []):
            _added(lib.rth_scene_add_texture(self.handle, t["image"], t.get("wrapU", 10497), t.get("wrapV", 10497)))
        env = scene.get("environment")
        if env is not None:
            px = np.ascontiguousarray(env["pixels"], dtype=np.float32)
            lib.rth_scene_set_environment(self.handle, _p(px), px.shape[1], px.shape[0], env.get("rotation", 0.0),
                                          env.get("strength", 1.0), env.get("exposure", 0.0))

    def close(self):
        if self.handle:
            load().rth_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LoadedScene(HostScene):
    """rt::Scene read from a .gltf / .glb file by rt::loadScene, with read access to what it holds."""

    def __init__(self, path):
        lib = load()
        handle = lib.rth_scene_load(os.fsencode(str(path)))
        if not handle:
            raise engine.EngineError(lib.rth_last_error().decode())
        self.handle = C.c_void_p(handle)

    def counts(self):
        out = (_u32 * 8)()
        load().rth_scene_counts(self.handle, out)
        keys = ("geometries", "instances", "materials", "images", "textures", "cameras", "lights", "environment_pixels")
        return dict(zip(keys, (int(v) for v in out)))

    def geometry(self, index):
        lib = load()
        sizes = (_u32 * 3)()
        name = C.create_string_buffer(256)
        _check(lib.rth_scene_get_geometry(self.handle, index, sizes, None, None, name, 256))
        vertices = np.empty((sizes[0], 14), dtype=np.float32)
        indices = np.empty(sizes[1], dtype=np.uint32)
        _check(lib.rth_scene_get_geometry(self.handle, index, sizes, _p(vertices), _p(indices), name, 256))
        return dict(name=name.value.decode(), vertices=vertices, indices=indices.reshape(-1, 3), material=int(sizes[2]))

    def instance(self, index):
        m = np.empty(16, dtype=np.float32)
        g = _u32(0)
        _check(load().rth_scene_get_instance(self.handle, index, _p(m), C.byref(g)))
        return int(g.value), m

    def material(self, index):
        m = GpuMaterial()
        cutoff, flags = _f32(0), _u32(0)
        _check(load().rth_scene_get_material(self.handle, index, C.byref(m), C.byref(cutoff), C.byref(flags)))
        return m, float(cutoff.value), int(flags.value)

    def camera(self, index):
        m, v = np.empty(16, dtype=np.float32), np.empty(8, dtype=np.float32)
        _check(load().rth_scene_get_camera(self.handle, index, _p(m), _p(v)))
        return dict(transform=m, verticalFov=float(v[0]), aspectRatio=float(v[1]), nearPlane=float(v[2]),
                    farPlane=float(v[3]), aperture=float(v[4]), focusDistance=float(v[5]))

    def light(self, index):
        m, v = np.empty(16, dtype=np.float32), np.empty(8, dtype=np.float32)
        _check(load().rth_scene_get_light(self.handle, index, _p(m), _p(v)))
        return dict(transform=m, color=v[:3].copy(), intensity=float(v[3]), range=float(v[4]), innerCone=float(v[5]),
                    outerCone=float(v[6]), type=int(v[7]))

    def image(self, index):
        sizes = (_u32 * 4)()
        _check(load().rth_scene_get_image(self.handle, index, sizes, None))
        rgba = np.empty((sizes[1], sizes[0], 4), dtype=np.uint8)
        _check(load().rth_scene_get_image(self.handle, index, sizes, _p(rgba)))
        return dict(width=int(sizes[0]), height=int(sizes[1]), mimeType=int(sizes[2]), encoded_bytes=int(sizes[3]), rgba=rgba)

    def texture(self, index):
        v = np.empty(6, dtype=np.int32)
        _check(load().rth_scene_get_texture(self.handle, index, _p(v)))
        return dict(image=int(v[0]), magFilter=int(v[1]), minFilter=int(v[2]), wrapU=int(v[3]), wrapV=int(v[4]))

    def environment(self):
        v = np.empty(5, dtype=np.float32)
        _check(load().rth_scene_get_environment(self.handle, _p(v), None))
        pixels = np.empty((int(v[1]), int(v[0]), 4), dtype=np.float32)
        _check(load().rth_scene_get_environment(self.handle, _p(v), _p(pixels)))
        return dict(width=int(v[0]), height=int(v[1]), rotation=float(v[2]), strength=float(v[3]),
                    exposure=float(v[4]), pixels=pixels)


def load_scene(path):
    """rt::loadScene: a .gltf / .glb file as a scene a Renderer accepts."""
    return LoadedScene(path)


def decode_png(data):
    """The loader's PNG reader on a bytes object: (height, width, 4) uint8."""
    lib = load()
    buf = np.frombuffer(data, dtype=np.uint8)
    w, h = _u32(0), _u32(0)
    _check(lib.rth_decode_png(_p(buf), buf.size, C.byref(w), C.byref(h), None))
    out = np.empty((h.value, w.value, 4), dtype=np.uint8)
    _check(lib.rth_decode_png(_p(buf), buf.size, C.byref(w), C.byref(h), _p(out)))
    return out


def decode_jpeg(data):
    """The loader's JPEG reader on a bytes object: (height, width, 4) uint8."""
    lib = load()
    buf = np.frombuffer(data, dtype=np.uint8)
    w, h = _u32(0), _u32(0)
    _check(lib.rth_decode_jpeg(_p(buf), buf.size, C.byref(w), C.byref(h), None))
    out = np.empty((h.value, w.value, 4), dtype=np.uint8)
    _check(lib.rth_decode_jpeg(_p(buf), buf.size, C.byref(w), C.byref(h), _p(out)))
    return out


def decode_hdr(data):
    """The loader's Radiance .hdr reader on a bytes object: (height, width, 4) float32."""
    lib = load()
    buf = np.frombuffer(data, dtype=np.uint8)
    w, h = _u32(0), _u32(0)
    _check(lib.rth_decode_hdr(_p(buf), buf.size, C.byref(w), C.byref(h), None))
    out = np.empty((h.value, w.value, 4), dtype=np.float32)
    _check(lib.rth_decode_hdr(_p(buf), buf.size, C.byref(w), C.byref(h), _p(out)))
    return out


def write_png(path, rgba):
    """rt::writePng: (height, width, 4) uint8, row 0 = bottom row."""
    a = np.ascontiguousarray(rgba, dtype=np.uint8)
    _check(load().rth_write_png(os.fsencode(str(path)), a.shape[1], a.shape[0], _p(a)))


def write_exr(path, rgba):
    """rt::writeExr: (height, width, 4) float32, row 0 = bottom row."""
    a = np.ascontiguousarray(rgba, dtype=np.float32)
    _check(load().rth_write_exr(os.fsencode(str(path)), a.shape[1], a.shape[0], _p(a)))


CLI_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "raytracer")


class Renderer:
    """rt::Renderer: Renderer(scene, width, height, profile); render(); pixels(); linear_pixels(); samples()."""

    def __init__(self, scene, width, height, profile=PREVIEW):
        lib = load()
        self._scene = scene if isinstance(scene, HostScene) else HostScene(scene)
        self.width, self.height = width, height
        handle = lib.rth_renderer_create(self._scene.handle, width, height, profile)
        if not handle:
            raise engine.EngineError(lib.rth_last_error().decode())
        self.handle = C.c_void_p(handle)

    def render(self, output=None):
        _check(load().rth_renderer_render(self.handle, C.c_void_p(output.data_ptr()) if output is not None else None))

    def denoise(self):
        _check(load().rth_renderer_denoise(self.handle))

    def set_camera(self, c):
        mm = np.ascontiguousarray(c["transform"], dtype=np.float32)
        _check(load().rth_renderer_set_camera(self.handle, _p(mm), c["verticalFov"], c["aspectRatio"], c["nearPlane"],
                                              c["aperture"], c["focusDistance"]))

    def set_profile(self, profile):
        _check(load().rth_renderer_set_profile(self.handle, profile))

    def copy_output(self, tensor):
        _check(load().rth_renderer_copy_output(self.handle, C.c_void_p(tensor.data_ptr())))

    def pixels(self):
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        _check(load().rth_renderer_pixels(self.handle, _p(out)))
        return out

    def linear_pixels(self):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        _check(load().rth_renderer_linear_pixels(self.handle, _p(out)))
        return out

    def samples(self):
        return int(load().rth_renderer_samples(self.handle))

    # ---- hooks beyond the reference interface ----
    def set_photon_budget(self, emissions, capacity, buckets):
        _check(load().rth_renderer_set_photon_budget(self.handle, emissions, capacity, buckets))

    def set_sample_sequence(self, first, stride):
        _check(load().rth_renderer_set_sample_sequence(self.handle, first, stride))

    def set_pixel_range(self, begin, end):
        _check(load().rth_renderer_set_pixel_range(self.handle, begin, end))

    def set_row_interleave(self, first, stride):
        """render() covers every stride-th band of eight pixel rows from band `first` on (replaces a pixel range)."""
        _check(load().rth_renderer_set_row_interleave(self.handle, first, stride))

    def set_concurrent_tiles(self, count, samples_in_flight=1):
        """render() cuts its pixels into `count` regions that run side by side; with samples_in_flight > 1
        that many consecutive samples of a region may run at once.  Restarts the accumulation."""
        _check(load().rth_renderer_set_parts_and_depth(self.handle, count, samples_in_flight))

    def set_sample_batch(self, samples):
        """render() calls are collected into passes of up to `samples` samples that the GPU takes in one go
        (same image; for a GPU that renders a small share of a frame).  Restarts the accumulation."""
        _check(load().rth_renderer_set_sample_batch(self.handle, samples))

    def sample_batch(self):
        return int(load().rth_renderer_sample_batch(self.handle))

    def samples_in_flight(self):
        return int(load().rth_renderer_samples_in_flight(self.handle))

    def set_ray_share(self, numerator, denominator):
        """Share of the GPU's wave slots a ray launch of one part of the frame takes (reset by set_concurrent_tiles)."""
        _check(load().rth_renderer_set_ray_share(self.handle, numerator, denominator))

    def concurrent_tiles(self):
        return int(load().rth_renderer_concurrent_tiles(self.handle))

    def set_manual_photon_passes(self, manual=True):
        load().rth_renderer_set_manual_photon_passes(self.handle, int(manual))

    def photon_pass_due(self):
        return bool(load().rth_renderer_photon_pass_due(self.handle))

    def begin_photon_pass(self):
        _check(load().rth_renderer_begin_photon_pass(self.handle))

    def trace_photon_range(self, first, count, staging=False):
        """Photons stored so far in this pass; with staging they go to photon_staging_buffer()."""
        stored = C.c_uint32(0)
        _check(load().rth_renderer_trace_photon_range(self.handle, first, count, int(staging), C.byref(stored)))
        return stored.value

    def photon_buffer(self, which=0):
        return int(load().rth_renderer_photon_buffer_of(self.handle, which) or 0)

    def photon_staging_buffer(self, which=0):
        return int(load().rth_renderer_photon_staging_buffer_of(self.handle, which) or 0)

    def photon_passes_together(self):
        """Consecutive photon passes begin_photon_passes() can start at once (those of the maps one camera pass
        gathers from)."""
        return int(load().rth_renderer_photon_passes_together(self.handle))

    def begin_photon_passes(self, wanted):
        """Starts up to `wanted` consecutive passes; returns how many were started (at least one)."""
        begun = _u32(0)
        _check(load().rth_renderer_begin_photon_passes(self.handle, wanted, C.byref(begun)))
        return int(begun.value)

    def trace_photon_range_passes(self, first, count, passes, staging=False):
        """The emission range for every begun pass; returns the stored count per pass."""
        stored = (_u32 * MAX_PHOTON_PASSES)()
        _check(load().rth_renderer_trace_photon_range_passes(self.handle, first, count, int(staging), stored))
        return [int(stored[k]) for k in range(passes)]

    def finish_photon_passes(self, totals):
        values = (_u32 * MAX_PHOTON_PASSES)(*totals)
        _check(load().rth_renderer_finish_photon_passes(self.handle, values))

    def output_buffer(self):
        return int(load().rth_renderer_output_buffer(self.handle) or 0)

    def photon_capacity(self):
        return int(load().rth_renderer_photon_capacity(self.handle))

    def photon_emissions(self):
        return int(load().rth_renderer_photon_emissions(self.handle))

    def finish_photon_pass(self, total):
        _check(load().rth_renderer_finish_photon_pass(self.handle, total))

    def accumulation_buffer(self, index=0):
        return int(load().rth_renderer_accumulation_buffer(self.handle, index) or 0)

    def statistics(self):
        out = (C.c_uint64 * 8)()
        _check(load().rth_renderer_statistics(self.handle, out))
        return dict(extend_rays=out[0], shadow_rays=out[1], gather_queries=out[2], gathered_photons=out[3],
                    nodes=out[4], triangles=out[5], instances=out[6], refused_references=out[7])

    def set_measurement(self, time_stages=False, count_traversal=False, gather_per_lane=False):
        _check(load().rth_renderer_set_measurement(self.handle, int(time_stages), int(count_traversal),
                                                   int(gather_per_lane)))

    def stage_timings(self):
        """{stage: (milliseconds, launches)} of the event-timed launches since the last call."""
        ms = (C.c_float * 5)()
        n = (C.c_uint32 * 5)()
        _check(load().rth_renderer_stage_timings(self.handle, ms, n))
        names = ("generate", "trace", "shade", "advance", "accumulate")
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(names)}

    def synchronize(self):
        _check(load().rth_renderer_synchronize(self.handle))

    def launch_params(self):
        p = LaunchParams()
        load().rth_renderer_launch_params(self.handle, C.byref(p))
        return p

    def caustic_radius(self):
        return float(load().rth_renderer_caustic_radius(self.handle))

    def close(self):
        if self.handle:
            load().rth_renderer_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _DeviceMemory:
    """A raw device address range as an object torch.as_tensor() accepts (the CUDA array interface,
    which the ROCm build of torch reads the same way)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 3, "strides": None}


def device_view(ptr, nbytes, dtype=None):
    """A torch tensor over nbytes of device memory at `ptr` on the current device, without a copy: the
    renderer's photon list and accumulation buffers as tensors that torch.distributed can send from and
    receive into.  The memory stays the renderer's; the view must not outlive it."""
    import torch
    if nbytes == 0:
        flat = torch.empty(0, dtype=torch.uint8, device="cuda")
    else:
        flat = torch.as_tensor(_DeviceMemory(ptr, nbytes), device="cuda")
    return flat if dtype is None else flat.view(dtype)


def device_to_host(ptr, nbytes):
    """Copies nbytes from a raw device address into a numpy byte array."""
    import torch
    torch.cuda.synchronize()
    return device_view(ptr, nbytes).cpu().numpy()
