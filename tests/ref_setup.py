"""numpy restatement of the reference's host-side set-up (src/renderer.cpp) for the oracle and for
checking the product's host layer: material table (:121-143, 256-262), light table (:436-507),
environment CDF (:301-341), scene bounds and photon constants (:509-535), camera basis and caustic
radius (:672-700).  Produces a LaunchParams whose pointers are host addresses."""
import ctypes as C
import math

import numpy as np

from raytracer_amd import renderer as R

f32 = np.float32


def _point(m, p):
    """renderer.cpp:145-151 with a column-major 4x4."""
    m = m.astype(f32)
    return np.array([m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12],
                     m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13],
                     m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14]], dtype=f32)


def _luminance(v):
    return f32(f32(f32(0.2126) * v[0] + f32(0.7152) * v[1]) + f32(0.0722) * v[2])


def _normalized(x, y, z):
    inv = f32(1.0) / np.sqrt(f32(f32(x * x + y * y) + z * z))
    return np.array([x * inv, y * inv, z * inv], dtype=f32)


TEXTURE_SLOTS = ("baseColorTexture", "metallicRoughnessTexture", "normalTexture", "emissiveTexture",
                 "transmissionTexture", "thicknessTexture")


def material_record(m):
    """renderer.cpp:108-143 restated: a scene material (dict with the fields of scene.hpp:52-73) as the
    272-byte device record; a slot without a texture is index -1, scale (1, 1), strength 1 (scene.hpp:43-50)."""
    g = R.GpuMaterial()
    g.baseColor = R.Float4(*[f32(v) for v in m["baseColor"]])
    g.emissive = R.Float3(*[f32(v) for v in m["emissive"]])
    g.attenuationColor = R.Float3(*[f32(v) for v in m["attenuationColor"]])
    g.metallic, g.roughness, g.transmission, g.ior = m["metallic"], m["roughness"], m["transmission"], m["ior"]
    g.thickness, g.attenuationDistance = m["thickness"], m["attenuationDistance"]
    g.emissiveStrength, g.dispersion = m["emissiveStrength"], m["dispersion"]
    for slot in TEXTURE_SLOTS:
        t = m.get(slot) or {}
        if t.get("texture", -1) >= 0 and t.get("texCoord", 0) > 1:
            raise ValueError("Only glTF TEXCOORD_0 and TEXCOORD_1 are supported")
        ref = getattr(g, slot)
        ref.texture, ref.texCoord = t.get("texture", -1), t.get("texCoord", 0)
        ref.offset = R.Float2(*t.get("offset", (0.0, 0.0)))
        ref.scale = R.Float2(*t.get("scale", (1.0, 1.0)))
        ref.rotation, ref.strength = t.get("rotation", 0.0), t.get("strength", 1.0)
    return g


class Setup:
    def __init__(self, scene, width, height, final=False, photon_budget=None):
        self.keep = []
        p = R.LaunchParams()
        p.width, p.height = width, height
        p.maxDepth = 32 if final else 5
        # materials
        mats = (R.GpuMaterial * len(scene["materials"]))(*[material_record(m) for m in scene["materials"]])
        self.materials = mats
        self.keep.append(mats)
        p.materials = C.addressof(mats)
        # textures
        texs = scene.get("textures", [])
        if texs:
            recs = (R.GpuTexture * len(texs))()
            for r, t in zip(recs, texs):
                img = np.ascontiguousarray(scene["images"][t["image"]], dtype=np.uint8)
                self.keep.append(img)
                r.pixels, r.width, r.height = img.ctypes.data, img.shape[1], img.shape[0]
                r.wrapU, r.wrapV = t.get("wrapU", 10497), t.get("wrapV", 10497)
            self.keep.append(recs)
            p.textures = C.addressof(recs)
        # environment, renderer.cpp:301-341
        env = scene.get("environment")
        self.environment_power = f32(0.0)
        if env is not None:
            px = np.ascontiguousarray(env["pixels"], dtype=f32)
            h, w = px.shape[:2]
            p.environmentRotation, p.environmentStrength = env.get("rotation", 0.0), env.get("strength", 1.0)
            p.exposure = env.get("exposure", 0.0)
            total = 0.0
            cdf = np.zeros(h * w, dtype=f32)
            for y in range(h):
                sine = f32(math.sin(float(f32(f32(3.14159265) * f32(f32(y) + f32(0.5)) / f32(h)))))
                lum = (f32(0.2126) * px[y, :, 0] + f32(0.7152) * px[y, :, 1]).astype(f32) + f32(0.0722) * px[y, :, 2]
                for x in range(w):
                    total += float(f32(f32(lum[x]) * sine))
                    cdf[y * w + x] = f32(total)
            if total > 0.0:
                cdf = (cdf.astype(np.float64) / total).astype(f32)
                self.keep += [px, cdf]
                self.environment_cdf = cdf
                p.environment, p.environmentCdf = px.ctypes.data, cdf.ctypes.data
                p.environmentWidth, p.environmentHeight = w, h
                self.environment_power = f32(total * 19.7392088 / float(w * h) * float(f32(env.get("strength", 1.0))))
                p.environmentWeight = self.environment_power
        # lights, renderer.cpp:436-507
        lights = []
        for ii, (gi, m) in enumerate(scene["instances"]):
            g = scene["geometries"][gi]
            mat = scene["materials"][g["material"]]
            emission = np.array(mat["emissive"], dtype=f32) * f32(mat["emissiveStrength"])
            if _luminance(emission) <= 0:
                continue
            m = np.asarray(m, dtype=f32)
            for t, (i0, i1, i2) in enumerate(g["indices"]):
                a, b, c = (_point(m, g["vertices"][i, :3]) for i in (i0, i1, i2))
                e1, e2 = (b - a).astype(f32), (c - a).astype(f32)
                av = np.array([e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2],
                               e1[0] * e2[1] - e1[1] * e2[0]], dtype=f32)
                twice = np.sqrt(f32(f32(av[0] * av[0] + av[1] * av[1]) + av[2] * av[2]))
                if twice <= f32(1e-8):
                    continue
                area = f32(0.5) * twice
                L = R.GpuLight()
                L.a, L.b, L.c = R.Float3(*a), R.Float3(*b), R.Float3(*c)
                L.normal = R.Float3(*(av / twice).astype(f32))
                L.emission = R.Float3(*emission)
                L.area, L.weight = area, f32(area * _luminance(emission))
                L.instance, L.primitive, L.type = ii, t, 3
                lights.append(L)
        for l in scene["lights"]:
            m = np.asarray(l["transform"], dtype=f32)
            emission = (np.array(l["color"], dtype=f32) * f32(l["intensity"])).astype(f32)
            solid = (f32(6.2831853) * f32(f32(1.0) - f32(math.cos(l["outerCone"]))) if l["type"] == 1
                     else f32(12.566371) if l["type"] == 2 else f32(1.0))
            L = R.GpuLight()
            L.a = R.Float3(m[12], m[13], m[14])
            L.b = R.Float3(*_normalized(-m[8], -m[9], -m[10]))
            L.emission = R.Float3(*emission)
            L.weight = f32(_luminance(emission) * solid)
            L.range, L.innerCone, L.outerCone = l["range"], l["innerCone"], l["outerCone"]
            L.instance = L.primitive = 0xffffffff
            L.type = l["type"]
            lights.append(L)
        if self.environment_power > 0:
            L = R.GpuLight()
            L.weight = self.environment_power
            L.instance = L.primitive = 0xffffffff
            L.type = 4
            lights.append(L)
        weight = f32(0.0)
        for L in lights:
            weight = f32(weight + f32(L.weight))
        p.lightWeight, p.lightCount = weight, len(lights)
        if lights:
            arr = (R.GpuLight * len(lights))(*lights)
            self.keep.append(arr)
            p.lights = C.addressof(arr)
        self.lights = lights
        # scene bounds, renderer.cpp:509-528
        lo = np.full(3, np.finfo(f32).max, dtype=f32)
        hi = -lo
        for gi, m in scene["instances"]:
            m = np.asarray(m, dtype=f32)
            v = scene["geometries"][gi]["vertices"][:, :3].astype(f32)
            w = np.stack([(m[0] * v[:, 0] + m[4] * v[:, 1]).astype(f32) + m[8] * v[:, 2] + m[12],
                          (m[1] * v[:, 0] + m[5] * v[:, 1]).astype(f32) + m[9] * v[:, 2] + m[13],
                          (m[2] * v[:, 0] + m[6] * v[:, 1]).astype(f32) + m[10] * v[:, 2] + m[14]], axis=1).astype(f32)
            lo, hi = np.minimum(lo, w.min(0)), np.maximum(hi, w.max(0))
        center = ((lo + hi) * f32(0.5)).astype(f32)
        d = (hi - lo).astype(f32)
        p.sceneCenter = R.Float3(*center)
        p.sceneRadius = f32(np.sqrt(f32(f32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])) * f32(0.525))
        emissions, capacity, buckets = photon_budget or (1 << 24, 1 << 23, 1 << 22)
        p.photonEmissions, p.photonCapacity, p.photonBucketCount = emissions, capacity, buckets
        # camera, renderer.cpp:672-700
        cam = scene["cameras"][0]
        m = np.asarray(cam["transform"], dtype=f32)
        right, up = _normalized(m[0], m[1], m[2]), _normalized(m[4], m[5], m[6])
        back = _normalized(m[8], m[9], m[10])
        aspect = f32(cam["aspectRatio"]) if cam["aspectRatio"] > 0 else f32(width) / f32(height)
        scale = f32(math.tan(float(f32(f32(cam["verticalFov"]) * f32(0.5)))))
        p.eye = R.Float3(m[12], m[13], m[14])
        p.cameraU = R.Float3(*(right * scale * aspect).astype(f32))
        p.cameraV = R.Float3(*(up * scale).astype(f32))
        p.cameraW = R.Float3(*(-back))
        p.lensU, p.lensV = R.Float3(*right), R.Float3(*up)
        p.aperture, p.focusDistance = cam["aperture"], cam["focusDistance"]
        depth = min(max(f32(cam["focusDistance"]), f32(cam["nearPlane"])), f32(f32(p.sceneRadius) * f32(2.0)))
        self.caustic_radius = max(f32(f32(f32(2.0) * f32(1.0)) * depth * scale / f32(height)), f32(0.0005))
        p.photonRadius = self.caustic_radius if final else 0.0
        # accumulation buffers
        n = width * height
        self.aov = [np.zeros((n, 4), f32) for _ in range(7)]
        for name, a in zip(("accumulation", "diffuse", "reflection", "refraction", "caustics", "albedoGuide",
                            "normalGuide"), self.aov):
            setattr(p, name, a.ctypes.data)
        self.output = np.zeros((n, 4), np.uint8)
        p.output = self.output.ctypes.data
        self.photons = np.zeros((capacity if final else 1, 9), f32)
        self.sorted = np.zeros((capacity if final else 1, 9), f32)
        self.cell_start = np.zeros(buckets + 1, np.uint32)
        p.photons, p.sortedPhotons = self.photons.ctypes.data, self.sorted.ctypes.data
        p.photonCellStart = self.cell_start.ctypes.data
        self.params = p
