"""Shared test plumbing: the same scene dict fed to the engine (GPU) and to the oracle (CPU)."""
import numpy as np


def oracle_scene(oracle, scene):
    s = oracle.Scene()
    for g in scene["geometries"]:
        s.add_geometry(g["vertices"], g["indices"], g["material"])
    for gi, m in scene["instances"]:
        s.add_instance(gi, m)
    s.commit()
    return s


def engine_scene(engine, scene):
    import torch
    blases = []
    for g in scene["geometries"]:
        v = torch.from_numpy(np.ascontiguousarray(g["vertices"], dtype=np.float32)).cuda()
        i = torch.from_numpy(np.ascontiguousarray(g["indices"].astype(np.int32))).cuda()
        blases.append(engine.Blas(v, i))
    top = engine.Scene([(blases[gi], m) for gi, m in scene["instances"]])
    return top, blases


def compare_closest(hits_gpu, inst_gpu, tuv, prim, inst):
    """Bit-exact comparison; returns the number of mismatching rays and a report."""
    h = hits_gpu.cpu().numpy()
    t_g, u_g, v_g = h[:, 0], h[:, 1], h[:, 2]
    prim_g = h[:, 3].view(np.uint32)
    inst_g = inst_gpu.cpu().numpy().view(np.uint32)
    bad = (prim_g != prim) | (inst_g != inst) | (t_g.view(np.uint32) != tuv[:, 0].view(np.uint32)) | \
        (u_g.view(np.uint32) != tuv[:, 1].view(np.uint32)) | (v_g.view(np.uint32) != tuv[:, 2].view(np.uint32))
    idx = np.nonzero(bad)[0]
    report = [(int(i), int(prim_g[i]), int(prim[i]), int(inst_g[i]), int(inst[i]), float(t_g[i]), float(tuv[i, 0]))
              for i in idx[:10]]
    return len(idx), report


class HostParams:
    """A host-memory copy of a renderer's launch parameters for the oracle: the tables the device
    pointers refer to are copied back, the seven accumulation buffers are fresh zeroed arrays."""

    def __init__(self, renderer, scene):
        import ctypes as C
        from raytracer_amd import renderer as R
        p = renderer.launch_params()
        self.keep = []
        q = R.LaunchParams.from_buffer_copy(p)
        n_pix = p.width * p.height

        def table(ptr, nbytes):
            a = R.device_to_host(ptr, nbytes) if ptr and nbytes else np.zeros(max(nbytes, 1), np.uint8)
            self.keep.append(a)
            return a

        mats = table(p.materials, 272 * len(scene["materials"]))
        q.materials = mats.ctypes.data
        lights = table(p.lights, 92 * p.lightCount)
        q.lights = lights.ctypes.data if p.lightCount else None
        n_tex = len(scene.get("textures", []))
        if n_tex:
            tex = table(p.textures, 24 * n_tex)
            recs = (R.GpuTexture * n_tex).from_buffer(tex)
            for r in recs:
                px = table(r.pixels, r.width * r.height * 4)
                r.pixels = px.ctypes.data
            q.textures = tex.ctypes.data
        else:
            q.textures = None
        if p.environment:
            n_env = p.environmentWidth * p.environmentHeight
            q.environment = table(p.environment, 16 * n_env).ctypes.data
            q.environmentCdf = table(p.environmentCdf, 4 * n_env).ctypes.data
        self.aov = [np.zeros((n_pix, 4), np.float32) for _ in range(7)]
        for name, a in zip(("accumulation", "diffuse", "reflection", "refraction", "caustics", "albedoGuide",
                            "normalGuide"), self.aov):
            setattr(q, name, a.ctypes.data)
        self.output = np.zeros((n_pix, 4), np.uint8)
        q.output = self.output.ctypes.data
        self.photons = np.zeros((max(p.photonCapacity, 1), 9), np.float32)
        self.sorted = np.zeros((max(p.photonCapacity, 1), 9), np.float32)
        self.cell_start = np.zeros(p.photonBucketCount + 1, np.uint32)
        q.photons = self.photons.ctypes.data
        q.sortedPhotons = self.sorted.ctypes.data
        q.photonCellStart = self.cell_start.ctypes.data
        q.photonCount = None
        q.scene = 0
        self.params = q


def rel_rmse(a, b):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).mean()) / max(np.sqrt((b ** 2).mean()), 1e-30))
