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
