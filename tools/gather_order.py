"""Development aid: gather time of the photon microbenchmark with the queries in row-major, 8x8-tile and
Morton-by-cell order (same queries, same photon map)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracer_amd import engine as eng, workloads


def morton2(x, z):
    def spread(v):
        v = v.astype(np.uint64) & 0xffff
        v = (v | (v << 8)) & 0x00ff00ff
        v = (v | (v << 4)) & 0x0f0f0f0f
        v = (v | (v << 2)) & 0x33333333
        v = (v | (v << 1)) & 0x55555555
        return v
    return spread(x) | (spread(z) << 1)


def main():
    eng.load()
    n_ph, buckets, radius = 1 << 23, workloads.PHOTON_BUCKETS, 0.0015
    photons = torch.from_numpy(workloads.caustic_photons(n_ph, seed=2)).cuda()
    grid = eng.PhotonGrid(n_ph, buckets)
    grid.build(photons, radius)
    w, h = 1920, 1080
    xs = (np.arange(w, dtype=np.float32) + 0.5) / w * 2.0 - 1.0
    zs = ((np.arange(h, dtype=np.float32) + 0.5) / h * 2.0 - 1.0) * (h / w)
    qp = np.zeros((h, w, 3), dtype=np.float32)
    qp[..., 0], qp[..., 2] = xs[None, :], zs[:, None]
    base = qp.reshape(-1, 3)
    cx = np.floor(base[:, 0] / radius).astype(np.int64) + (1 << 15)
    cz = np.floor(base[:, 2] / radius).astype(np.int64) + (1 << 15)
    orders = {"row-major": np.arange(len(base)),
              "tiles 8x8": np.argsort((np.arange(h)[:, None] // 8 * (w // 8) + np.arange(w)[None, :] // 8).reshape(-1) * 64
                                      + (np.arange(h)[:, None] % 8 * 8 + np.arange(w)[None, :] % 8).reshape(-1), kind="stable"),
              "morton cell": np.argsort(morton2(cx, cz), kind="stable"),
              "random": np.random.default_rng(1).permutation(len(base))}
    for name, order in orders.items():
        q = torch.from_numpy(np.ascontiguousarray(base[order])).cuda()
        qn = torch.zeros_like(q)
        qn[:, 1] = 1.0
        flux = torch.empty_like(q)
        grid.gather(q, qn, out=flux)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(5):
            grid.gather(q, qn, out=flux)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / 5
        print(f"{name:12s} {ms:7.3f} ms  {len(base) / ms / 1e3:8.1f} M queries/s  checksum {float(flux.double().sum()):.6e}", flush=True)


main()
