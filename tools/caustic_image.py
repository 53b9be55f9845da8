"""Renders the caustic-lab scene with the final profile and saves the seven accumulation buffers:
two builds of the library can be compared bit for bit (python tools/caustic_image.py out.npy)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

from raytracer_amd import renderer as R, workloads

R.load()
w, h = 192, 108
r = R.Renderer(workloads.caustic_lab(), w, h, 1)
r.set_photon_budget(1 << 19, 1 << 18, 1 << 16)
for _ in range(5):
    r.render()
r.synchronize()
buffers = [R.device_to_host(r.accumulation_buffer(k), w * h * 16).view(np.float32).reshape(h, w, 4) for k in range(7)]
np.save(sys.argv[1], np.stack(buffers))
print(r.statistics(), float(buffers[4][..., :3].sum()))
