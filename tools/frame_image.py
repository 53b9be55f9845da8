"""Renders a few samples of a workload (default: the dragon stand-in) with the preview profile and saves the seven
accumulation buffers: two builds of the library can be compared bit for bit
(python tools/frame_image.py out.npy [workload function])."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

from raytracer_amd import renderer as R, workloads

R.load()
w, h = 480, 270
scene = getattr(workloads, sys.argv[2] if len(sys.argv) > 2 else "dragon_cornell")()
r = R.Renderer(scene, w, h, 0)
for _ in range(4):
    r.render()
r.synchronize()
buffers = [R.device_to_host(r.accumulation_buffer(k), w * h * 16).view(np.float32).reshape(h, w, 4) for k in range(7)]
np.save(sys.argv[1], np.stack(buffers))
print(r.statistics(), float(buffers[0][..., :3].sum()))
